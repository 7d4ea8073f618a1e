#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r3p; mkdir -p $O
export TMPDIR=/tmp
md5sum ik_llama.cpp_amd/libggml-hip-cdna4.so > $O/lib.md5
( time timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | tail -15 ) > $O/tests_all.log 2>&1
timeout 900 python bench.py --steps 5 --warmup 2 > $O/bench.json 2> $O/bench.err
tail -n 8 $O/tests_all.log
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r3p/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['roofline']['frac'], d['roofline']['decode_token']['frac'], d['roofline_prefill'])
for k,v in d.get('configs',{}).items(): print(k, v.get('value'), v.get('roofline',{}).get('frac'), v.get('roofline_prefill'))
print(d.get('llama_bench')); print(d['cpu_baseline'].get('value'), d['cpu_baseline'].get('sample'))
PY
