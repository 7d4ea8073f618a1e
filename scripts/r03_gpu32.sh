#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r3F; mkdir -p $O
export TMPDIR=/tmp
md5sum ik_llama.cpp_amd/libggml-hip-cdna4.so > $O/lib.md5
( time timeout 900 python -m pytest tests -q -m gpu 2>&1 | grep -v "cluster\|RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -6 ) > $O/tests_all.log 2>&1
cat $O/tests_all.log
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_n1.json 2> $O/bench_n1.err; echo bench rc=$?
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r3F/bench_n1.json').read().strip().splitlines()[-1])
print(d['value'], d['config']['pp512_tok_s'], d['config']['tg128_tok_s'], d['roofline']['frac'], d['roofline']['decode_token']['frac'], d['roofline_prefill']['frac'], d['roofline_prefill'].get('kernel_only',{}).get('frac'), d['roofline_prefill']['n4096'])
for k,v in d.get('configs',{}).items(): print(k, v.get('value'), v.get('roofline',{}).get('frac'), v.get('roofline_prefill',{}).get('frac'), v.get('roofline_prefill',{}).get('kernel_only',{}).get('frac'))
print(d['llama_bench'].get('pp512_tok_s'), d['llama_bench'].get('tg128_tok_s'), d['cpu_baseline'].get('value'))
PY
