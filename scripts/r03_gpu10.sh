#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r3j; mkdir -p $O
export TMPDIR=/tmp
md5sum ik_llama.cpp_amd/libggml-hip-cdna4.so > $O/lib.md5
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_ggml_backend.py tests/test_gpu_bitnet.py -q 2>&1 | tail -15 > $O/tests1.log
timeout 600 python -m pytest tests/test_gpu_legacy_quants.py -q -x -k "IQ2_XXS or IQ2_XS or IQ3_XXS or IQ1_S or IQ1_M or iq2_xxs or iq2_xs or iq3_xxs or iq1" 2>&1 | tail -5 > $O/tests2.log
timeout 600 python scripts/iq_exp.py one base > $O/iq_base.log 2>&1
timeout 900 python bench.py --steps 5 --warmup 2 > $O/bench.json 2> $O/bench.err
tail -n 6 $O/tests1.log $O/tests2.log; cat $O/iq_base.log
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r3j/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['roofline'])
for k,v in d.get('configs',{}).items(): print(k, v.get('value'), v.get('roofline'))
PY
