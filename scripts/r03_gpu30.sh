#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; ROOT=$PWD
O=$ROOT/gpurun_out/r3D; mkdir -p $O
export TMPDIR=/tmp
md5sum ik_llama.cpp_amd/libggml-hip-cdna4.so > $O/lib.md5
timeout 900 python -m pytest tests/test_gpu_ggml_backend.py tests/test_gpu_llama.py tests/test_gpu_ops.py tests/test_gpu_round2.py -q 2>&1 | grep -v "cluster\|RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -3 > $O/tests.log
cat $O/tests.log
timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-pmc --no-extra-configs > $O/bench.json 2> $O/bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r3D/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['config']['pp512_tok_s'], d['config']['tg128_tok_s'], d['llama_bench'].get('pp512_tok_s'), d['llama_bench'].get('tg128_tok_s'))
PY
