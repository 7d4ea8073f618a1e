#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r3m; mkdir -p $O
export TMPDIR=/tmp
md5sum ik_llama.cpp_amd/libggml-hip-cdna4.so > $O/lib.md5
MB_MOE_T=128,512,2048 timeout 600 python scripts/microbench.py moe mixtral > $O/moe_mixtral.log 2>&1
MB_MOE_T=512,2048 timeout 600 python scripts/microbench.py moe > $O/moe_qwen.log 2>&1
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_kt.py tests/test_gpu_round2.py -q -k "moe or id" 2>&1 | tail -6 > $O/tests1.log
timeout 900 python -m pytest tests/test_gpu_llama.py -q -k "moe" 2>&1 | tail -6 > $O/tests2.log
grep moe $O/moe_mixtral.log $O/moe_qwen.log; tail -n 4 $O/tests1.log $O/tests2.log
