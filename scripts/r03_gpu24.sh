#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r3x; mkdir -p $O
export TMPDIR=/tmp
MB_ONLY_N=512 timeout 600 python scripts/mb_prefill.py base > $O/base.log 2>&1
CDNA4_GEMM_KS2_NT4=1 MB_ONLY_N=512 timeout 600 python scripts/mb_prefill.py ks2 > $O/ks2.log 2>&1
CDNA4_GEMM_KS2_NT4=1 timeout 600 python -m pytest tests/test_gpu_prefill.py tests/test_gpu_round2.py -q -x 2>&1 | tail -3 > $O/tests.log
grep -h "N= 512" $O/base.log $O/ks2.log; tail -2 $O/tests.log
