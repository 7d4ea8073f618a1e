#!/bin/bash
# round-3 GPU call 3: window-reduce tests (two-shot fix, 4 ranks), split-K determinism, long-prompt kernel, grouped-MoE tile change, prefill microbench A/B
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r3c; mkdir -p $O
export TMPDIR=/tmp
md5sum ik_llama.cpp_amd/libggml-hip-cdna4.so > $O/lib.md5
( time timeout 600 python -m pytest tests/test_gpu_window_reduce.py -q ) > $O/pytest_window.log 2>&1; echo "rc=$?" >> $O/pytest_window.log
( time timeout 900 python -m pytest tests/test_gpu_prefill.py tests/test_gpu_round2.py -q ) > $O/pytest_prefill.log 2>&1; echo "rc=$?" >> $O/pytest_prefill.log
AB=ik_llama.cpp_amd/build/ab_6e9a64af4f35/libggml-hip-cdna4.so
timeout 300 python scripts/mb_prefill.py new > $O/mb_prefill_new.log 2>&1
[ -f "$AB" ] && CDNA4_LIB=$AB timeout 300 python scripts/mb_prefill.py old > $O/mb_prefill_old.log 2>&1
MB_ONLY_N=512 CDNA4_GEMM_KSPLIT_ATOMIC=1 timeout 200 python scripts/mb_prefill.py ksatomic > $O/mb_prefill_ksatomic.log 2>&1
CDNA4_MOE_NT=2 timeout 300 python scripts/mb_prefill.py moe_nt2 2>&1 | grep moe > $O/mb_prefill_moe_nt2.log
for v in 0 2 3; do MB_ONLY_N=4096 CDNA4_GEMM_BIG_V=$v timeout 200 python scripts/mb_prefill.py bigV$v > $O/mb_prefill_bigV$v.log 2>&1; done
MB_ONLY_N=4096 CDNA4_GEMM_BIG=0 timeout 200 python scripts/mb_prefill.py nobig > $O/mb_prefill_nobig.log 2>&1
MB_ONLY_N=1024 timeout 200 python scripts/mb_prefill.py big1024 > $O/mb_prefill_big1024.log 2>&1
MB_ONLY_N=1024 CDNA4_GEMM_BIG=0 timeout 200 python scripts/mb_prefill.py nobig1024 > $O/mb_prefill_nobig1024.log 2>&1
