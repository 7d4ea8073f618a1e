#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r3s; mkdir -p $O
export TMPDIR=/tmp
md5sum ik_llama.cpp_amd/libggml-hip-cdna4.so > $O/lib.md5
timeout 600 python scripts/iq_exp.py one base > $O/iq_base.log 2>&1
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_r4.py tests/test_gpu_round2.py tests/test_gpu_mfma_cols.py -q -x 2>&1 | tail -6 > $O/tests1.log
timeout 900 python -m pytest tests/test_gpu_llama.py -q -k "iq or dense" 2>&1 | tail -6 > $O/tests2.log
cat $O/iq_base.log | tail -2; tail -n 4 $O/tests1.log $O/tests2.log
