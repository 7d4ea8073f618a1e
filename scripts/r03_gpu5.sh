#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r3e; mkdir -p $O
export TMPDIR=/tmp
md5sum ik_llama.cpp_amd/libggml-hip-cdna4.so > $O/lib.md5
( time timeout 900 python -m pytest tests/test_gpu_bitnet.py tests/test_gpu_prefill.py tests/test_gpu_parity.py tests/test_gpu_legacy_quants.py -q ) > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log
timeout 300 python scripts/mb_prefill.py new 2>&1 | grep -v amdgpu | grep "moe\|512" > $O/mb_prefill_new.log
