#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r3u; mkdir -p $O
export TMPDIR=/tmp
md5sum ik_llama.cpp_amd/libggml-hip-cdna4.so > $O/lib.md5
timeout 300 python scripts/iq_exp.py one base > $O/iq.log 2>&1
MB_ONLY=IQ timeout 600 python scripts/mb_legacy.py > $O/mb_legacy.log 2>&1
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_legacy_quants.py tests/test_gpu_r4.py -q -x 2>&1 | tail -4 > $O/tests1.log
timeout 900 python -m pytest tests/test_gpu_llama.py tests/test_gpu_ggml_backend.py -q -x 2>&1 | tail -4 > $O/tests2.log
grep -v amdgpu.ids $O/iq.log; tail -n 3 $O/tests1.log $O/tests2.log; tail -40 $O/mb_legacy.log | cut -c1-200
