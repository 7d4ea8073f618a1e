#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r3f; mkdir -p $O
export TMPDIR=/tmp
md5sum ik_llama.cpp_amd/libggml-hip-cdna4.so > $O/lib.md5
( time timeout 900 python -m pytest tests/test_gpu_bitnet.py tests/test_gpu_prefill.py tests/test_gpu_legacy_quants.py tests/test_gpu_llama.py tests/test_gpu_ggml_backend.py -q ) > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log
