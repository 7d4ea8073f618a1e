#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r3l; mkdir -p $O
export TMPDIR=/tmp
md5sum ik_llama.cpp_amd/libggml-hip-cdna4.so > $O/lib.md5
timeout 900 python -m pytest tests/test_gpu_kt.py tests/test_gpu_prefill.py -q -x 2>&1 | tail -25 > $O/tests1.log
timeout 900 python -m pytest tests/test_gpu_ggml_backend.py tests/test_gpu_ops.py -q -x 2>&1 | tail -8 > $O/tests2.log
tail -n 12 $O/tests1.log $O/tests2.log
