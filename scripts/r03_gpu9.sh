#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r3i; mkdir -p $O
export TMPDIR=/tmp
md5sum ik_llama.cpp_amd/libggml-hip-cdna4.so > $O/lib.md5
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_ggml_backend.py tests/test_gpu_bitnet.py -q -x 2>&1 | tail -15 > $O/tests1.log
timeout 600 python -m pytest tests/test_gpu_prefill.py tests/test_gpu_llama.py -q -k "off_the_128 or more_weight_types" 2>&1 | tail -15 > $O/tests2.log
timeout 900 python scripts/iq_exp.py run > $O/iq_exp.log 2>&1
tail -5 $O/tests1.log $O/tests2.log; cat $O/iq_exp.log
