#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r3g; mkdir -p $O
export TMPDIR=/tmp
md5sum ik_llama.cpp_amd/libggml-hip-cdna4.so > $O/lib.md5
for i in 1 2 3; do timeout 300 python -m pytest "tests/test_gpu_llama.py::test_logits_more_weight_types_vs_cpu" -q 2>&1 | grep -E "passed|failed|AssertionError: \(" >> $O/iqk_repeat.log; done
GGML_CDNA4_NO_MM_FUSION=1 timeout 300 python -m pytest "tests/test_gpu_llama.py::test_logits_more_weight_types_vs_cpu" -q 2>&1 | grep -E "passed|failed|AssertionError: \(" > $O/iqk_nofusion.log
GGML_CDNA4_PARAMS=graphs=0 timeout 300 python -m pytest "tests/test_gpu_llama.py::test_logits_more_weight_types_vs_cpu" -q 2>&1 | grep -E "passed|failed|AssertionError: \(" > $O/iqk_nographs.log
GGML_CDNA4_PARAMS=fusion=0 timeout 300 python -m pytest "tests/test_gpu_llama.py::test_logits_more_weight_types_vs_cpu" -q 2>&1 | grep -E "passed|failed|AssertionError: \(" > $O/iqk_fusion0.log
AB=ik_llama.cpp_amd/build/ab_6e9a64af4f35/libggml-hip-cdna4.so
