#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r3k; mkdir -p $O
export TMPDIR=/tmp
md5sum ik_llama.cpp_amd/libggml-hip-cdna4.so > $O/lib.md5
timeout 900 python -m pytest tests/test_gpu_kt.py -q 2>&1 | tail -25 > $O/tests1.log
timeout 900 python -m pytest tests/test_gpu_llama.py -q -k "more_weight_types" 2>&1 | tail -25 > $O/tests2.log
timeout 600 python scripts/mb_kt.py > $O/mb_kt.log 2>&1
tail -n 12 $O/tests1.log $O/tests2.log; cat $O/mb_kt.log | tail -8
