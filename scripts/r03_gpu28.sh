#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r3B; mkdir -p $O
export TMPDIR=/tmp
md5sum ik_llama.cpp_amd/libggml-hip-cdna4.so > $O/lib.md5
( time timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | grep -v "cluster\|RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -6 ) > $O/tests_all.log 2>&1
cat $O/tests_all.log
bash scripts/r03_profiles.sh 2>&1 | tail -12
