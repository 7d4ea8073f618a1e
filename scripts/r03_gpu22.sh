#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r3v; mkdir -p $O; rm -f $O/iq.log
export TMPDIR=/tmp
timeout 300 python scripts/iq_exp.py one base >> $O/iq.log 2>&1
CDNA4_LIB=$PWD/ik_llama.cpp_amd/exp/lib_iq_w12.so CDNA4_GEMV_WAVES=12 CDNA4_GEMV_PER_CU=1 timeout 300 python scripts/iq_exp.py one w12 >> $O/iq.log 2>&1
CDNA4_LIB=$PWD/ik_llama.cpp_amd/exp/lib_iq_w12.so CDNA4_GEMV_WAVES=8 CDNA4_GEMV_PER_CU=1 timeout 300 python scripts/iq_exp.py one lb768_w8 >> $O/iq.log 2>&1
grep -v amdgpu.ids $O/iq.log
