#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r3t; mkdir -p $O; rm -f $O/iq.log
export TMPDIR=/tmp
timeout 300 python scripts/iq_exp.py one base >> $O/iq.log 2>&1
CDNA4_GEMV_IQ_DEPTH2=1 timeout 300 python scripts/iq_exp.py one d2 >> $O/iq.log 2>&1
CDNA4_GEMV_IQ_DEPTH2=1 CDNA4_GEMV_PER_CU=3 timeout 300 python scripts/iq_exp.py one d2_p3 >> $O/iq.log 2>&1
CDNA4_GEMV_IQ_DEPTH2=1 CDNA4_GEMV_PER_CU=4 timeout 300 python scripts/iq_exp.py one d2_p4 >> $O/iq.log 2>&1
CDNA4_GEMV_IQ_DEPTH2=1 CDNA4_GEMV_PER_CU=1 CDNA4_GEMV_WAVES=8 timeout 300 python scripts/iq_exp.py one d2_w8 >> $O/iq.log 2>&1
CDNA4_GEMV_PER_CU=3 timeout 300 python scripts/iq_exp.py one d4_p3 >> $O/iq.log 2>&1
grep -v amdgpu.ids $O/iq.log
