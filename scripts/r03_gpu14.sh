#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r3n; mkdir -p $O
export TMPDIR=/tmp
md5sum ik_llama.cpp_amd/libggml-hip-cdna4.so > $O/lib.md5
for rb in 8 4 2 1; do
  echo "== CDNA4_MOE_RB=$rb" >> $O/moe.log
  CDNA4_MOE_RB=$rb MB_MOE_T=128,512,2048 timeout 600 python scripts/microbench.py moe mixtral 2>&1 | grep "moe q4" >> $O/moe.log
  CDNA4_MOE_RB=$rb MB_MOE_T=128,512,2048 timeout 600 python scripts/microbench.py moe 2>&1 | grep "moe q4" >> $O/moe.log
done
cut -c1-60,88-100,150-175 $O/moe.log
