#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; ROOT=$PWD
O=$ROOT/gpurun_out/r3y; mkdir -p $O
export TMPDIR=/tmp; cd /tmp
for v in base ks2; do
  E="X=1"; [ $v = ks2 ] && E="CDNA4_GEMM_KS2_NT4=1"
  env $E MB_ONLY_N=512 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$v -o p -- python $ROOT/scripts/mb_prefill.py $v > $O/$v.log 2>&1
  echo "== $v"; grep -h "gemm_mfma\|rows_to_f16\|fillBuffer" $O/$v/p_kernel_stats.csv | cut -d, -f1-4,6,7 | cut -c1-170
  find $O -name "*kernel_trace.csv" -delete
done
