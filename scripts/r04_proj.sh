#!/bin/bash
# 512-token prompt GEMMs on grids of <= 1 workgroup per CU: K-split depth / workgroup shape knobs (the write-through split-K makes deeper splits cheap)
cd "$(dirname "$0")/.."
CASES="--case 12:4096:4096:512 --case 12:4096:14336:512 --case 14:4096:14336:512 --case 12:1024:4096:512 --case 14:1024:4096:512"
run() { echo "== $*"; env "$@" timeout 120 python scripts/nt_bench.py $CASES --iters 100 2>&1 | python -c "
import sys,json
for ln in sys.stdin:
    try: r=json.loads(ln); print('   %-22s %8.2f us  %6.1f TF' % (r['case'], r['us'], r['tflops']))
    except Exception: print(ln.rstrip()[:200])"; }
run A=default
run CDNA4_GEMM_KSPLIT_MULT=2
run CDNA4_GEMM_KS2_NT4=0
run CDNA4_GEMM_KS2_NT4=0 CDNA4_GEMM_KSPLIT_MULT=2
run CDNA4_GEMM_KS2_NT4=0 CDNA4_GEMM_KSPLIT_MULT=4
run CDNA4_GEMM_NT_MIN=2 CDNA4_GEMM_KS2_NT4=0 CDNA4_GEMM_KSPLIT_MULT=2
run CDNA4_GEMM_NT_MIN=2 CDNA4_GEMM_KS2_NT4=0 CDNA4_GEMM_KSPLIT_MULT=4
