mkdir -p gpurun_out/r2c
(timeout 300 python -m pytest tests -m gpu -x -q -k "iq2 or iq3 or IQ2 or IQ3 or round2" 2>&1 | tail -5) > gpurun_out/r2c/pytest.log
for cfg in "0 0" "2 4" "3 4" "4 4" "1 8" "2 8"; do set -- $cfg; echo "== PER_CU=$1 WAVES=$2"; CDNA4_GEMV_PER_CU=$1 CDNA4_GEMV_WAVES=$2 MB_ONLY_IQ=1 timeout 120 python scripts/microbench.py gemv 2>&1 | grep -i "iq2\|iq3"; done > gpurun_out/r2c/sweep.log 2>&1
cat gpurun_out/r2c/pytest.log | tail -3; cat gpurun_out/r2c/sweep.log
