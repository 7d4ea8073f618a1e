#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; ROOT=$PWD
O=$ROOT/gpurun_out/r3z; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_prefill.py tests/test_gpu_round2.py tests/test_gpu_kt.py tests/test_gpu_bitnet.py -q -x 2>&1 | tail -3 > $O/tests.log
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_legacy_quants.py -q -x -k "moe or id or prefill or prompt or mfma or batch" 2>&1 | tail -3 > $O/tests2.log
MB_MOE_T=512 timeout 300 python scripts/microbench.py moe mixtral 2>&1 | grep "moe q4" > $O/moe.log
cd /tmp
MB_ONLY_N=512 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/p512 -o p -- python $ROOT/scripts/mb_prefill.py new > $O/p512.log 2>&1
MB_ONLY_N=4096 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/p4096 -o p -- python $ROOT/scripts/mb_prefill.py new > $O/p4096.log 2>&1
find $O -name "*kernel_trace.csv" -delete
cd $ROOT; tail -n 2 $O/tests.log $O/tests2.log; cat $O/moe.log | cut -c1-220
python - <<'PY'
import csv
for n in ('p512','p4096'):
    print("==", n)
    for r in csv.DictReader(open('gpurun_out/r3z/%s/p_kernel_stats.csv'%n)):
        if 'gemm_mfma' in r['Name'] or 'rows_to' in r['Name']:
            print("%-72s calls %4s avg %8.1f us min %7.1f max %7.1f" % (r['Name'][:72], r['Calls'], float(r['AverageNs'])/1e3, float(r['MinNs'])/1e3, float(r['MaxNs'])/1e3))
PY
