#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; ROOT=$PWD
O=$ROOT/gpurun_out/r3C; mkdir -p $O
export TMPDIR=/tmp
md5sum ik_llama.cpp_amd/libggml-hip-cdna4.so > $O/lib.md5
timeout 900 python -m pytest tests/test_gpu_prefill.py tests/test_gpu_round2.py tests/test_gpu_parity.py -q -x 2>&1 | grep -v "cluster\|RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -3 > $O/tests.log
cd /tmp
MB_ONLY_N=512 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/p512 -o p -- python $ROOT/scripts/mb_prefill.py new > $O/p512.log 2>&1
find $O -name "*kernel_trace.csv" -delete
cd $ROOT; cat $O/tests.log
python - <<'PY'
import csv
for r in csv.DictReader(open('gpurun_out/r3C/p512/p_kernel_stats.csv')):
    if 'gemm_mfma' in r['Name'] or 'rows_to' in r['Name']:
        print("%-72s calls %4s avg %8.1f us min %7.1f max %7.1f" % (r['Name'][:72], r['Calls'], float(r['AverageNs'])/1e3, float(r['MinNs'])/1e3, float(r['MaxNs'])/1e3))
PY
timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-pmc --no-extra-configs > $O/bench.json 2> $O/bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r3C/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['config']['pp512_tok_s'], d['config']['tg128_tok_s'], d['roofline_prefill']['frac'], d['llama_bench'].get('pp512_tok_s'), d['llama_bench'].get('tg128_tok_s'))
PY
