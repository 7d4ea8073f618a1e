#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; ROOT=$PWD
O=$ROOT/gpurun_out/r3A; mkdir -p $O
export TMPDIR=/tmp; cd /tmp
CDNA4_GEMM_KS2_NT4=1 MB_ONLY_N=512 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks2 -o p -- python $ROOT/scripts/mb_prefill.py ks2 > $O/ks2.log 2>&1
find $O -name "*kernel_trace.csv" -delete
cd $ROOT
python - <<'PY'
import csv
for r in csv.DictReader(open('gpurun_out/r3A/ks2/p_kernel_stats.csv')):
    if 'gemm_mfma' in r['Name']:
        print("%-72s calls %4s avg %8.1f us min %7.1f max %7.1f" % (r['Name'][:72], r['Calls'], float(r['AverageNs'])/1e3, float(r['MinNs'])/1e3, float(r['MaxNs'])/1e3))
PY
timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-llama-bench > $O/bench.json 2> $O/bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r3A/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['config']['pp512_tok_s'], d['config']['tg128_tok_s'], d['roofline_prefill'])
for k,v in d.get('configs',{}).items(): print(k, v.get('value'), v['config'].get('pp%d_tok_s' % (512 if k!='c1' else 128)) if 'config' in v else None, v.get('roofline',{}).get('frac'), v.get('roofline_prefill',{}).get('frac'), v.get('roofline_prefill',{}).get('kernel_only',{}).get('frac'))
PY
