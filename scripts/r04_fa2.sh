#!/bin/bash
# decode attention, round-4 key layout (CDNA4_FA_DECODE_V2, default on) against the round-3 one: timing at several visible-key counts, float64 checks, random-mask stress; then the graph tests and llama-bench tg128
cd "$(dirname "$0")/.."; mkdir -p gpurun_out
OPS="--op fa:32:8:256:1:256 --op fa:32:8:256:1:128 --op fa:32:8:256:1:64 --op fa:32:8:256:1:17 --op fa:32:8:128:1:100 --op fa:64:8:256:1:64 --op fa:32:8:320:1:300"
for v2 in 0 1; do echo "== CDNA4_FA_DECODE_V2=$v2"; CDNA4_FA_DECODE_V2=$v2 timeout 120 python scripts/nt_bench.py $OPS --check --iters 300 --stress $((v2 * 40)) 2>&1 | python -c "
import sys,json
for ln in sys.stdin:
    try: r=json.loads(ln); print('   %-24s %7.2f us  nmse %.2e %s' % (r['op'], r['us'], r.get('nmse_vs_f64', -1), ('stress bad %d worst %.1e' % (r['stress_bad'], r['stress_worst_nmse'])) if 'stress_bad' in r else ''))
    except Exception: print(ln.rstrip()[:200])"; done
timeout 300 python -m pytest tests/test_gpu_ops.py tests/test_gpu_attn_fused.py -q -m gpu -x -p no:cacheprovider -k "flash or attn" 2>&1 | tail -4
M=/tmp/llama3-8b-synth-q4km-32.gguf
[ -f $M ] || python tests/gguf_synth.py $M 32 > /dev/null || exit 1
for v2 in 0 1; do CDNA4_FA_DECODE_V2=$v2 timeout 300 oracle/_ref/llama/bin/llama-bench -m $M -p 0 -n 128 -ngl 99 -fa 1 -t 8 -r 5 -o json 2>/dev/null | python -c "
import json,sys
for x in json.load(sys.stdin): print('CDNA4_FA_DECODE_V2=$v2 tg%d %.1f +- %.1f tok/s' % (x['n_gen'], x['avg_ts'], x['stddev_ts']))"; done
