#!/bin/bash
# llama-bench through the shim on a synthetic Mixtral-8x7B-shaped GGUF with N layers (default 4: 3.3 GB), + rocprofv3 kernel stats (graphs off)
cd "$(dirname "$0")/.."; ROOT=$PWD
NL=${1:-4}
M=/tmp/mixtral-synth-$NL.gguf
[ -f $M ] || python - <<PY
import sys; sys.path.insert(0, "$ROOT/tests")
import gguf_synth as gs
gs.bench_model("$M", n_embd=4096, n_ff=14336, n_head=32, n_head_kv=8, n_layer=$NL, n_vocab=32000, n_expert=8, n_used=2, name="Mixtral-8x7B-synth")
PY
mkdir -p gpurun_out
timeout 600 oracle/_ref/llama/bin/llama-bench -m $M -p 512 -n 128 -ngl 99 -fa 1 -t 8 -r 3 -o json > gpurun_out/lb_moe.json 2> gpurun_out/lb_moe.err; echo rc=$?
python - <<PY
import json
try:
    for x in json.load(open("gpurun_out/lb_moe.json")): print("n_prompt=%d n_gen=%d  %.1f +- %.1f tok/s" % (x["n_prompt"], x["n_gen"], x["avg_ts"], x["stddev_ts"]))
except Exception as e: print("no result", e); print(open("gpurun_out/lb_moe.err").read()[-2000:])
PY
export TMPDIR=/tmp; OUT=$ROOT/gpurun_out/lb_moe_prof; mkdir -p $OUT
cd /tmp && GGML_CDNA4_PARAMS=graphs=0 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o run -- $ROOT/oracle/_ref/llama/bin/llama-bench -m $M -p 0 -n 64 -ngl 99 -fa 1 -t 8 -r 2 > $OUT/out.txt 2> $OUT/err.txt
find $OUT -name "*kernel_trace.csv" -delete
python - <<PY
import csv, glob
for f in glob.glob("$OUT/**/*kernel_stats.csv", recursive=True):
    rows = list(csv.DictReader(open(f))); tot = sum(float(r["TotalDurationNs"]) for r in rows)
    print("decode kernels total ms", tot / 1e6)
    for r in rows[:18]: print("%6.2f%% %8d calls %8.2f us  %s" % (100 * float(r["TotalDurationNs"]) / tot, int(r["Calls"]), float(r["AverageNs"]) / 1e3, r["Name"][:100]))
PY
