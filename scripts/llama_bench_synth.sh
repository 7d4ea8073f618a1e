#!/bin/bash
# llama-bench (the reference's own binary, oracle/_ref/llama, linked against the shim) on a full-size synthetic GGUF of the BASELINE configs.
#   scripts/llama_bench_synth.sh [n_layer=32] [extra llama-bench args...]        (run on a GPU box; writes gpurun_out/llama_bench_*.json)
cd "$(dirname "$0")/.."
NL=${1:-32}; shift
M=/tmp/llama3-8b-synth-q4km-$NL.gguf
[ -f $M ] || python tests/gguf_synth.py $M $NL > /dev/null || exit 1
mkdir -p gpurun_out
timeout 600 oracle/_ref/llama/bin/llama-bench -m $M -p 512 -n 128 -ngl 99 -fa 1 -t 8 -r 3 -o json "$@" > gpurun_out/llama_bench_$NL.json 2> gpurun_out/llama_bench_$NL.err
echo rc=$?
python - <<PY
import json
try:
    r = json.load(open("gpurun_out/llama_bench_$NL.json"))
    for x in r: print("n_prompt=%d n_gen=%d  %.1f +- %.1f tok/s" % (x["n_prompt"], x["n_gen"], x["avg_ts"], x["stddev_ts"]))
except Exception as e:
    print("no result:", e); print(open("gpurun_out/llama_bench_$NL.err").read()[-3000:])
PY
