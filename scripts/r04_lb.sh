#!/bin/bash
# llama-bench through the shim on the synthetic Llama-3-8B Q4_K_M GGUF (torch-free): tok/s with graphs, then a rocprofv3 kernel trace of tg128 (graphs off: one row per kernel) and of pp512
#   gpurun --timeout 600 -- 'bash scripts/r04_lb.sh [tag]'
cd "$(dirname "$0")/.."; TAG=${1:-r04_lb}; mkdir -p gpurun_out
M=/tmp/llama3-8b-synth-q4km-32.gguf
[ -f $M ] || python tests/gguf_synth.py $M 32 > /dev/null || exit 1
GGML_CDNA4_STATS=1 timeout 300 oracle/_ref/llama/bin/llama-bench -m $M -p 512 -n 128 -ngl 99 -fa 1 -t 8 -r 5 -o json > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
python - <<PY
import json
try:
    for x in json.load(open("gpurun_out/${TAG}_bench.json")): print("n_prompt=%d n_gen=%d  %.1f +- %.1f tok/s" % (x["n_prompt"], x["n_gen"], x["avg_ts"], x["stddev_ts"]))
except Exception as e: print("no result", e)
PY
grep "cdna4\[" gpurun_out/${TAG}_bench.err | tail -3
export GGML_CDNA4_PARAMS=graphs=0
bash scripts/llama_bench_prof.sh ${TAG}_tg -p 0 -n 128 2>&1 | tail -16
bash scripts/llama_bench_prof.sh ${TAG}_pp -p 512 -n 0 2>&1 | tail -22
