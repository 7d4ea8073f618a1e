#!/bin/bash
# fused attention + attn_output launch: C-ABI bit-identity test, attention / logits tests, soak, then llama-bench with and without it
cd "$(dirname "$0")/.."; mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_gpu_attn_fused.py -q -m gpu -x -p no:cacheprovider 2>&1 | tail -15
timeout 300 python -m pytest tests/test_gpu_llama.py -q -m gpu -x -p no:cacheprovider -k "logits or soak or hip_graph" 2>&1 | tail -5
timeout 200 python scripts/soak_logits.py --iters 300 --models wide,dense --budget-s 200 --out gpurun_out/r04_soak_attn.json 2>&1 | tail -20
M=/tmp/llama3-8b-synth-q4km-32.gguf
[ -f $M ] || python tests/gguf_synth.py $M 32 > /dev/null || exit 1
for off in 0 1; do
  if [ $off = 1 ]; then export CDNA4_NO_ATTN_FUSION=1; fi
  GGML_CDNA4_STATS=1 timeout 300 oracle/_ref/llama/bin/llama-bench -m $M -p 0 -n 128 -ngl 99 -fa 1 -t 8 -r 5 -o json > gpurun_out/r04_attn_$off.json 2> gpurun_out/r04_attn_$off.err
  python - <<PY
import json
for x in json.load(open("gpurun_out/r04_attn_$off.json")): print("CDNA4_NO_ATTN_FUSION=$off n_prompt=%d n_gen=%d  %.1f +- %.1f tok/s" % (x["n_prompt"], x["n_gen"], x["avg_ts"], x["stddev_ts"]))
PY
  grep "graph_compute calls" gpurun_out/r04_attn_$off.err | tail -1
done
