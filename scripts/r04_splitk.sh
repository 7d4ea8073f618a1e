#!/bin/bash
# deterministic write-through split-K as the default: prompt-GEMM tests, soak of the wide model (every launch of an 8B-shaped layer), op-level timing against the f32-atomics form
cd "$(dirname "$0")/.."; mkdir -p gpurun_out
for atom in 0 1; do echo "== CDNA4_SPLITK_ATOMICS=$atom"; CDNA4_SPLITK_ATOMICS=$atom timeout 120 python scripts/nt_bench.py --case 12:4096:4096:512 --case 14:4096:14336:512 --case 12:1024:4096:512 --case 12:4096:4096:128 --case 12:4096:4096:64 --iters 100 2>&1 | cut -c1-200; done
timeout 200 python scripts/soak_logits.py --iters 200 --models wide,dense --budget-s 150 --out gpurun_out/r04_soak_splitk.json 2>&1 | tail -18
timeout 400 python -m pytest tests/test_gpu_prefill.py tests/test_gpu_attn_fused.py -q -m gpu -x -p no:cacheprovider 2>&1 | tail -6
M=/tmp/llama3-8b-synth-q4km-32.gguf
[ -f $M ] || python tests/gguf_synth.py $M 32 > /dev/null || exit 1
for atom in 0 1; do CDNA4_SPLITK_ATOMICS=$atom timeout 300 oracle/_ref/llama/bin/llama-bench -m $M -p 512 -n 0 -ngl 99 -fa 1 -t 8 -r 10 -o json 2>/dev/null | python -c "
import json,sys
for x in json.load(sys.stdin): print('CDNA4_SPLITK_ATOMICS=$atom pp%d %.1f +- %.1f tok/s' % (x['n_prompt'], x['avg_ts'], x['stddev_ts']))"; done
