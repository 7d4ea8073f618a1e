#!/bin/bash
# First GPU call of round 4 (torch-free: charged ~30 s): the opt-in decode-attention instantiation of profiles/r03_notes.md section 12 against the default one --
# time and float64 check at the llama-bench tg shapes (32 q heads / 8 KV heads, 128 and 256 keys), a ragged context (clamped tail rows), a 70B-shaped head count, then the
# reference-backed flash-attention tests with the knob on.  If all of it is green and faster: make FAST the default (csrc/ops.hip, cdna4_op_flash_attn) and port the two
# default for both kernels (the split-KV kernel from 1024 keys on carries the same opt-in instantiation: fa:32:8:4096 exercises it).
#   gpurun --timeout 120 -- 'bash scripts/r04_first.sh'
cd "$(dirname "$0")/.."; mkdir -p gpurun_out
OPS="--op fa:32:8:256 --op fa:32:8:128 --op fa:32:8:200 --op fa:32:8:77 --op fa:64:8:256 --op fa:32:8:1000"
OPS_SPLIT="--op fa:32:8:4096 --op fa:32:8:3000"
for knob in 0 1; do
  CDNA4_FA_FAST_ADDR=$knob CDNA4_FA_SPLIT_MIN_KV=100000 timeout 60 python scripts/nt_bench.py $OPS --check --iters 300 > gpurun_out/r04_fa_knob$knob.log 2>&1; echo "knob $knob rc=$?" >> gpurun_out/r04_fa_knob$knob.log
  CDNA4_FA_FAST_ADDR=$knob timeout 60 python scripts/nt_bench.py $OPS_SPLIT --check --iters 200 >> gpurun_out/r04_fa_knob$knob.log 2>&1; echo "knob $knob (split-KV kernel) rc=$?" >> gpurun_out/r04_fa_knob$knob.log
done
CDNA4_FA_FAST_ADDR=1 timeout 200 python -m pytest tests/test_gpu_ops.py -q -m gpu -k "flash or attn" -p no:cacheprovider > gpurun_out/r04_fa_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r04_fa_pytest.log
# the lean q,k,v flush (gemv.cuh FX = 4, profiles/r03_notes.md section 12): logits of the tiny models with the knob on (the decode steps go through it), then llama-bench tg on the
# tiny dense model with the knobs off / on (a 2-layer model: launch-bound, the per-launch tails are what it shows)
CDNA4_GEMV_QKV_LEAN=1 CDNA4_FA_FAST_ADDR=1 timeout 300 python -m pytest tests/test_gpu_llama.py -q -m gpu -k "logits_offloaded or hip_graph or r4_model" -p no:cacheprovider > gpurun_out/r04_lean_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r04_lean_pytest.log
tail -3 gpurun_out/r04_lean_pytest.log
paste -d'\n' gpurun_out/r04_fa_knob0.log gpurun_out/r04_fa_knob1.log | cut -c1-200; tail -3 gpurun_out/r04_fa_pytest.log
