#!/bin/bash
# The GPU calls of round 4, one case per measurement (each was one `gpurun` call; all torch-free: a call is charged for the box, the push and the run only).
#   gpurun --timeout 900 -- 'bash scripts/r04_gpu.sh <step> [args]'      steps: first lb fa fa2 attn proj splitk gemm_ab soak
cd "$(dirname "$0")/.."; mkdir -p gpurun_out
STEP=${1:-help}; shift
case "$STEP" in
first)
# First GPU call of round 4 (torch-free: charged ~30 s): the opt-in decode-attention instantiation of profiles/r03_notes.md section 12 against the default one --
# time and float64 check at the llama-bench tg shapes (32 q heads / 8 KV heads, 128 and 256 keys), a ragged context (clamped tail rows), a 70B-shaped head count, then the
# reference-backed flash-attention tests with the knob on.  If all of it is green and faster: make FAST the default (csrc/ops.hip, cdna4_op_flash_attn) and port the two
# default for both kernels (the split-KV kernel from 1024 keys on carries the same opt-in instantiation: fa:32:8:4096 exercises it).
#   gpurun --timeout 120 -- 'bash scripts/r04_first.sh'
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
;;
lb)
# llama-bench through the shim on the synthetic Llama-3-8B Q4_K_M GGUF (torch-free): tok/s with graphs, then a rocprofv3 kernel trace of tg128 (graphs off: one row per kernel) and of pp512
#   gpurun --timeout 600 -- 'bash scripts/r04_lb.sh [tag]'
TAG=${1:-r04_lb}
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
;;
fa)
# decode attention: per-head kernel vs split-KV kernel (write-through hand-off vs the fenced one) at short contexts; torch-free
OPS="--op fa:32:8:128 --op fa:32:8:256 --op fa:32:8:512 --op fa:32:8:768 --op fa:32:8:1024 --op fa:32:8:2048 --op fa:32:8:4096 --op fa:64:8:256"
echo "== per-head kernel (split from 100000 keys)"; CDNA4_FA_SPLIT_MIN_KV=100000 timeout 120 python scripts/nt_bench.py $OPS --check --iters 300 2>&1 | cut -c1-200
echo "== split kernel, write-through hand-off (split from 64 keys)"; CDNA4_FA_SPLIT_MIN_KV=64 timeout 120 python scripts/nt_bench.py $OPS --check --iters 300 --stress 60 2>&1 | cut -c1-260
echo "== split kernel, fenced hand-off (split from 64 keys)"; CDNA4_FA_SPLIT_MIN_KV=64 CDNA4_FA_SPLIT_FENCE=1 timeout 120 python scripts/nt_bench.py $OPS --check --iters 300 2>&1 | cut -c1-200
echo "== split kernel, write-through, 128-key chunks"; for s in 2 4; do CDNA4_FA_SPLIT_MIN_KV=64 CDNA4_FA_SPLITS=$s timeout 120 python scripts/nt_bench.py --op fa:32:8:256 --op fa:32:8:512 --check --iters 300 2>&1 | cut -c1-200; done
;;
fa2)
# decode attention, round-4 key layout (CDNA4_FA_DECODE_V2, default on) against the round-3 one: timing at several visible-key counts, float64 checks, random-mask stress; then the graph tests and llama-bench tg128
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
;;
attn)
# fused attention + attn_output launch: C-ABI bit-identity test, attention / logits tests, soak, then llama-bench with and without it
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
;;
proj)
# 512-token prompt GEMMs on grids of <= 1 workgroup per CU: K-split depth / workgroup shape knobs (the write-through split-K makes deeper splits cheap)
CASES="--case 12:4096:4096:512 --case 12:4096:14336:512 --case 14:4096:14336:512 --case 12:1024:4096:512 --case 14:1024:4096:512"
run() { echo "== $*"; env "$@" timeout 120 python scripts/nt_bench.py $CASES --iters 100 2>&1 | python -c "
import sys,json
for ln in sys.stdin:
    try: r=json.loads(ln); print('   %-22s %8.2f us  %6.1f TF' % (r['case'], r['us'], r['tflops']))
    except Exception: print(ln.rstrip()[:200])"; }
run A=default
run CDNA4_GEMM_KSPLIT_MULT=2
run CDNA4_GEMM_KS2_NT4=0
run CDNA4_GEMM_KS2_NT4=0 CDNA4_GEMM_KSPLIT_MULT=2
run CDNA4_GEMM_KS2_NT4=0 CDNA4_GEMM_KSPLIT_MULT=4
run CDNA4_GEMM_NT_MIN=2 CDNA4_GEMM_KS2_NT4=0 CDNA4_GEMM_KSPLIT_MULT=2
run CDNA4_GEMM_NT_MIN=2 CDNA4_GEMM_KS2_NT4=0 CDNA4_GEMM_KSPLIT_MULT=4
;;
splitk)
# deterministic write-through split-K as the default: prompt-GEMM tests, soak of the wide model (every launch of an 8B-shaped layer), op-level timing against the f32-atomics form
for atom in 0 1; do echo "== CDNA4_SPLITK_ATOMICS=$atom"; CDNA4_SPLITK_ATOMICS=$atom timeout 120 python scripts/nt_bench.py --case 12:4096:4096:512 --case 14:4096:14336:512 --case 12:1024:4096:512 --case 12:4096:4096:128 --case 12:4096:4096:64 --iters 100 2>&1 | cut -c1-200; done
timeout 200 python scripts/soak_logits.py --iters 200 --models wide,dense --budget-s 150 --out gpurun_out/r04_soak_splitk.json 2>&1 | tail -18
timeout 400 python -m pytest tests/test_gpu_prefill.py tests/test_gpu_attn_fused.py -q -m gpu -x -p no:cacheprovider 2>&1 | tail -6
M=/tmp/llama3-8b-synth-q4km-32.gguf
[ -f $M ] || python tests/gguf_synth.py $M 32 > /dev/null || exit 1
for atom in 0 1; do CDNA4_SPLITK_ATOMICS=$atom timeout 300 oracle/_ref/llama/bin/llama-bench -m $M -p 512 -n 0 -ngl 99 -fa 1 -t 8 -r 10 -o json 2>/dev/null | python -c "
import json,sys
for x in json.load(sys.stdin): print('CDNA4_SPLITK_ATOMICS=$atom pp%d %.1f +- %.1f tok/s' % (x['n_prompt'], x['avg_ts'], x['stddev_ts']))"; done
;;
gemm_ab)
# prompt GEMM: the in-tree library against a variant build (A/B in one process, interleaved): plain and fused launches at 512 / 4096 tokens; then the rope + KV-store tests
#   python -c "build_library(extra_flags=[...], out='ik_llama.cpp_amd/libvariant_X.so', tag='X', only=['gemm_12', 'gemm_14'])" first (ik_llama.cpp_amd/build.py)
V=${1:-ik_llama.cpp_amd/libvariant_dma_guarded.so}
timeout 300 python scripts/nt_bench.py --lib ik_llama.cpp_amd/libggml-hip-cdna4.so --lib $V --case 12:4096:4096:512 --case 12:14336:4096:512 --case 14:4096:14336:512 --case 12:1024:4096:512 --case 12:4096:4096:4096 --case 12:14336:4096:4096 --op upgate:12:14336:4096:512 --op upgate:12:14336:4096:4096 --iters 60 --rounds 3 2>&1 | python -c "
import sys,json
for ln in sys.stdin:
    try: r=json.loads(ln); print('   %-28s %-44s %9.2f us  %7.1f TF  %.4f' % (r.get('case') or r.get('op'), r['lib'], r['us'], r.get('tflops', 0), r.get('frac_mfma', 0)))
    except Exception: print(ln.rstrip()[:200])"
timeout 300 python -m pytest tests/test_gpu_ops.py -q -m gpu -x -p no:cacheprovider -k "rope" 2>&1 | tail -3
;;
fap)
# prompt attention: one wave per 32 queries over all keys (CDNA4_FA_PREFILL_V1=1) vs the keys of a query block spread over four waves; causal masks; then the parity tests
OPS="--op fa:32:8:512:512:0 --op fa:32:8:1024:512:0 --op fa:32:8:4096:512:0 --op fa:32:8:2048:2048:0 --op fa:32:8:256:64:0 --op fa:8:8:512:512:0"
echo "== v1"; CDNA4_FA_PREFILL_V1=1 timeout 200 python scripts/nt_bench.py $OPS --check --iters 100 2>&1 | cut -c1-200
echo "== split"; timeout 200 python scripts/nt_bench.py $OPS --check --iters 100 2>&1 | cut -c1-200
timeout 600 python -m pytest tests/test_gpu_ops.py -q -k "flash_attn" 2>&1 | grep -E "^E  |passed|failed" | head -20
echo "== v1 tests"; CDNA4_FA_PREFILL_V1=1 timeout 600 python -m pytest tests/test_gpu_ops.py -q -k "flash_attn" 2>&1 | grep -E "^E  |passed|failed" | head -20
;;
pf)
# prompt-pass work of round 4: split prompt attention + [ADD +] norm in the image launch -- parity first, then llama-bench through the shim with the kernel trace
timeout 900 python -m pytest tests/test_gpu_prompt_fused.py tests/test_gpu_ops.py tests/test_gpu_llama.py tests/test_gpu_ggml_backend.py tests/test_gpu_prefill.py -x -q 2>&1 | tail -8
bash scripts/r04_gpu.sh lb ${1:-r04_lb2}
;;
pp)
# llama-bench pp512 through the shim: prompt-size graphs eager (default) vs captured (rounds 1-3), blocking vs stream-queued small uploads for tg128
M=/tmp/llama3-8b-synth-q4km-32.gguf
[ -f $M ] || python tests/gguf_synth.py $M 32 > /dev/null || exit 1
run() { echo "== $1"; shift; env "$@" GGML_CDNA4_STATS=1 timeout 300 oracle/_ref/llama/bin/llama-bench -m $M -p 512 -n 128 -ngl 99 -fa 1 -t 8 -r 5 -o json > gpurun_out/pp_tmp.json 2> gpurun_out/pp_tmp.err
  python - <<PY
import json
for x in json.load(open("gpurun_out/pp_tmp.json")): print("n_prompt=%d n_gen=%d  %.1f +- %.1f tok/s  %s" % (x["n_prompt"], x["n_gen"], x["avg_ts"], x["stddev_ts"], [round(v, 1) for v in x["samples_ts"]]))
PY
  grep "small uploads\|host time" gpurun_out/pp_tmp.err | tail -2 | cut -c1-220; }
run "default (eager prompt graphs, queued small uploads)" A=1
run "captured prompt graphs" GGML_CDNA4_GRAPH_MAX_BATCH=1000000
run "no kernel preload at upload" GGML_CDNA4_NO_PRELOAD=1
run "default again" A=1
;;
smallmm)
# the small prompt GEMMs of an 8B layer at 512 tokens (v: 1024 x 4096 Q6_K, k+v, o): token-tile / K-split knobs
CASES="--case 14:1024:4096:512 --case 12:1024:4096:512 --case 12:2048:4096:512 --case 12:4096:4096:512 --case 12:5120:4096:512"
echo "== default"; timeout 120 python scripts/nt_bench.py $CASES --iters 100 2>&1 | cut -c1-220
echo "== CDNA4_GEMM_NT_MIN=4"; CDNA4_GEMM_NT_MIN=4 timeout 120 python scripts/nt_bench.py $CASES --iters 100 2>&1 | cut -c1-220
echo "== CDNA4_GEMM_KSPLIT_MULT=2"; CDNA4_GEMM_KSPLIT_MULT=2 timeout 120 python scripts/nt_bench.py $CASES --iters 100 2>&1 | cut -c1-220
echo "== CDNA4_GEMM_NT_MIN=4 CDNA4_GEMM_KSPLIT_MULT=2"; CDNA4_GEMM_NT_MIN=4 CDNA4_GEMM_KSPLIT_MULT=2 timeout 120 python scripts/nt_bench.py $CASES --iters 100 2>&1 | cut -c1-220
;;
prod)
# 224-row prompt tiles (14336-row matrices at 512 tokens, one workgroup per CU): seven compute waves issuing their own LDS-DMA pieces vs + a producer wave, three buffers, counted vmcnt
OPS="--op upgate:12:14336:4096:512 --op upgate:14:14336:4096:512 --op upgate:20:14336:4096:512"
CASES="--case 12:14336:4096:512 --case 14:14336:4096:512"
echo "== CDNA4_GEMM_PROD=0"; CDNA4_GEMM_PROD=0 timeout 200 python scripts/nt_bench.py $OPS $CASES --iters 100 2>&1 | cut -c1-230
echo "== producer wave"; timeout 200 python scripts/nt_bench.py $OPS $CASES --iters 100 2>&1 | cut -c1-230
timeout 900 python -m pytest tests/test_gpu_prefill.py tests/test_gpu_prompt_fused.py -x -q 2>&1 | tail -4
;;
final)
# Round-4 evidence run: the driver's bench command, rocprofv3 kernel stats of bench.py (headline config) and of llama-bench through the shim, SQ counters of the prompt GEMM.
# Summaries are copied to profiles/ by hand (profiles/README.md).
export TMPDIR=/tmp; ROOT=$PWD; OUT=$ROOT/gpurun_out/r04; mkdir -p $OUT
md5sum ik_llama.cpp_amd/libggml-hip-cdna4.so > $OUT/lib.md5
timeout 1200 python $ROOT/bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_n1.json 2> $OUT/bench_n1.err; echo bench rc=$?
cp gpurun_out/bench_details.json $OUT/bench_details.json 2>/dev/null
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench -o bench -- python $ROOT/bench.py --no-cpu-baseline --no-llama-bench --no-pmc --no-extra-configs > $OUT/bench_stats_stdout.json 2> $OUT/bench_stats_stderr.txt; echo bench-stats rc=$?
M=/tmp/llama3-8b-synth-q4km-32.gguf; [ -f $M ] || python $ROOT/tests/gguf_synth.py $M 32 > /dev/null
GGML_CDNA4_PARAMS=graphs=0 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/llama -o lb -- $ROOT/oracle/_ref/llama/bin/llama-bench -m $M -p 512 -n 128 -ngl 99 -fa 1 -t 8 -r 3 > $OUT/llama_stdout.txt 2> $OUT/llama_stderr.txt; echo llama rc=$?
A="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVES"
B="SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_INSTS_VMEM_RD"
timeout 300 rocprofv3 --pmc $A --kernel-trace --output-format csv -d $OUT/pmc_gemm_a -o p -- python $ROOT/scripts/gemm_prof.py 4096 > /dev/null 2>&1; echo pmcA rc=$?
timeout 300 rocprofv3 --pmc $B --kernel-trace --output-format csv -d $OUT/pmc_gemm_b -o p -- python $ROOT/scripts/gemm_prof.py 4096 > /dev/null 2>&1; echo pmcB rc=$?
python $ROOT/scripts/pmc_summary.py $OUT/pmc_gemm_a/p_counter_collection.csv "rocprofv3 --pmc $A --kernel-trace -- python scripts/gemm_prof.py 4096" > $OUT/pmc_gemm_a.json
python $ROOT/scripts/pmc_summary.py $OUT/pmc_gemm_b/p_counter_collection.csv "rocprofv3 --pmc $B --kernel-trace -- python scripts/gemm_prof.py 4096" > $OUT/pmc_gemm_b.json
cd $ROOT
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*counter_collection.csv" -delete; find $OUT -name "*agent_info.csv" -delete
find $OUT -type f | head -40; du -sh $OUT
tail -c 4700 $OUT/bench_n1.json
;;
tg)
# llama-bench tg128 through the shim, A/B of one environment switch, interleaved (A B A B) on the same box:  bash scripts/r04_gpu.sh tg CDNA4_QKV_PREFETCH=0
M=/tmp/llama3-8b-synth-q4km-32.gguf
[ -f $M ] || python tests/gguf_synth.py $M 32 > /dev/null || exit 1
one() { env "$@" timeout 300 oracle/_ref/llama/bin/llama-bench -m $M -p 0 -n 128 -ngl 99 -fa 1 -t 8 -r 5 -o json 2>/dev/null | python -c "import json,sys; x=json.load(sys.stdin)[0]; print('  tg128 %.1f +- %.1f' % (x['avg_ts'], x['stddev_ts']))"; }
for i in 1 2 3; do echo "default"; one A=1; echo "$1"; one "$1"; done
;;
soak)
# 300-repetition hashed soak through libllama (scripts/soak_logits.py): standard switch combinations, or --bisect / --fusion-masks
python scripts/soak_logits.py "$@"
;;
*) echo "steps: first lb fa fa2 attn proj splitk gemm_ab soak" ;;
esac
