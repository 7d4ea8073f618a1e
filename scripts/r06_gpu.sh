#!/bin/bash
# The GPU calls of round 6, one case per measurement (each one `gpurun` call).   gpurun --timeout 900 -- 'bash scripts/r06_gpu.sh <step> [args]'
cd "$(dirname "$0")/.."; mkdir -p gpurun_out
STEP=${1:-help}; shift
case "$STEP" in
newtests)
# the parity cases added in round 6: full-size shapes (output.weight, Mixtral-size experts, 70B TP=8 shard launches), hand-off self-test + fenced fallback, the ping-pong prompt
# GEMM bit for bit against the per-wave kernel, NEOX partial rotation, the binding's argument counts
timeout 1500 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_handoff.py tests/test_abi.py -q -m "gpu or not gpu" -p no:cacheprovider 2>&1 | tail -25
timeout 900 python -m pytest tests/test_gpu_prefill.py -q -m gpu -p no:cacheprovider -k "ping_pong" 2>&1 | tail -15
timeout 600 python -m pytest tests/test_gpu_ops.py -q -m gpu -p no:cacheprovider -k "rope" 2>&1 | tail -8
;;
pp_test)
timeout 900 python -m pytest tests/test_gpu_prefill.py -q -m gpu -x -p no:cacheprovider -k "ping_pong or 4k_tokens or f16_image" 2>&1 | tail -15
;;
forms)
# prompt-GEMM forms interleaved in one process:  r06_gpu.sh forms [mb_forms.py arguments]
timeout 800 python scripts/mb_forms.py "$@" 2>&1 | tail -80
;;
forms_libs)
# variant libraries (ik_llama.cpp_amd/exp/lib_<name>.so, built by scripts/pp_exp.py) one process each, the in-tree library first and last:  r06_gpu.sh forms_libs "<mb_forms args>" name ...
ARGS="$1"; shift
echo "== in-tree"; timeout 300 python scripts/mb_forms.py $ARGS 2>&1 | tail -30
for v in "$@"; do echo "== $v"; CDNA4_LIB=$PWD/ik_llama.cpp_amd/exp/lib_$v.so timeout 300 python scripts/mb_forms.py $ARGS 2>&1 | tail -30; done
echo "== in-tree (again)"; timeout 300 python scripts/mb_forms.py $ARGS 2>&1 | tail -30
;;
pp_pmc)
# SQ counters of the fused up*gate prompt GEMM at 4096 tokens, two passes of 8 counters:  r06_gpu.sh pp_pmc <form> [type]
ROOT=$PWD; OUT=$ROOT/gpurun_out/r06_pmc_form$1; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
A="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVES"
B="SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_WAIT_INST_LDS"
CMD="python $ROOT/scripts/nt_bench.py --op upgate:${2:-12}:14336:4096:4096 --iters 6 --warmup 2 --rounds 1"
CDNA4_GEMM_WLDS=$1 timeout 300 rocprofv3 --pmc $A --kernel-trace --output-format csv -d $OUT/a -o p -- $CMD > /dev/null 2>&1; echo pmcA rc=$?
CDNA4_GEMM_WLDS=$1 timeout 300 rocprofv3 --pmc $B --kernel-trace --output-format csv -d $OUT/b -o p -- $CMD > /dev/null 2>&1; echo pmcB rc=$?
cd $ROOT
python scripts/pmc_summary.py $(find $OUT/a -name "*counter_collection.csv" | head -1) "rocprofv3 --pmc $A --kernel-trace -- CDNA4_GEMM_WLDS=$1 $CMD" > $OUT/a.json
python scripts/pmc_summary.py $(find $OUT/b -name "*counter_collection.csv" | head -1) "rocprofv3 --pmc $B --kernel-trace -- CDNA4_GEMM_WLDS=$1 $CMD" > $OUT/b.json
python - <<PY
import json
for f in ("$OUT/a.json", "$OUT/b.json"):
    d = json.load(open(f))
    for k, v in d["kernels"].items():
        if "gemm" in k: print(k[:70]); print("   ", {a: round(b) for a, b in v.items()})
PY
;;
lb)
# llama-bench through the shim, interleaved environment variants:  r06_gpu.sh lb "<llama-bench args>" VAR=1 "A=1 B=2" ...
M=/tmp/llama3-8b-synth-q4km-32.gguf
[ -f $M ] || python tests/gguf_synth.py $M 32 > /dev/null || exit 1
LBARGS="$1"; shift
one() { env $1 timeout 400 oracle/_ref/llama/bin/llama-bench -m $M $LBARGS -ngl 99 -fa 1 -t 8 -r 3 -o json 2>/dev/null | python -c "
import json,sys
for x in json.load(sys.stdin): print('  p%d n%d ub%d: %.1f +- %.1f tok/s' % (x['n_prompt'], x['n_gen'], x.get('n_ubatch', 0), x['avg_ts'], x['stddev_ts']))"; }
for i in 1 2; do echo "default"; one A=1; for v in "$@"; do echo "$v"; one "$v"; done; done
;;
mtl)
# phase timeline of the per-wave prompt GEMM on the 512-token launches (scripts/mfma_timeline.py):  r06_gpu.sh mtl <lib name> [type] [M:K:N ...]
L=$1; shift
CDNA4_LIB=$PWD/ik_llama.cpp_amd/exp/lib_$L.so timeout 600 python scripts/mfma_timeline.py "$@" 2>&1 | tail -60
;;
final)
# Round-6 evidence run: the driver's bench command, rocprofv3 kernel stats of bench.py (headline config) and of llama-bench through the shim (graphs off: one row per kernel).
# Summaries are copied to profiles/ by hand (profiles/README.md).
export TMPDIR=/tmp; ROOT=$PWD; OUT=$ROOT/gpurun_out/r06; mkdir -p $OUT
md5sum ik_llama.cpp_amd/libggml-hip-cdna4.so > $OUT/lib.md5
timeout 1500 python $ROOT/bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_n1.json 2> $OUT/bench_n1.err; echo bench rc=$?
cp gpurun_out/bench_details.json $OUT/bench_details.json 2>/dev/null
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench -o bench -- python $ROOT/bench.py --no-cpu-baseline --no-llama-bench --no-pmc --no-extra-configs > $OUT/bench_stats_stdout.json 2> $OUT/bench_stats_stderr.txt; echo bench-stats rc=$?
M=/tmp/llama3-8b-synth-q4km-32.gguf; [ -f $M ] || python $ROOT/tests/gguf_synth.py $M 32 > /dev/null
GGML_CDNA4_PARAMS=graphs=0 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/llama -o lb -- $ROOT/oracle/_ref/llama/bin/llama-bench -m $M -p 512 -n 128 -ngl 99 -fa 1 -t 8 -r 3 > $OUT/llama_stdout.txt 2> $OUT/llama_stderr.txt; echo llama rc=$?
cd $ROOT
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*counter_collection.csv" -delete; find $OUT -name "*agent_info.csv" -delete
find $OUT -type f | head -40; du -sh $OUT
cut -c1-1200 $OUT/bench_n1.json
;;
suite)
timeout 3000 python -m pytest tests/ -q -m gpu -x -p no:cacheprovider 2>&1 | tail -15
;;
bench)
timeout 900 python bench.py "$@" 2> gpurun_out/r06_bench.err | tee gpurun_out/r06_bench.json | cut -c1-600; tail -5 gpurun_out/r06_bench.err
;;
*) echo "steps: newtests pp_test forms forms_libs pp_pmc lb mtl suite bench" ;;
esac
