#!/bin/bash
# The GPU calls of round 5, one case per measurement (each one `gpurun` call).   gpurun --timeout 900 -- 'bash scripts/r05_gpu.sh <step> [args]'
cd "$(dirname "$0")/.."; mkdir -p gpurun_out
STEP=${1:-help}; shift
case "$STEP" in
wlds_test)
# parity of the workgroup-shared weight tile prompt GEMM: bit-identity with the per-wave kernel, the 2048 / 4096-token cases in both forms
timeout 800 python -m pytest tests/test_gpu_prefill.py -q -m gpu -x -p no:cacheprovider -k "4k_tokens or shared_weight_tile" 2>&1 | tail -15
;;
wlds_perf)
# the two prompt-GEMM forms on the Llama-3-8B shapes, separate processes on one box, interleaved twice (op = f16 activation image + GEMM, HIP events)
OPS="--op upgate:12:14336:4096:4096 --op upgate:12:14336:4096:512 --op upgate:14:14336:4096:4096 --op upgate:20:14336:4096:4096"
CASES="--case 12:14336:4096:4096 --case 12:4096:14336:4096 --case 14:4096:14336:4096 --case 12:4096:4096:4096"
for rep in 1 2; do for f in 0 1; do
  echo "== CDNA4_GEMM_WLDS=$f (rep $rep)"
  CDNA4_GEMM_WLDS=$f timeout 200 python scripts/nt_bench.py $OPS $CASES --iters 30 --warmup 5 2>&1 | python -c "
import sys, json
for ln in sys.stdin:
    try: r = json.loads(ln); print('   %-32s %-8s %9.1f us  %7.1f TF  %.4f' % (r.get('op', r.get('case', '?')), r.get('type', ''), r['us'], r.get('tflops', 0), r.get('frac_mfma', r.get('frac', 0))))
    except Exception: print(ln.rstrip()[:200])"
done; done
;;
wlds_ab)
# variant libraries of the shared-weight-tile GEMM (scripts/wlds_exp.py) against the in-tree one, interleaved in ONE process:  r05_gpu.sh wlds_ab <name> [<name> ...]
LIBS="--lib ik_llama.cpp_amd/libggml-hip-cdna4.so"; for v in "$@"; do LIBS="$LIBS --lib ik_llama.cpp_amd/exp/lib_wlds_$v.so"; done
OPS="--op upgate:12:14336:4096:4096 --op upgate:12:14336:4096:512 --op upgate:14:14336:4096:4096 --op upgate:20:14336:4096:4096"
CASES="--case 12:14336:4096:4096 --case 12:4096:14336:4096 --case 14:4096:14336:4096"
timeout 600 python scripts/nt_bench.py $LIBS $OPS $CASES --iters 30 --warmup 5 --rounds 3 2>&1 | python -c "
import sys, json
for ln in sys.stdin:
    try: r = json.loads(ln); print('   %-32s %-8s %-44s %9.1f us  %7.1f TF  %.4f' % (r.get('op', r.get('case', '?')), r.get('type', ''), r.get('lib', ''), r['us'], r.get('tflops', 0), r.get('frac_mfma', r.get('frac', 0))))
    except Exception: print(ln.rstrip()[:200])"
;;
wlds_pmc)
# SQ counters of the fused up*gate prompt GEMM at 4096 tokens (torch-free driver), two passes of 8 counters; `wlds_pmc 0` = the per-wave kernel for comparison
ROOT=$PWD; OUT=$ROOT/gpurun_out/r05_pmc_$1; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
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
tgm)
# llama-bench tg128 through the shim: several environment variants against the default, interleaved on one box:  r05_gpu.sh tgm VAR=1 "A=1 B=2" ...
M=/tmp/llama3-8b-synth-q4km-32.gguf
[ -f $M ] || python tests/gguf_synth.py $M 32 > /dev/null || exit 1
one() { env $1 timeout 300 oracle/_ref/llama/bin/llama-bench -m $M -p 0 -n 128 -ngl 99 -fa 1 -t 8 -r 5 -o json 2>/dev/null | python -c "import json,sys; x=json.load(sys.stdin)[0]; print('  tg128 %.1f +- %.1f' % (x['avg_ts'], x['stddev_ts']))"; }
for i in 1 2; do echo "default"; one A=1; for v in "$@"; do echo "$v"; one "$v"; done; done
;;
final)
# Round-5 evidence run: the driver's bench command, rocprofv3 kernel stats of bench.py (headline config) and of llama-bench through the shim (graphs off: one row per kernel).
# Summaries are copied to profiles/ by hand (profiles/README.md).
export TMPDIR=/tmp; ROOT=$PWD; OUT=$ROOT/gpurun_out/r05; mkdir -p $OUT
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
tail -c 5200 $OUT/bench_n1.json
;;
c1prof)
# kernel stats of a decoded token of the Qwen3-0.6B-shaped IQ4_NL model (BASELINE configs[0]) through the shim, graphs off: one row per kernel
export TMPDIR=/tmp; ROOT=$PWD; OUT=$ROOT/gpurun_out/r05_c1; mkdir -p $OUT
M=/tmp/qwen3-06b-synth.gguf; [ -f $M ] || python -c "
import sys; sys.path.insert(0, '$ROOT/tests'); import gguf_synth as gs; gs.qwen3_06b_model('$M')"
if [ "$1" = 8b ]; then M=/tmp/llama3-8b-synth-q4km-32.gguf; [ -f $M ] || python $ROOT/tests/gguf_synth.py $M 32 > /dev/null; OUT=$ROOT/gpurun_out/r05_8b; mkdir -p $OUT; fi
if [ "$1" = moe ]; then M=/tmp/mixtral-synth-L4.gguf; [ -f $M ] || python -c "
import sys; sys.path.insert(0, '$ROOT/tests'); import gguf_synth as gs; gs.bench_model('$M', n_embd=4096, n_ff=14336, n_head=32, n_head_kv=8, n_layer=4, n_vocab=32000, n_expert=8, n_used=2, seed=5, name='Mixtral-8x7B-synth')"; OUT=$ROOT/gpurun_out/r05_moe; mkdir -p $OUT; fi
cd /tmp
GGML_CDNA4_STATS=1 timeout 300 $ROOT/oracle/_ref/llama/bin/llama-bench -m $M -p 128 -n 32 -ngl 99 -fa 1 -t 8 -r 5 -o json 2> $OUT/lb.err | python -c "import json,sys; [print('  p%d n%d %.1f +- %.1f' % (x['n_prompt'], x['n_gen'], x['avg_ts'], x['stddev_ts'])) for x in json.load(sys.stdin)]"
grep "cdna4\[" $OUT/lb.err | tail -4
GGML_CDNA4_PARAMS=graphs=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/p -o lb -- $ROOT/oracle/_ref/llama/bin/llama-bench -m $M -p 0 -n 32 -ngl 99 -fa 1 -t 8 -r 2 > /dev/null 2>&1
cd $ROOT; find $OUT -name "*kernel_trace.csv" -delete
python - <<PY
import csv, glob
for f in glob.glob("$OUT/**/*kernel_stats.csv", recursive=True):
    rows = list(csv.DictReader(open(f))); tot = sum(float(r["TotalDurationNs"]) for r in rows); ntok = 65.0
    print("decode kernels total ms", tot / 1e6, " per token us", tot / 1e3 / ntok, " launches per token", sum(int(r["Calls"]) for r in rows) / ntok)
    for r in rows[:24]: print("%6.2f%% %8.1f calls/token %8.2f us  %s" % (100 * float(r["TotalDurationNs"]) / tot, int(r["Calls"]) / ntok, float(r["AverageNs"]) / 1e3, r["Name"][:110]))
PY
;;
fa)
# the per-head decode attention: phase stamps of one launch (probe), the attention tests, then tg128 of the 8B model and of the Qwen3-0.6B shape
P=scripts/probes/bin/fa_timeline_probe; for a in "100 256 4" "30 256 4" "600 768 8"; do timeout 60 $P $a | head -1; done; timeout 60 $P 100 256 4 | tail -11
timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_attn_fused.py -q -x -k "flash or attn or attention" 2>&1 | tail -3
M=/tmp/llama3-8b-synth-q4km-32.gguf; [ -f $M ] || python tests/gguf_synth.py $M 32 > /dev/null || exit 1
Q=/tmp/qwen3-06b-synth.gguf; [ -f $Q ] || python -c "
import sys; sys.path.insert(0, 'tests'); import gguf_synth as gs; gs.qwen3_06b_model('$Q')"
one() { timeout 300 oracle/_ref/llama/bin/llama-bench -m $1 -p 0 -n $2 -ngl 99 -fa 1 -t 8 -r 5 -o json 2>/dev/null | python -c "import json,sys; x=json.load(sys.stdin)[0]; print('  $1 tg$2 %.1f +- %.1f' % (x['avg_ts'], x['stddev_ts']))"; }
for i in 1 2; do one $M 128; one $Q 128; one $Q 32; done; one $M 512
;;
deepprof)
# kernel stats of decode at depth 8192 (8B model): what a token's time is made of there
export TMPDIR=/tmp; ROOT=$PWD; OUT=$ROOT/gpurun_out/r05_deep; mkdir -p $OUT
M=/tmp/llama3-8b-synth-q4km-32.gguf; [ -f $M ] || python $ROOT/tests/gguf_synth.py $M 32 > /dev/null
cd /tmp
for v in 1 0; do
GGML_CDNA4_PARAMS=graphs=0 CDNA4_FA_SPLIT_MFMA=$v timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/p$v -o lb -- $ROOT/oracle/_ref/llama/bin/llama-bench -m $M -p 0 -n 0 -gp 8192,16 -ngl 99 -fa 1 -t 8 -r 1 > /dev/null 2>&1
done
cd $ROOT; find $OUT -name "*kernel_trace.csv" -delete
python - <<PY
import csv, glob
for v in (1, 0):
    for f in glob.glob("$OUT/p%d/**/*kernel_stats.csv" % v, recursive=True):
        rows = list(csv.DictReader(open(f)))
        print("CDNA4_FA_SPLIT_MFMA=%d" % v)
        for r in rows:
            if "attn" in r["Name"] or "copyBuffer" in r["Name"] or "fill" in r["Name"]: print("  %8d calls %9.2f us  %s" % (int(r["Calls"]), float(r["AverageNs"]) / 1e3, r["Name"][:100]))
PY
;;
fadeep)
# decode at depth (split-KV attention): llama-bench -gp <depth>,32 of the 8B model; environment variants interleaved:  r05_gpu.sh fadeep "CDNA4_FA_SPLIT_MFMA=0" ...
M=/tmp/llama3-8b-synth-q4km-32.gguf; [ -f $M ] || python tests/gguf_synth.py $M 32 > /dev/null || exit 1
one() { env $1 timeout 600 oracle/_ref/llama/bin/llama-bench -m $M -p 0 -n 0 -gp 512,32 -gp 2048,32 -gp 8192,32 -ngl 99 -fa 1 -t 8 -r 3 -o json 2>/dev/null | python -c "import json,sys; print('  %-28s' % '$1', '  '.join('depth %d: %.1f' % (x['n_prompt'], x['avg_ts']) for x in json.load(sys.stdin)))"; }
for i in 1 2; do one A=1; for v in "$@"; do one "$v"; done; done
;;
ppab)
# prompt passes through llama-bench (8B model): two library builds interleaved:  r05_gpu.sh ppab <before.so>
L=ik_llama.cpp_amd/libggml-hip-cdna4.so; V=$1; cp $L /tmp/new.so
M=/tmp/llama3-8b-synth-q4km-32.gguf; [ -f $M ] || python tests/gguf_synth.py $M 32 > /dev/null || exit 1
one() { timeout 600 oracle/_ref/llama/bin/llama-bench -m $M -p 512,2048,8192 -n 0 -ngl 99 -fa 1 -t 8 -r 3 -o json 2>/dev/null | python -c "import json,sys; print('  %-8s' % '$1', '  '.join('pp%d: %.0f' % (x['n_prompt'], x['avg_ts']) for x in json.load(sys.stdin)))"; }
for i in 1 2; do cp $V $L; one before; cp /tmp/new.so $L; one after; done
;;
libab)
# A/B of two builds of the library through llama-bench tg128 (8B) and tg32 (Qwen3-0.6B shape), interleaved on one box:  r05_gpu.sh libab <variant.so>
L=ik_llama.cpp_amd/libggml-hip-cdna4.so; V=$1; cp $L /tmp/base.so
M=/tmp/llama3-8b-synth-q4km-32.gguf; [ -f $M ] || python tests/gguf_synth.py $M 32 > /dev/null || exit 1
Q=/tmp/qwen3-06b-synth.gguf; [ -f $Q ] || python -c "
import sys; sys.path.insert(0, 'tests'); import gguf_synth as gs; gs.qwen3_06b_model('$Q')"
one() { timeout 300 oracle/_ref/llama/bin/llama-bench -m $2 -p 0 -n $3 -ngl 99 -fa 1 -t 8 -r 5 -o json 2>/dev/null | python -c "import json,sys; x=json.load(sys.stdin)[0]; print('  $1 tg$3 %.1f +- %.1f' % (x['avg_ts'], x['stddev_ts']))"; }
for i in 1 2 3; do cp /tmp/base.so $L; one base $M 128; one base $Q 32; cp $V $L; one variant $M 128; one variant $Q 32; done
cp /tmp/base.so $L
;;
qknorm)
# the q / k norm + ROPE + KV store launch: C-ABI bit-for-bit test, the shim cases, then the small model's token
timeout 900 python -m pytest tests/test_gpu_qk_norm_rope.py tests/test_gpu_ops.py -q -x -k "norm_rope or per_head or rope" 2>&1 | tail -8
bash $0 c1prof
;;
*) echo "steps: wlds_test wlds_perf wlds_ab wlds_pmc tgm final c1prof qknorm fa fadeep deepprof ppab libab";;
esac
