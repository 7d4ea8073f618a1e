#!/bin/bash
# Round-3 evidence run (GPU box): the driver's bench command, rocprofv3 kernel stats of bench.py (headline config) and of llama-bench, SQ counters of the prefill GEMM,
# of the grouped MoE GEMM and of the IQ2_S fused decode launch.  Summaries are copied to profiles/ by hand (profiles/README.md).
cd "$(dirname "$0")/.."; ROOT=$PWD
export TMPDIR=/tmp; OUT=$ROOT/gpurun_out/r03; mkdir -p $OUT
md5sum ik_llama.cpp_amd/libggml-hip-cdna4.so > $OUT/lib.md5
timeout 900 python $ROOT/bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_n1.json 2> $OUT/bench_n1.err; echo bench rc=$?
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
# IQ2_S fused decode launch + the grouped MoE GEMM: the same two counter sets
MB_ONLY_IQ=1 timeout 300 rocprofv3 --pmc $A --kernel-trace --output-format csv -d $OUT/pmc_iq_a -o p -- python $ROOT/scripts/iq_exp.py one base > /dev/null 2>&1; echo pmcIQa rc=$?
MB_ONLY_IQ=1 timeout 300 rocprofv3 --pmc $B --kernel-trace --output-format csv -d $OUT/pmc_iq_b -o p -- python $ROOT/scripts/iq_exp.py one base > /dev/null 2>&1; echo pmcIQb rc=$?
python $ROOT/scripts/pmc_summary.py $OUT/pmc_iq_a/p_counter_collection.csv "rocprofv3 --pmc $A --kernel-trace -- python scripts/iq_exp.py one base" > $OUT/pmc_iq_a.json
python $ROOT/scripts/pmc_summary.py $OUT/pmc_iq_b/p_counter_collection.csv "rocprofv3 --pmc $B --kernel-trace -- python scripts/iq_exp.py one base" > $OUT/pmc_iq_b.json
MB_MOE_T=512 timeout 300 rocprofv3 --pmc $A --kernel-trace --output-format csv -d $OUT/pmc_moe_a -o p -- python $ROOT/scripts/microbench.py moe mixtral > /dev/null 2>&1; echo pmcMoEa rc=$?
python $ROOT/scripts/pmc_summary.py $OUT/pmc_moe_a/p_counter_collection.csv "rocprofv3 --pmc $A --kernel-trace -- MB_MOE_T=512 python scripts/microbench.py moe mixtral" > $OUT/pmc_moe_a.json
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*counter_collection.csv" -delete; find $OUT -name "*agent_info.csv" -delete
find $OUT -type f | head -40; du -sh $OUT
python - <<'PY'
import json
d=json.loads(open('/root/repo/gpurun_out/r03/bench_n1.json').read().strip().splitlines()[-1])
print(d['value'], d['roofline']['frac'], d['roofline']['avg_launch_us'], d['roofline']['decode_token']['frac'], d['roofline_prefill'])
for k,v in d.get('configs',{}).items(): print(k, v.get('value'), v.get('roofline',{}).get('frac'), v.get('roofline_prefill',{}).get('frac'), v.get('roofline_prefill',{}).get('kernel_only',{}).get('frac'))
print(d['llama_bench'].get('pp512_tok_s'), d['llama_bench'].get('tg128_tok_s'), d['cpu_baseline'].get('value'))
PY
