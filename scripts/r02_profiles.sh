#!/bin/bash
# Round-2 evidence run (GPU box): rocprofv3 kernel stats of bench.py and of llama-bench, SQ counters of the prefill GEMM and of the flash-attention kernel.
cd "$(dirname "$0")/.."; ROOT=$PWD
export TMPDIR=/tmp; OUT=$ROOT/gpurun_out/r02; mkdir -p $OUT
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench -o bench -- python $ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-llama-bench --no-pmc > $OUT/bench_stdout.json 2> $OUT/bench_stderr.txt; echo bench rc=$?
M=/tmp/llama3-8b-synth-q4km-32.gguf; [ -f $M ] || python $ROOT/tests/gguf_synth.py $M 32 > /dev/null
GGML_CDNA4_PARAMS=graphs=0 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/llama -o lb -- $ROOT/oracle/_ref/llama/bin/llama-bench -m $M -p 512 -n 128 -ngl 99 -fa 1 -t 8 -r 3 > $OUT/llama_stdout.txt 2> $OUT/llama_stderr.txt; echo llama rc=$?
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVES --kernel-trace --output-format csv -d $OUT/pmc_gemm_a -o p -- python $ROOT/scripts/gemm_prof.py 4096 > /dev/null 2>&1; echo pmcA rc=$?
timeout 300 rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_INSTS_VMEM_RD --kernel-trace --output-format csv -d $OUT/pmc_gemm_b -o p -- python $ROOT/scripts/gemm_prof.py 4096 > /dev/null 2>&1; echo pmcB rc=$?
python $ROOT/scripts/pmc_summary.py $OUT/pmc_gemm_a/p_counter_collection.csv "rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVES --kernel-trace -- python scripts/gemm_prof.py 4096" > $OUT/pmc_gemm_a.json
python $ROOT/scripts/pmc_summary.py $OUT/pmc_gemm_b/p_counter_collection.csv "rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_INSTS_VMEM_RD --kernel-trace -- python scripts/gemm_prof.py 4096" > $OUT/pmc_gemm_b.json
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*counter_collection.csv" -delete
find $OUT -type f | head -30; du -sh $OUT
