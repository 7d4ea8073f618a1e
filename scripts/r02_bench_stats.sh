#!/bin/bash
# rocprofv3 kernel stats of the default bench command restricted to the headline config (c2), for profiles/r02_bench_kernel_stats.csv
cd "$(dirname "$0")/.."; ROOT=$PWD
export TMPDIR=/tmp; OUT=$ROOT/gpurun_out/r02b; mkdir -p $OUT; cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o bench -- python $ROOT/bench.py --no-cpu-baseline --no-llama-bench --no-pmc --no-extra-configs > $OUT/bench_stdout.json 2> $OUT/bench_stderr.txt; echo rc=$?
find $OUT -name "*kernel_trace.csv" -delete
head -8 $OUT/bench_kernel_stats.csv | cut -c1-150; cat $OUT/bench_stdout.json | head -c 600
