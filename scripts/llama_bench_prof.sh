#!/bin/bash
# rocprofv3 kernel stats of llama-bench (reference binary + shim) on the synthetic 8B model: scripts/llama_bench_prof.sh <tag> [llama-bench args]
cd "$(dirname "$0")/.."
TAG=${1:-prof}; shift
M=/tmp/llama3-8b-synth-q4km-32.gguf
[ -f $M ] || python tests/gguf_synth.py $M 32 > /dev/null || exit 1
export TMPDIR=/tmp; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o run -- $OLDPWD/oracle/_ref/llama/bin/llama-bench -m $M -ngl 99 -fa 1 -t 8 -r 2 "$@" > $OUT/stdout.txt 2> $OUT/stderr.txt
echo rc=$?; tail -4 $OUT/stdout.txt
python - <<PY
import csv, glob
for f in glob.glob("$OUT/**/*kernel_stats.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    print(f, "total ms", tot / 1e6)
    for r in rows[:22]:
        print("%6.2f%% %9d calls %9.2f us avg  %s" % (100 * float(r["TotalDurationNs"]) / tot, int(r["Calls"]), float(r["AverageNs"]) / 1e3, r["Name"][:110]))
PY
