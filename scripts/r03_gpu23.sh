#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r3w; mkdir -p $O
export TMPDIR=/tmp
for v in default w8 w8p1; do
  E=""; [ $v = w8 ] && E="CDNA4_GEMV_WAVES=8"; [ $v = w8p1 ] && E="CDNA4_GEMV_WAVES=8 CDNA4_GEMV_PER_CU=1"
  env $E timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-llama-bench --no-pmc --no-extra-configs > $O/bench_$v.json 2> $O/bench_$v.err
  python - $v <<'PY'
import json,sys
d=json.loads(open('gpurun_out/r3w/bench_%s.json'%sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], d['value'], d['config']['tg128_tok_s'], d['config']['pp512_tok_s'], d['roofline']['avg_launch_us'], d['roofline']['decode_token']['ms'])
PY
done
