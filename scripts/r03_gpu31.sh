#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; ROOT=$PWD
O=$ROOT/gpurun_out/r3E; mkdir -p $O
export TMPDIR=/tmp
for i in 1 2 3 4 5 6; do
  timeout 300 python -m pytest tests/test_gpu_llama.py -q -x -k "more_weight_types_vs_cpu and iqk" 2>&1 | grep -v "cluster\|RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | grep -E "passed|failed|AssertionError|assert |first|second|device_run" | cut -c1-700 >> $O/runs.log
  echo "--- run $i" >> $O/runs.log
done
cat $O/runs.log
