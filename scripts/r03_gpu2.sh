#!/bin/bash
# round-3 GPU call 2: the whole -m gpu suite (two-shot window reduce with 2 / 4 ranks, sliced peer reduce, row-meta K splits, router ties, capture-safe check) and the
# default bench (timed: the CPU legs are now bounded)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r3b; mkdir -p $O
export TMPDIR=/tmp
( time timeout 1200 python -m pytest tests -m gpu -x -q ) > $O/pytest_gpu.log 2>&1; echo "rc=$?" >> $O/pytest_gpu.log
( time timeout 900 python bench.py ) > $O/bench_default.json 2> $O/bench_default.err; echo "rc=$?" >> $O/bench_default.err
timeout 200 python scripts/mb_window.py 4 > $O/mb_window4.log 2>&1; echo "rc=$?" >> $O/mb_window4.log
