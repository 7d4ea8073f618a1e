#!/bin/bash
# round-3 GPU call 4: whole -m gpu suite, grouped-MoE / split-K microbench, default bench
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r3d; mkdir -p $O
export TMPDIR=/tmp
md5sum ik_llama.cpp_amd/libggml-hip-cdna4.so > $O/lib.md5
( time timeout 1500 python -m pytest tests -m gpu -q ) > $O/pytest_gpu.log 2>&1; echo "rc=$?" >> $O/pytest_gpu.log
timeout 300 python scripts/mb_prefill.py new > $O/mb_prefill_new.log 2>&1
MB_ONLY_N=512 CDNA4_DETERMINISTIC=1 timeout 200 python scripts/mb_prefill.py det > $O/mb_prefill_det.log 2>&1
( time timeout 900 python bench.py --steps 20 --warmup 5 ) > $O/bench.json 2> $O/bench.err; echo "rc=$?" >> $O/bench.err
