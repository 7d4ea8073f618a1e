#!/bin/bash
# decode attention: per-head kernel vs split-KV kernel (write-through hand-off vs the fenced one) at short contexts; torch-free
cd "$(dirname "$0")/.."; mkdir -p gpurun_out
OPS="--op fa:32:8:128 --op fa:32:8:256 --op fa:32:8:512 --op fa:32:8:768 --op fa:32:8:1024 --op fa:32:8:2048 --op fa:32:8:4096 --op fa:64:8:256"
echo "== per-head kernel (split from 100000 keys)"; CDNA4_FA_SPLIT_MIN_KV=100000 timeout 120 python scripts/nt_bench.py $OPS --check --iters 300 2>&1 | cut -c1-200
echo "== split kernel, write-through hand-off (split from 64 keys)"; CDNA4_FA_SPLIT_MIN_KV=64 timeout 120 python scripts/nt_bench.py $OPS --check --iters 300 --stress 60 2>&1 | cut -c1-260
echo "== split kernel, fenced hand-off (split from 64 keys)"; CDNA4_FA_SPLIT_MIN_KV=64 CDNA4_FA_SPLIT_FENCE=1 timeout 120 python scripts/nt_bench.py $OPS --check --iters 300 2>&1 | cut -c1-200
echo "== split kernel, write-through, 128-key chunks"; for s in 2 4; do CDNA4_FA_SPLIT_MIN_KV=64 CDNA4_FA_SPLITS=$s timeout 120 python scripts/nt_bench.py --op fa:32:8:256 --op fa:32:8:512 --check --iters 300 2>&1 | cut -c1-200; done
