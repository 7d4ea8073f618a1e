#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r3q; mkdir -p $O
export TMPDIR=/tmp
export HSA_ENABLE_IPC_MODE_LEGACY=0
CDNA4_BENCH_DEBUG_ONE_DEVICE=0 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29571 bench.py --gpus 2 --steps 3 --warmup 1 > $O/bench2.json 2> $O/bench2.err
echo "rc=$?"; tail -c 1500 $O/bench2.json; echo; tail -n 15 $O/bench2.err | cut -c1-300
