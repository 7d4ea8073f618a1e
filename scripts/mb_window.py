#!/usr/bin/env python3
"""Latency of the one-launch all-reduce over IPC-mapped windows (cdna4_window_*), message size x wire type, captured in a HIP graph and replayed:
    python scripts/mb_window.py [world]        (default: one rank per visible GPU, at least 2; on a 1-GPU box all ranks time-share device 0)
Each rank = one process (multiprocessing spawn); handles travel through queues.  Prints one line per (bytes, wire) from rank 0; correctness is
checked against the sum computed on the host.  Wrap in `timeout`: the window kernels wait for their peers with a bound, the script itself does not."""
import multiprocessing as mp
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")


def rank_main(rank, world, q_out, q_in, res):
    sys.path.insert(0, ROOT)
    import torch
    from __graft_entry__ import _load_package
    pkg = _load_package()
    dev = rank % torch.cuda.device_count(); torch.cuda.set_device(dev)
    be = pkg.Cdna4Backend(dev)
    q_out.put((rank, be.window_create(rank, world, 16 << 20)))
    for _ in range(world - 1):
        r, h = q_in.get(timeout=120); be.window_attach(r, h)
    lines = []
    for n, wire in ((4096, None), (8192, None), (32 * 4096, None), (512 * 4096, torch.bfloat16), (512 * 8192, torch.bfloat16), (512 * 4096, None)):
        parts = [torch.from_numpy(np.random.default_rng(n + r).standard_normal(n).astype(np.float32)) for r in range(world)]
        x = parts[rank].cuda(); y = torch.empty_like(x)
        be.window_reduce(y.copy_(x), check=True, wire=wire)
        want = sum((p.to(wire).float() if wire else p) for p in parts)
        ok = bool((y.cpu() - want).abs().max() <= 1e-5 * world * float(want.abs().max()))
        st = torch.cuda.Stream(); g = torch.cuda.CUDAGraph(); reps = 20
        with torch.cuda.graph(g, stream=st):
            for _ in range(reps):
                be.window_reduce(y, wire=wire)          # (y keeps growing: only the time matters here)
        y.copy_(x); g.replay(); torch.cuda.synchronize()
        t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
        y.copy_(x); t0.record()
        for _ in range(5):
            g.replay()
        t1.record(); torch.cuda.synchronize()
        us = t0.elapsed_time(t1) * 1e3 / (5 * reps)
        lines.append("%9d B on the wire (%s)  %7.2f us per reduce  %s" % (n * (2 if wire else 4), "bf16" if wire else "f32", us, "ok" if ok else "MISMATCH"))
    be.window_free(); be.close()
    res.put((rank, lines))


def main():
    import torch
    world = int(sys.argv[1]) if len(sys.argv) > 1 else max(2, torch.cuda.device_count())
    ctx = mp.get_context("spawn")
    qs = [ctx.Queue() for _ in range(world)]; out = ctx.Queue(); res = ctx.Queue()
    procs = [ctx.Process(target=rank_main, args=(r, world, out, qs[r], res)) for r in range(world)]
    [p.start() for p in procs]
    for r, h in [out.get(timeout=300) for _ in range(world)]:
        for o in range(world):
            if o != r:
                qs[o].put((r, h))
    results = dict(res.get(timeout=600) for _ in range(world))
    [p.join(60) for p in procs]
    print("world %d on %d device(s)" % (world, torch.cuda.device_count()))
    print("\n".join(results[0]))


if __name__ == "__main__":
    main()
