#!/usr/bin/env python3
"""Latency by message size of the three all-reduce paths of the one-process-per-GPU deployment, in ONE run (the first thing to run on a multi-GPU node):
    one-shot window reduce | two-shot window reduce (reduce-scatter + all-gather) | RCCL all-reduce (C-ABI communicator)
each captured in a HIP graph (20 reduces per graph) and replayed:
    python scripts/mb_window.py [world]        (default: one rank per visible GPU, at least 2; on a 1-GPU box all ranks time-share device 0 and RCCL is skipped)
Each rank = one process (multiprocessing spawn); window handles and the RCCL unique id travel through queues.  Rank 0 prints one table; the window results are checked against the
sum computed on the host.  The cross-over the library uses (CDNA4_WINDOW_TWO_SHOT_MIN, default 256 KiB, world > 2) should sit where the two window columns cross.
Wrap in `timeout`: the window kernels wait for their peers with a bound, the script itself does not."""
import multiprocessing as mp
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
SIZES = [(4096, None), (16384, None), (65536, None), (512 * 512, "bf16"), (512 * 2048, "bf16"), (512 * 4096, "bf16"), (512 * 8192, "bf16"), (512 * 4096, None)]     # (elements, wire)


def rank_main(rank, world, mode, q_out, q_in, res):
    sys.path.insert(0, ROOT)
    import torch
    from __graft_entry__ import _load_package
    pkg = _load_package()
    ndev = torch.cuda.device_count(); dev = rank % ndev; torch.cuda.set_device(dev)
    be = pkg.Cdna4Backend(dev)
    rows = {}
    if mode == "rccl":
        if rank == 0:
            uid = be.comm_unique_id()
            q_out.put((0, uid))
        else:
            q_out.put((rank, None))
        uid = q_in.get(timeout=120)
        be.comm_init(uid, rank, world)
    else:
        q_out.put((rank, be.window_create(rank, world, 16 << 20)))
        for _ in range(world - 1):
            r, h = q_in.get(timeout=120); be.window_attach(r, h)
    for n, wire_name in SIZES:
        wire = torch.bfloat16 if wire_name else None
        parts = [torch.from_numpy(np.random.default_rng(n + r).standard_normal(n).astype(np.float32)) for r in range(world)]
        x = parts[rank].cuda()
        if mode == "rccl":
            y = x.to(wire) if wire is not None else x.clone()
            red = lambda: be.reduce(y)
            red(); torch.cuda.synchronize()
            want = sum((p.to(wire).float() if wire is not None else p) for p in parts)
            ok = bool((y.float().cpu() - want).abs().max() <= 2e-2 * world * float(want.abs().max()))
        else:
            y = torch.empty_like(x)
            red = lambda: be.window_reduce(y, wire=wire)
            be.window_reduce(y.copy_(x), check=True, wire=wire)
            want = sum((p.to(wire).float() if wire is not None else p) for p in parts)
            ok = bool((y.cpu() - want).abs().max() <= (1e-5 if wire is None else 1e-2) * world * float(want.abs().max()))
        st = torch.cuda.Stream(); g = torch.cuda.CUDAGraph(); reps = 20
        try:
            with torch.cuda.graph(g, stream=st, capture_error_mode="thread_local"):
                for _ in range(reps):
                    red()          # (y keeps growing: only the time matters here)
            run = g.replay; per = reps
        except Exception:          # (a collective library that refuses capture: eager launches)
            run = red; per = 1
        run(); torch.cuda.synchronize()
        t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
        t0.record()
        for _ in range(5 if per > 1 else 100):
            run()
        t1.record(); torch.cuda.synchronize()
        rows[(n, wire_name)] = (t0.elapsed_time(t1) * 1e3 / ((5 if per > 1 else 100) * per), ok)
    if mode != "rccl":
        be.window_free()
    be.close()
    res.put((rank, rows))


def run_mode(world, mode, env):
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        ctx = mp.get_context("spawn")
        qs = [ctx.Queue() for _ in range(world)]; out = ctx.Queue(); res = ctx.Queue()
        procs = [ctx.Process(target=rank_main, args=(r, world, mode, out, qs[r], res)) for r in range(world)]
        [p.start() for p in procs]
        got = [out.get(timeout=300) for _ in range(world)]
        if mode == "rccl":
            uid = [h for r, h in got if r == 0][0]
            for q in qs:
                q.put(uid)
        else:
            for r, h in got:
                for o in range(world):
                    if o != r:
                        qs[o].put((r, h))
        results = dict(res.get(timeout=900) for _ in range(world))
        [p.join(60) for p in procs]
        return results[0]
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def main():
    import torch
    ndev = torch.cuda.device_count()
    world = int(sys.argv[1]) if len(sys.argv) > 1 else max(2, ndev)
    cols = {"one-shot": run_mode(world, "window", {"CDNA4_WINDOW_TWO_SHOT_MIN": str(1 << 40)}),
            "two-shot": run_mode(world, "window", {"CDNA4_WINDOW_TWO_SHOT_MIN": "0", "CDNA4_WINDOW_TWO_SHOT_ANY_WORLD": "1"})}
    if ndev >= world:
        try:
            cols["rccl"] = run_mode(world, "rccl", {})
        except Exception as e:      # noqa: BLE001
            print("rccl column unavailable: %r" % (e,))
    print("world %d on %d device(s); us per all-reduce (captured, replayed)" % (world, ndev))
    print("%12s %5s " % ("wire bytes", "wire") + " ".join("%12s" % c for c in cols))
    for n, w in SIZES:
        cells = []
        for c in cols:
            us, ok = cols[c][(n, w)]
            cells.append("%9.2f %s" % (us, "ok" if ok else "!!"))
        print("%12d %5s " % (n * (2 if w else 4), w or "f32") + " ".join("%12s" % x for x in cells))


if __name__ == "__main__":
    main()
