#!/usr/bin/env python3
"""Prompt GEMM microbench: the Llama-3-8B shapes (and the Mixtral grouped launch) at N = 512 / 4096, HIP-event timed over distinct weights.
    python scripts/mb_prefill.py [tag]      one line per (kernel, N): us per launch, TFLOP/s, fraction of 2.5 PF"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from __graft_entry__ import _load_package  # noqa: E402

Q4_K, Q6_K = 12, 14


def timed(fn, n_items, reps=3):
    fn(); torch.cuda.synchronize()
    best = 1e30
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / n_items)
    return best


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else ""
    dev = torch.device("cuda", 0); torch.cuda.set_device(0)
    pkg = _load_package(); be = pkg.Cdna4Backend(0)
    gen = torch.Generator(device=dev); gen.manual_seed(1)
    NL = 6
    W = {}
    for name, t, m, k in (("wo", Q4_K, 4096, 4096), ("kv", Q4_K, 1024, 4096), ("up", Q4_K, 14336, 4096), ("gate", Q4_K, 14336, 4096), ("down4", Q4_K, 4096, 14336), ("down6", Q6_K, 4096, 14336)):
        W[name] = (t, [bench.synth_weights(t, m, k, gen, dev) for _ in range(NL)], m, k)
    be.reserve_workspace(4096 * 14336 * 2 + (128 << 20))
    only = os.environ.get("MB_ONLY_N")
    for n in ((int(only),) if only else (512, 4096)):
        x4 = torch.randn((n, 4096), device=dev, generator=gen); x14 = torch.randn((n, 14336), device=dev, generator=gen)
        for name in ("wo", "kv", "up", "down4", "down6"):
            t, ws, m, k = W[name]; x = x14 if k == 14336 else x4; out = torch.empty((n, m), device=dev)
            us = timed(lambda: [be.mul_mat(t, w, x, out=out) for w in ws], NL)
            fl = 2.0 * m * k * n
            print("%s %-22s N=%4d  %8.1f us  %6.1f TF  %.3f" % (tag, "%s %dx%d %s" % (name, m, k, "Q4_K" if t == Q4_K else "Q6_K"), n, us, fl / us * 1e-6, fl / us * 1e-6 / 2500), flush=True)
        t, up, m, k = W["up"]; gate = W["gate"][1]; out = torch.empty((n, m), device=dev)
        us = timed(lambda: [be.fused_up_gate(t, u, g, x4, out=out) for u, g in zip(up, gate)], NL)
        fl = 4.0 * m * k * n
        print("%s %-22s N=%4d  %8.1f us  %6.1f TF  %.3f" % (tag, "fused up*gate Q4_K", n, us, fl / us * 1e-6, fl / us * 1e-6 / 2500), flush=True)
    if only:
        be.close(); return
    # Mixtral grouped launch: 8 experts, top-2, 512 tokens (bench.py c5)
    n, E, U = 512, 8, 2
    ups = [torch.stack([bench.synth_weights(Q4_K, 14336, 4096, gen, dev) for _ in range(E)]) for _ in range(2)]
    gates = [torch.stack([bench.synth_weights(Q4_K, 14336, 4096, gen, dev) for _ in range(E)]) for _ in range(2)]
    downs = [torch.stack([bench.synth_weights(Q4_K, 4096, 14336, gen, dev) for _ in range(E)]) for _ in range(2)]
    x3 = torch.randn((n, 1, 4096), device=dev, generator=gen)
    ids = torch.stack([torch.randperm(E, device=dev, generator=gen)[:U] for _ in range(n)]).to(torch.int32).contiguous()
    ffn = torch.empty((n, U, 14336), device=dev); dn = torch.empty((n, U, 4096), device=dev)
    be.reserve_workspace((n * U + 512) * 14336 * 2 + (64 << 20))
    us = timed(lambda: [be.moe_fused_up_gate(Q4_K, u, g, x3, ids, out=ffn) for u, g in zip(ups, gates)], 2)
    fl = 4.0 * 14336 * 4096 * n * U
    print("%s %-22s N=%4d  %8.1f us  %6.1f TF  %.3f" % (tag, "moe fused up*gate", n, us, fl / us * 1e-6, fl / us * 1e-6 / 2500), flush=True)
    us = timed(lambda: [be.mul_mat_id(Q4_K, d, ffn, ids, out=dn) for d in downs], 2)
    fl = 2.0 * 14336 * 4096 * n * U
    print("%s %-22s N=%4d  %8.1f us  %6.1f TF  %.3f" % (tag, "moe down (mul_mat_id)", n, us, fl / us * 1e-6, fl / us * 1e-6 / 2500), flush=True)
    be.close()


if __name__ == "__main__":
    main()
