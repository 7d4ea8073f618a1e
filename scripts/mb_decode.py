#!/usr/bin/env python3
"""Decode mat-vec launches of a Llama-3-8B Q4_K_M layer as dependent chains over distinct weights (HIP events; one line per shape):  python scripts/mb_decode.py [tag]"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import bench  # noqa: E402
from __graft_entry__ import _load_package  # noqa: E402

def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else ""
    dev = torch.device("cuda", 0); torch.cuda.set_device(0)
    pkg = _load_package(); be = pkg.Cdna4Backend(0)
    gen = torch.Generator(device=dev); gen.manual_seed(1)
    NL = 24
    for name, t, m, k in (("down Q6_K 4096x14336", 14, 4096, 14336), ("down Q4_K 4096x14336", 12, 4096, 14336), ("wo Q4_K 4096x4096", 12, 4096, 4096), ("v Q6_K 1024x4096", 14, 1024, 4096),
                          ("output Q6_K 128256x4096", 14, 128256, 4096)):
        nl = 4 if m > 100000 else NL
        ws = [bench.synth_weights(t, m, k, gen, dev) for _ in range(nl)]
        x = torch.randn((1, k), device=dev, generator=gen); out = torch.empty((1, m), device=dev)
        with be.record() as plan:
            for w in ws:
                be.mul_mat(t, w, x, out=out)
        plan.replay(be._check); torch.cuda.synchronize()
        best = 1e9
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); plan.replay(be._check); plan.replay(be._check); e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) * 1e3 / (2 * nl))
        by = m * (k // bench.BLCK_SIZE[t]) * bench.TYPE_SIZE[t] + 4 * k + 4 * m
        print("%s %-26s %7.2f us  %.3f of 8 TB/s  %s" % (tag, name, best, by / (best * 1e-6) / 1e9 / 8000, be.last_launch_info().get("kernel")), flush=True)
        del ws
    be.close()

if __name__ == "__main__":
    main()
