#!/usr/bin/env python3
"""The norm-carrying decode launches as dependent chains over distinct weights (HIP events): fused up*gate with ffn_norm in its prologue (FX = 1), Q4_K 2 x 14336 x 4096.
    [CDNA4_LIB=...] python scripts/mb_norm.py [tag]"""
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
    NL, t, m, k = 16, 12, 14336, 4096
    ws = [(bench.synth_weights(t, m, k, gen, dev), bench.synth_weights(t, m, k, gen, dev)) for _ in range(NL)]
    nw = [torch.rand(k, device=dev, generator=gen) + 0.5 for _ in range(NL)]
    x = torch.randn((1, k), device=dev, generator=gen); out = torch.empty((1, m), device=dev)
    for norm in (True, False):
        with be.record() as plan:
            for (u, g), w in zip(ws, nw):
                if norm: be.fused_up_gate_norm(t, u, g, x, w, 1e-5, out=out)
                else: be.fused_up_gate(t, u, g, x, out=out)
        plan.replay(be._check); torch.cuda.synchronize()
        best = []
        for _ in range(7):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(4): plan.replay(be._check)
            e1.record(); torch.cuda.synchronize()
            best.append(e0.elapsed_time(e1) * 1e3 / (4 * NL))
        best.sort()
        print("%s fused up*gate %s: best %.2f us, median %.2f us  %s" % (tag, "with norm (FX=1)" if norm else "plain           ", best[0], best[len(best) // 2], be.last_launch_info().get("fx")), flush=True)
    be.close()


if __name__ == "__main__":
    main()
