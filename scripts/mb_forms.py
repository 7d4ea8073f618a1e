#!/usr/bin/env python3
"""Prompt-GEMM forms side by side in ONE process, interleaved (guide rule 24): per-wave de-quantizing kernel (form 0), workgroup-shared weight tile (2), ping-pong (3),
the large-batch route = f16 weight image + f16 x f16 GEMM (4), the default dispatch (1).  Every op = f32 -> f16 activation image + GEMM, HIP events over NL distinct weight sets (no L2 reuse of weights between launches).
    python scripts/mb_forms.py [--types 12,14] [--ns 512,4096] [--rounds 3] [--forms 0,2,3] [--shapes fused,up,down,wo]
Prints the median and the best round per (shape, N, form) and checks form 3 == form 0 bit for bit where both are unsplit launches."""
import argparse
import os
import statistics
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from __graft_entry__ import _load_package  # noqa: E402

NAMES = {12: "Q4_K", 13: "Q5_K", 14: "Q6_K", 20: "IQ4_NL", 22: "IQ2_S", 21: "IQ3_S"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--types", default="12,14"); ap.add_argument("--ns", default="512,2048,4096"); ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--forms", default="0,2,3"); ap.add_argument("--shapes", default="fused,up,down,wo"); ap.add_argument("--nl", type=int, default=4)
    args = ap.parse_args()
    dev = torch.device("cuda", 0); torch.cuda.set_device(0)
    pkg = _load_package(); be = pkg.Cdna4Backend(0)
    gen = torch.Generator(device=dev); gen.manual_seed(1)
    forms = [int(f) for f in args.forms.split(",")]; NL = args.nl
    be.reserve_workspace(768 << 20)
    shapes = {"fused": (14336, 4096, True), "up": (14336, 4096, False), "down": (4096, 14336, False), "wo": (4096, 4096, False), "shard": (3584, 8192, True)}
    for t in [int(x) for x in args.types.split(",")]:
        for sname in args.shapes.split(","):
            m, k, fused = shapes[sname]
            ws = [bench.synth_weights(t, m, k, gen, dev) for _ in range(NL * (2 if fused else 1))]
            for n in [int(x) for x in args.ns.split(",")]:
                x = torch.randn((n, k), device=dev, generator=gen); out = torch.empty((n, m), device=dev)

                def run():
                    if fused:
                        for i in range(NL):
                            be.fused_up_gate(t, ws[2 * i], ws[2 * i + 1], x, out=out)
                    else:
                        for w in ws:
                            be.mul_mat(t, w, x, out=out)
                res = {f: [] for f in forms}; info = {}; ref = None; same = {}
                for f in forms:                      # warm-up + bit comparison against the first form listed
                    be.set_gemm_form(f); run(); torch.cuda.synchronize(); info[f] = be.last_launch_info()
                    o = out.clone()
                    if ref is None:
                        ref = o
                    else:
                        same[f] = bool(torch.equal(ref, o)) or ("maxdiff %.3g" % float((ref - o).abs().max() / ref.abs().max()))
                for _ in range(args.rounds):
                    for f in forms:
                        be.set_gemm_form(f)
                        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        e0.record(); run(); e1.record(); torch.cuda.synchronize()
                        res[f].append(e0.elapsed_time(e1) * 1e3 / NL)
                fl = (4.0 if fused else 2.0) * m * k * n
                for f in forms:
                    med, best = statistics.median(res[f]), min(res[f])
                    print("%-6s %-6s %5dx%-5d N=%4d form %d  median %8.1f us (%.3f)  best %8.1f us (%.3f)  %s %s%s" % (
                        NAMES.get(t, t), sname, m, k, n, f, med, fl / med * 1e-6 / 2500, best, fl / best * 1e-6 / 2500, info[f].get("kernel"), info[f].get("grid"),
                        ("  vs form %d: %s" % (forms[0], "bit-identical" if same[f] is True else same[f])) if f in same else ""), flush=True)
            del ws
    be.set_gemm_form(1); be.close()


if __name__ == "__main__":
    main()
