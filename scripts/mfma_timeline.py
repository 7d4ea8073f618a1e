#!/usr/bin/env python3
"""Phase timeline of the per-wave prompt GEMM (gemm_mfma_kernel) on the launches of a 512-token pass -- a -DGEMM_EXP_TIMELINE build of ONE translation unit:
    python scripts/pp_exp.py mtl12=-DGEMM_EXP_TIMELINE [--tus=gemm_12]         CDNA4_LIB=ik_llama.cpp_amd/exp/lib_mtl12.so python scripts/mfma_timeline.py [type] [M:K:N ...]
Every workgroup's wave 0 of each K-group stamps the 100 MHz wall clock at: entry, every K tile's barrier, end of the K loop, after the in-workgroup K-half reduction, after the
split-K slab stores have drained, after the ticket, after the result stores.  Printed: per phase the median / max over workgroups, in us from the launch's first entry stamp."""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from __graft_entry__ import _load_package  # noqa: E402


def main():
    t = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    shapes = [tuple(int(v) for v in s.split(":")) for s in sys.argv[2:]] or [(4096, 4096, 512), (4096, 14336, 512), (5120, 4096, 512)]
    dev = torch.device("cuda", 0); torch.cuda.set_device(0)
    pkg = _load_package(); be = pkg.Cdna4Backend(0)
    raw = C.CDLL(os.environ["CDNA4_LIB"])
    gen = torch.Generator(device=dev); gen.manual_seed(1)
    be.reserve_workspace(256 << 20); be.set_gemm_form(0)
    for m, k, n in shapes:
        ws = [bench.synth_weights(t, m, k, gen, dev) for _ in range(4)]
        x = torch.randn((n, k), device=dev, generator=gen); out = torch.empty((n, m), device=dev)
        for w in ws:
            be.mul_mat(t, w, x, out=out)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for w in ws:
            be.mul_mat(t, w, x, out=out)
        e1.record(); torch.cuda.synchronize()
        print("== type %d  %d x %d, %d tokens: %.1f us per op (image + GEMM)  %s" % (t, m, k, n, e0.elapsed_time(e1) * 250, be.last_launch_info()))
        assert raw.cdna4_exp_mfma_timeline_clear() == 0
        be.mul_mat(t, ws[0], x, out=out); torch.cuda.synchronize()
        buf = np.zeros((2048, 2, 32), np.uint64)
        assert raw.cdna4_exp_mfma_timeline(buf.ctypes.data_as(C.c_void_p)) == 0
        b = buf.astype(np.int64); live = b[:, :, 0] > 0
        t0 = b[:, :, 0][live].min()
        us = lambda a: (a - t0) / 100.0
        print("   workgroups x K-groups recorded: %d" % int(live.sum()))
        names = {0: "entry", 8: "first tile landed", 1: "K loop done", 2: "K-half reduction done", 3: "slab stores drained", 4: "ticket drawn", 5: "results stored (last arriver / unsplit)"}
        for i in (0, 8, 1, 2, 3, 4, 5):
            v = b[:, :, i][live & (b[:, :, i] > 0)]
            if v.size:
                v = us(v); print("   %-42s n=%4d  min %6.2f  median %6.2f  p90 %6.2f  max %6.2f us" % (names[i], v.size, v.min(), np.median(v), np.percentile(v, 90), v.max()))
        tiles = b[:, :, 8:32]; nt = int((tiles[live][0] > 0).sum())
        if nt > 1:
            d = np.diff(tiles[live][:, :nt], axis=1) / 100.0
            print("   K tiles per K-group: %d; barrier-to-barrier per tile: median %.2f us, mean %.2f, p90 %.2f; first three %s, last three %s" % (
                nt, np.median(d), d.mean(), np.percentile(d, 90), np.round(np.median(d[:, :3], axis=0), 2), np.round(np.median(d[:, -3:], axis=0), 2)))
        dur = b[:, :, 1][live] - b[:, :, 0][live]
        print("   entry -> K loop done per workgroup: median %.2f us" % (np.median(dur) / 100.0))
        del ws
    be.close()


if __name__ == "__main__":
    main()
