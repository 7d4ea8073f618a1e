#!/usr/bin/env python3
"""Interval timeline of ONE workgroup of the ping-pong prompt GEMM (a -DGEMM_PP_TIMELINE build, scripts/pp_exp.py):  CDNA4_LIB=ik_llama.cpp_amd/exp/lib_<name>.so python scripts/pp_timeline.py
Per wave: cycles of issue work per interval kind (first / second load interval, the two matrix intervals; even and odd stages) and cycles spent between the end of the work and the
release of the closing barrier (LDS / DMA drain + waiting for the slowest wave), averaged over the recorded stages."""
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
    t = int(os.environ.get("PP_TYPE", 12)); n = int(os.environ.get("PP_N", 4096))
    dev = torch.device("cuda", 0); torch.cuda.set_device(0)
    pkg = _load_package(); be = pkg.Cdna4Backend(0)
    gen = torch.Generator(device=dev); gen.manual_seed(1)
    up = bench.synth_weights(t, 14336, 4096, gen, dev); gate = bench.synth_weights(t, 14336, 4096, gen, dev)
    x = torch.randn((n, 4096), device=dev, generator=gen); out = torch.empty((n, 14336), device=dev)
    be.reserve_workspace(n * 4096 * 2 + (64 << 20)); be.set_gemm_form(int(os.environ.get("PP_FORM", 3)))
    for _ in range(3):
        be.fused_up_gate(t, up, gate, x, out=out)
    torch.cuda.synchronize()
    print(be.last_launch_info())
    buf = np.zeros((8, 64, 8), np.uint64)
    raw = C.CDLL(os.environ["CDNA4_LIB"])
    assert raw.cdna4_exp_pp_timeline(buf.ctypes.data_as(C.c_void_p)) == 0
    t0 = int(buf[:, :, 0][buf[:, :, 0] > 0].min())
    names = ["L1e", "C1e", "L2e", "C2e", "L1o", "C1o", "L2o", "C2o"]
    for w in range(8):
        r = buf[w].astype(np.int64)                          # [interval][slot]: 0 start (after the barrier), 1..3 section ends, 4 end of work, 5 behind the PREVIOUS interval's waits
        off = 0 if w < 4 else 1                             # group 1 opens with an extra (empty) interval
        rows = {k: [] for k in names}
        for i in range(off + 8, 63):                        # (skip the first two stages)
            kind = names[(i - off) % 8]
            st, e1, e2, e3, ew = r[i, 0], r[i, 1], r[i, 2], r[i, 3], r[i, 4]
            waited, nxt = r[i + 1, 5], r[i + 1, 0]
            secs = [(e1 - st) if e1 else 0, (e2 - e1) if e2 and e1 else 0, (e3 - e2) if e3 and e2 else 0]
            last = e3 or e2 or e1 or st
            rows[kind].append(secs + [ew - last, ew - st, waited - ew, nxt - waited, nxt - st])
        print("wave %d (group %d), first stamp %d:   reads | dequant+store | dma+raw | rest || work | closing waits | barrier || interval" % (w, w >> 2, r[0, 0] - t0))
        for k in names:
            if rows[k]:
                m = np.mean(np.array(rows[k]), axis=0)
                print("   %s  %5.0f %5.0f %5.0f %5.0f || %5.0f %5.0f %5.0f || %5.0f" % ((k,) + tuple(m)))
        if w in (0, 4):
            i0, i1 = off + 8, off + 8 + 8 * 5
            print("   cycles per 64-wide stage: %.0f (the matrix pipe needs 2048)" % ((r[i1, 0] - r[i0, 0]) / 10.0))
    be.close()


if __name__ == "__main__":
    main()
