#!/usr/bin/env python3
"""Trellis types on the device: decode (N = 1) and prompt (N = 512, the f16 route) times on 14336 x 4096, rotating weight buffers (cold caches).
    python scripts/mb_kt.py"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "scripts"))
import torch
from __graft_entry__ import _load_package
from oracle import bindings as ob
from microbench import rot_weights
be = _load_package().Cdna4Backend(0)
m, k = 14336, 4096
for t in ob.KT_TYPES + [ob.IQ2_S, ob.Q4_K]:
    ws = rot_weights(t, m, k, 512 << 20)
    row = []
    for n in (1, 4, 512):
        x = torch.randn(n, k, device="cuda"); out = torch.empty(n, m, device="cuda")
        for i in range(3): be.mul_mat(t, ws[i % len(ws)], x, out=out)
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        reps = 20 if n < 512 else 5
        e0.record()
        for i in range(reps): be.mul_mat(t, ws[i % len(ws)], x, out=out)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / reps
        mb = m * ob.row_size(t, k) / 1e6
        row.append("N=%d %.1f us (%s)" % (n, us, "%.2f TB/s" % (mb / us) if n < 512 else "%.0f TFLOP/s" % (2.0 * m * k * n / us / 1e6)))
    print("%-8s %s" % (ob.NAMES[t], " | ".join(row)), flush=True)
