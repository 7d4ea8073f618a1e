#!/usr/bin/env python3
"""Decode GEMV and prompt GEMM of the SURVEY 8 f3 weight types on Llama-3-8B shapes (scripts/mb_legacy.py [type names...])"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "scripts"))
from __graft_entry__ import _load_package
from oracle import bindings as ob
from microbench import rot_weights
be = _load_package().Cdna4Backend(0)
want = sys.argv[1:]
for t in [ob.IQ4_NL] + ob.LEGACY_TYPES:
    if want and ob.NAMES[t] not in want: continue
    for (m, k) in ((14336, 4096), (4096, 14336)):
        ws = rot_weights(t, m, k)
        line = []
        for n in (1, 4, 512):
            x = torch.randn(n, k, device="cuda"); out = torch.empty(n, m, device="cuda")
            ms = be.time_mul_mat(t, ws, x, out, warmup=5, iters=30)
            by = m * ob.row_size(t, k)
            line.append("N=%d %7.2f us (%s)" % (n, ms * 1e3, "%.2f TB/s" % (by / ms / 1e9) if n <= 8 else "%.0f TF" % (2.0 * m * k * n / ms / 1e9)))
        print("%-6s %5dx%-5d  %s" % (ob.NAMES[t], m, k, "  ".join(line)), flush=True)
        del ws
