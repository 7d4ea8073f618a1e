#!/usr/bin/env python3
"""quick A/B of the multi-column MFMA decode kernel: Q4_K / Q6_K 14336x4096 at N = 8 (env knobs are read once per process)"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "scripts"))
from __graft_entry__ import _load_package
from oracle import bindings as ob
from microbench import rot_weights
be = _load_package().Cdna4Backend(0)
out = []
for t in (ob.Q4_K, ob.Q6_K):
    for (m, k) in ((14336, 4096), (4096, 14336)):
        ws = rot_weights(t, m, k)
        for n in (2, 8):
            x = torch.randn(n, k, device="cuda"); o = torch.empty(n, m, device="cuda")
            ms = be.time_mul_mat(t, ws, x, o, warmup=5, iters=50)
            out.append("%s %dx%d N=%d %.2f" % (ob.NAMES[t], m, k, n, ms * 1e3))
        del ws
print(os.environ.get("CDNA4_GEMV_MFMA_KS", "auto"), " | ".join(out), flush=True)
