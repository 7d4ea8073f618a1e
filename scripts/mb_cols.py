#!/usr/bin/env python3
"""multi-column decode microbench: 14336x4096 and 4096x14336, N = 1, 2, 4, 8 for the K-quants (plain) and the fused up*gate launch.
CDNA4_GEMV_MFMA=0 selects the v_dot4 kernels for 2..8 columns (A/B)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "scripts"))
from __graft_entry__ import _load_package
from oracle import bindings as ob
from microbench import rot_weights
be = _load_package().Cdna4Backend(0)
for t in (ob.Q4_K, ob.Q5_K, ob.Q6_K):
    for (m, k) in ((14336, 4096), (4096, 14336)):
        ws = rot_weights(t, m, k)
        line = []
        for n in (1, 2, 4, 8):
            x = torch.randn(n, k, device="cuda"); out = torch.empty(n, m, device="cuda")
            ms = be.time_mul_mat(t, ws, x, out, warmup=5, iters=50)
            line.append("N=%d %6.2f us" % (n, ms * 1e3))
        print("%-5s %5dx%-5d  %s" % (ob.NAMES[t], m, k, "  ".join(line)), flush=True)
        del ws
for t in (ob.Q4_K,):
    m, k = 14336, 4096
    ws = rot_weights(t, m, k, 768 << 20); n_pairs = len(ws) // 2
    line = []
    for n in (1, 2, 4, 8):
        x = torch.randn(n, k, device="cuda"); out = torch.empty(n, m, device="cuda")
        def sweep():
            for i in range(n_pairs):
                be.fused_up_gate(t, ws[2 * i], ws[2 * i + 1], x, out=out)
        sweep(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            sweep()
        e1.record(); torch.cuda.synchronize()
        line.append("N=%d %6.2f us" % (n, e0.elapsed_time(e1) / (10 * n_pairs) * 1e3))
    print("fused up*gate %-5s  %s" % (ob.NAMES[t], "  ".join(line)), flush=True)
