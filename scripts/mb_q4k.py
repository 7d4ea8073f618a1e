import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from __graft_entry__ import _load_package
from oracle import bindings as ob
from microbench import rot_weights
pkg = _load_package(); be = pkg.Cdna4Backend(0)
for (m, k) in [(4096, 4096), (14336, 4096), (4096, 14336), (128256, 4096)]:
    ws = rot_weights(ob.Q4_K, m, k)
    x = torch.randn(1, k, device="cuda"); out = torch.empty(1, m, device="cuda")
    ms = min(be.time_mul_mat(ob.Q4_K, ws, x, out, warmup=5, iters=50) for _ in range(3))
    by = m * ob.row_size(ob.Q4_K, k)
    print("M=%6d K=%5d %8.2f us %7.1f GB/s" % (m, k, ms * 1e3, by / ms / 1e6))
