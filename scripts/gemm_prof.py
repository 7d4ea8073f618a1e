import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from __graft_entry__ import _load_package
from oracle import bindings as ob
from microbench import rot_weights
pkg = _load_package(); be = pkg.Cdna4Backend(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
m, k = 14336, 4096
ws = rot_weights(ob.Q4_K, m, k, 128 << 20)
x = torch.randn(n, k, device="cuda"); out = torch.empty(n, m, device="cuda")
for i in range(6):
    be.mul_mat(ob.Q4_K, ws[i % len(ws)], x, out=out)
torch.cuda.synchronize()
