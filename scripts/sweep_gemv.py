"""developer sweep: decode GEMV launch geometry (workgroups per CU x waves per workgroup) on the model's decode shapes."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = r'''
import os, sys, torch
ROOT = %r
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "scripts"))
from __graft_entry__ import _load_package
from oracle import bindings as ob
from microbench import rot_weights
pkg = _load_package(); be = pkg.Cdna4Backend(0)
res = []
for (t, m, k, ug) in [(ob.Q4_K, 4096, 4096, 0), (ob.Q4_K, 14336, 4096, 1), (ob.Q4_K, 4096, 14336, 0), (ob.Q6_K, 4096, 14336, 0), (ob.Q6_K, 128256, 4096, 0)]:
    ws = rot_weights(t, m, k, 512 << 20)
    x = torch.randn(1, k, device="cuda"); out = torch.empty(1, m, device="cuda")
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    def run():
        for i in range(len(ws) - 1):
            if ug: be.fused_up_gate(t, ws[i], ws[i + 1], x, out=out)
            else: be.mul_mat(t, ws[i], x, out=out)
    run(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0.record(); run(); run(); e1.record(); torch.cuda.synchronize(); best = min(best, e0.elapsed_time(e1) / (2 * (len(ws) - 1)))
    res.append("%%.2f" %% (best * 1e3))
print(" ".join(res))
''' % ROOT
for waves in (4, 8):
    for per_cu in (1, 2, 3, 4):
        env = dict(os.environ, CDNA4_GEMV_WAVES=str(waves), CDNA4_GEMV_PER_CU=str(per_cu))
        out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
        print("waves=%d per_cu=%d  us: %s" % (waves, per_cu, out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr[-300:]), flush=True)
