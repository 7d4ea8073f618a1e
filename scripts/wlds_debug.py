"""debug: where does gemm_wlds differ from gemm_mfma?  prints a map of 32-token x 32-row blocks (x = differs)"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from __graft_entry__ import _load_package
from common import activations, random_block_bytes
from oracle import bindings as ob
be = _load_package().Cdna4Backend(0)
for t, m, k, n in [(ob.Q4_K, 256, 256, 256), (ob.Q4_K, 256, 1024, 256), (ob.Q4_K, 512, 1024, 512), (ob.IQ4_NL, 256, 1024, 256), (ob.IQ2_S, 256, 1024, 256)]:
    w = torch.from_numpy(random_block_bytes(t, m, k, 1)).cuda(); x = torch.from_numpy(activations(n, k, 2)).cuda()
    be.set_gemm_form(0); ref = be.mul_mat(t, w, x); i0 = be.last_launch_info()
    be.set_gemm_form(2); got = be.mul_mat(t, w, x); i2 = be.last_launch_info()
    got2 = be.mul_mat(t, w, x)
    d = (ref - got).abs() > 1e-3 * ref.abs().max()
    print(ob.NAMES[t], m, k, n, i0["kernel"], i0.get("ksplit"), i2["kernel"], "mismatch %.4f" % d.float().mean().item(), "repeatable", torch.equal(got, got2))
    blk = d.reshape(n // 32, 32, m // 32, 32).any(dim=3).any(dim=1).cpu().numpy()
    for tb in range(n // 32):
        print("  tok %3d: %s" % (32 * tb, "".join("x" if v else "." for v in blk[tb])))
    # inside the first bad block: which tokens / rows
    bad = np.argwhere(blk)
    if len(bad):
        tb, rb = bad[0]; sub = d[32 * tb:32 * tb + 32, 32 * rb:32 * rb + 32].cpu().numpy()
        print("  first bad block (tok %d, row %d): tokens %s rows %s" % (32 * tb, 32 * rb, "".join("x" if v else "." for v in sub.any(axis=1)), "".join("x" if v else "." for v in sub.any(axis=0))))
