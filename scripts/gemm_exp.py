#!/usr/bin/env python3
"""What bounds the prefill GEMM?  Builds variants of the library with one pipeline stage knocked out
(-DGEMM_EXP_*; see gemm_mfma.cuh) and times the same Q4_K mat-mul with each.  Results of the variants are wrong by
construction; only the time matters.   build (no GPU):  python scripts/gemm_exp.py build
                                        run (GPU):       python scripts/gemm_exp.py run"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXP = os.path.join(ROOT, "ik_llama.cpp_amd", "exp")
VARIANTS = {"base": [], "no_dequant": ["-DGEMM_EXP_NO_DEQUANT"], "no_aread": ["-DGEMM_EXP_NO_AREAD"], "no_xstore": ["-DGEMM_EXP_NO_XSTORE"],
            "no_dequant_no_aread": ["-DGEMM_EXP_NO_DEQUANT", "-DGEMM_EXP_NO_AREAD"],
            "same_rows": ["-DGEMM_EXP_SAME_ROWS"], "no_wload": ["-DGEMM_EXP_NO_WLOAD"],
            "mfma_only": ["-DGEMM_EXP_NO_DEQUANT", "-DGEMM_EXP_NO_AREAD", "-DGEMM_EXP_NO_XSTORE"]}

if sys.argv[1] == "build":
    sys.path.insert(0, ROOT)
    from __graft_entry__ import _load_package
    _load_package(); import ik_llama_cpp_amd.build as b
    os.makedirs(EXP, exist_ok=True)
    only = sys.argv[2:]
    for name, fl in VARIANTS.items():
        if only and name not in only: continue
        print(name, b.build_library(extra_flags=fl, out=os.path.join(EXP, "lib_%s.so" % name), tag="gemm_exp_" + name))
elif sys.argv[1] == "run":
    for name in VARIANTS:
        if not os.path.exists(os.path.join(EXP, "lib_%s.so" % name)): continue
        env = dict(os.environ, CDNA4_LIB=os.path.join(EXP, "lib_%s.so" % name))
        out = subprocess.run([sys.executable, __file__, "one"], env=env, capture_output=True, text=True)
        print("%-22s %s" % (name, out.stdout.strip().replace("\n", " | ")), flush=True)
        if out.returncode: print(out.stderr[-2000:])
else:
    import torch
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "scripts"))
    from __graft_entry__ import _load_package
    from oracle import bindings as ob
    from microbench import rot_weights
    be = _load_package().Cdna4Backend(0)
    m, k = 14336, 4096
    ws = rot_weights(ob.Q4_K, m, k, 128 << 20)
    for n in (512, 4096):
        x = torch.randn(n, k, device="cuda"); out = torch.empty(n, m, device="cuda")
        ms = be.time_mul_mat(ob.Q4_K, ws, x, out, warmup=5, iters=30)
        print("N=%d %.1f us %.0f TF" % (n, ms * 1e3, 2.0 * m * k * n / ms / 1e9))
