#!/usr/bin/env python3
"""What bounds the multi-column MFMA decode kernel?  Knock-out builds (-DGMF_EXP_*: results wrong, only time matters).
   build (no GPU): python scripts/gmf_exp.py build        run (GPU): python scripts/gmf_exp.py run"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXP = os.path.join(ROOT, "ik_llama.cpp_amd", "exp")
VARIANTS = {"base": [], "no_load": ["-DGMF_EXP_NO_LOAD"], "no_epi": ["-DGMF_EXP_NO_EPI"], "no_load_no_epi": ["-DGMF_EXP_NO_LOAD", "-DGMF_EXP_NO_EPI"]}
if sys.argv[1] == "build":
    sys.path.insert(0, ROOT)
    from __graft_entry__ import _load_package
    _load_package(); import ik_llama_cpp_amd.build as b
    os.makedirs(EXP, exist_ok=True)
    for name, fl in VARIANTS.items():
        print(name, b.build_library(extra_flags=fl, out=os.path.join(EXP, "libgmf_%s.so" % name), tag="gmf_exp_" + name))
else:
    for name in VARIANTS:
        env = dict(os.environ, CDNA4_LIB=os.path.join(EXP, "libgmf_%s.so" % name))
        out = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "mb_cols2.py")], env=env, capture_output=True, text=True)
        print("%-16s %s" % (name, out.stdout.strip().split("\n")[-1]), flush=True)
