#!/usr/bin/env python3
"""Variant builds of the ping-pong prompt GEMM (csrc/gemm_pp.cuh): only the Q4_K GEMM translation unit (or --tus a,b) is recompiled with the variant's -D flags, everything else
links the base objects.   python scripts/pp_exp.py <name>=<flags,comma separated> ...     -> ik_llama.cpp_amd/exp/lib_<name>.so (select with CDNA4_LIB=...)"""
import os, sys
from concurrent.futures import ThreadPoolExecutor
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import _load_package
_load_package(); import ik_llama_cpp_amd.build as b
os.makedirs(os.path.join(ROOT, "ik_llama.cpp_amd", "exp"), exist_ok=True)
tus = ["gemm_12"]
specs = [a for a in sys.argv[1:] if "=" in a and not a.startswith("--")]
for a in sys.argv[1:]:
    if a.startswith("--tus="):
        tus = a[6:].split(",")
def one(spec):
    name, flags = spec.split("=", 1)
    out = os.path.join(ROOT, "ik_llama.cpp_amd", "exp", "lib_%s.so" % name)
    return b.build_library(extra_flags=[f for f in flags.split(",") if f], out=out, tag="pp_" + name, only=tus)
with ThreadPoolExecutor(len(specs) or 1) as ex:
    for r in ex.map(one, specs):
        print(r)
