#!/usr/bin/env python3
"""Variant builds of the workgroup-shared weight tile prompt GEMM (csrc/gemm_wlds.cuh): only the Q4_K / Q6_K / IQ4_NL GEMM translation units are recompiled with the variant's
-D flags, everything else links the base objects.   python scripts/wlds_exp.py build <name>=<flags,comma separated> ...      then on the GPU:
python scripts/nt_bench.py --lib ik_llama.cpp_amd/libggml-hip-cdna4.so --lib ik_llama.cpp_amd/exp/lib_wlds_<name>.so --op upgate:12:14336:4096:4096 ..."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import _load_package
_load_package(); import ik_llama_cpp_amd.build as b
os.makedirs(os.path.join(ROOT, "ik_llama.cpp_amd", "exp"), exist_ok=True)
for spec in sys.argv[2:]:
    name, flags = spec.split("=", 1)
    out = os.path.join(ROOT, "ik_llama.cpp_amd", "exp", "lib_wlds_%s.so" % name)
    print(b.build_library(extra_flags=[f for f in flags.split(",") if f], out=out, tag="wlds_" + name, only=["gemm_12", "gemm_14", "gemm_20"]))
