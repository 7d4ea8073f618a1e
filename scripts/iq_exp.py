#!/usr/bin/env python3
"""What bounds the IQ2_S / IQ3_S decode GEMV (0.245 of the HBM roof on the fused 37.7 MB launch)?  Knock-out / geometry variants of the two types' TUs only
(the rest of the library links as the base objects), timed against the base library and against Q4_K in ONE run.
   build (no GPU):  python scripts/iq_exp.py build        run (GPU):  python scripts/iq_exp.py run
   nocompute  -DGEMV_EXP_NO_COMPUTE   the weight ring is consumed by a checksum: the load path alone (5 loads of 2 ... 8 bytes per lane per 64 weights)
   w12        -DGEMV_MAX_THREADS=768  + CDNA4_GEMV_WAVES=12 CDNA4_GEMV_PER_CU=1: three waves per SIMD in ONE workgroup per CU (one prologue)
Results of the nocompute build are wrong by construction; only the time matters."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXP = os.path.join(ROOT, "ik_llama.cpp_amd", "exp")
TUS = ["gemv_%d_%s" % (t, k) for t in (22, 21, 12) for k in ("plain", "upgate")]
VARIANTS = {"nocompute": (["-DGEMV_EXP_NO_COMPUTE"], {}), "w12": (["-DGEMV_MAX_THREADS=768"], {"CDNA4_GEMV_WAVES": "12", "CDNA4_GEMV_PER_CU": "1"}),
            "w12p2": (["-DGEMV_MAX_THREADS=768"], {"CDNA4_GEMV_WAVES": "12", "CDNA4_GEMV_PER_CU": "2"}),
            "w12x": (["-DGEMV_MAX_THREADS=768"], {"CDNA4_GEMV_WAVES": "12", "CDNA4_GEMV_PER_CU": "1", "CDNA4_GEMV_NR": "1"}),
            "base_w8": ([], {"CDNA4_GEMV_WAVES": "8", "CDNA4_GEMV_PER_CU": "1"}), "base_nr1": ([], {"CDNA4_GEMV_NR": "1"})}


def lib_of(v):
    v = {"w12x": "w12", "w12p2": "w12"}.get(v, v)
    return os.path.join(EXP, "lib_iq_%s.so" % v) if VARIANTS[v][0] else os.path.join(ROOT, "ik_llama.cpp_amd", "libggml-hip-cdna4.so")


if sys.argv[1] == "build":
    sys.path.insert(0, ROOT)
    from __graft_entry__ import _load_package
    _load_package(); import ik_llama_cpp_amd.build as b
    os.makedirs(EXP, exist_ok=True)
    done = {}
    for v, (flags, _) in VARIANTS.items():
        if flags and tuple(flags) not in done:
            print(v, b.build_library(extra_flags=flags, out=lib_of(v), tag="iq_" + v, only=TUS)); done[tuple(flags)] = lib_of(v)
elif sys.argv[1] == "run":
    for v in ["base"] + list(VARIANTS):
        env = dict(os.environ); env.update(VARIANTS[v][1] if v != "base" else {})
        if v != "base" and VARIANTS[v][0]: env["CDNA4_LIB"] = lib_of(v)
        r = subprocess.run([sys.executable, __file__, "one", v], env=env, capture_output=True, text=True, timeout=300)
        print(r.stdout.strip() or r.stderr[-600:], flush=True)
else:
    import torch
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "scripts"))
    from __graft_entry__ import _load_package
    from oracle import bindings as ob
    from microbench import rot_weights
    be = _load_package().Cdna4Backend(0)
    cells = []
    for name, t in (("iq2_s", ob.IQ2_S), ("iq3_s", ob.IQ3_S), ("q4_K", ob.Q4_K)):
        m, k = 14336, 4096
        ws = rot_weights(t, m, k, 512 << 20); x = torch.randn(1, k, device="cuda"); out = torch.empty(1, m, device="cuda")
        mb = m * ob.row_size(t, k) / 1e6
        for fused in (False, True):
            def call(i):
                if fused: be.fused_up_gate(t, ws[i % len(ws)], ws[(i + 1) % len(ws)], x, out=out)
                else: be.mul_mat(t, ws[i % len(ws)], x, out=out)
            g = torch.cuda.CUDAGraph(); st = torch.cuda.Stream()
            call(0); torch.cuda.synchronize()
            with torch.cuda.graph(g, stream=st):
                for i in range(len(ws)): call(i)
            g.replay(); torch.cuda.synchronize()
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10): g.replay()
            e1.record(); torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / (10 * len(ws))
            cells.append("%s%s %.2f us (%.2f TB/s)" % (name, "*2" if fused else "", us, mb * (2 if fused else 1) / us))
    print("%-10s %s" % (sys.argv[2], " | ".join(cells)))
