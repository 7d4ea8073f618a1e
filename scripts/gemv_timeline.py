#!/usr/bin/env python3
"""Where does a decode GEMV launch spend its time?  An experiment build (-DGEMV_EXP_TIMELINE) stamps the 100 MHz wall clock in
every workgroup at: start, first weight loads issued, prologue (activation quantization) done, done.
   build (no GPU): python scripts/gemv_timeline.py build        run (GPU): python scripts/gemv_timeline.py run"""
import ctypes as C, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "ik_llama.cpp_amd", "exp", "lib_timeline.so")

if sys.argv[1] == "build":
    sys.path.insert(0, ROOT)
    from __graft_entry__ import _load_package
    _load_package(); import ik_llama_cpp_amd.build as b
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    print(b.build_library(extra_flags=["-DGEMV_EXP_TIMELINE"], out=LIB, tag="timeline"))
else:
    os.environ["CDNA4_LIB"] = LIB
    import numpy as np, torch
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "scripts"))
    from __graft_entry__ import _load_package
    from oracle import bindings as ob
    from microbench import rot_weights
    be = _load_package().Cdna4Backend(0)
    lib = be.lib
    lib.cdna4_exp_set_timeline.argtypes = [C.c_void_p]; lib.cdna4_exp_timeline_wgs.restype = C.c_int
    tl = torch.zeros(4096 * 8, dtype=torch.int64, device="cuda")
    only = os.environ.get("TL_ONLY")
    cases = [("q4_K 4096x4096", ob.Q4_K, 4096, 4096, False), ("q4_K 6144x4096", ob.Q4_K, 6144, 4096, False), ("q4_K 4096x14336 (down)", ob.Q4_K, 4096, 14336, False),
             ("q6_K 4096x14336 (down)", ob.Q6_K, 4096, 14336, False), ("q4_K up*gate 14336x4096", ob.Q4_K, 14336, 4096, True), ("q6_K 128256x4096", ob.Q6_K, 128256, 4096, False),
             ("iq2_s 14336x4096", ob.IQ2_S, 14336, 4096, False), ("iq3_s 14336x4096", ob.IQ3_S, 14336, 4096, False), ("iq4_nl 14336x4096", ob.IQ4_NL, 14336, 4096, False),
             ("q4_K 14336x4096", ob.Q4_K, 14336, 4096, False), ("q4_K up*gate 14336x4096 + norm in the prologue", ob.Q4_K, 14336, 4096, "norm")]
    nw = torch.rand(4096, device="cuda") + 0.5
    for name, t, m, k, fused in cases:
        if only and not any(o in name for o in only.split(",")):
            continue
        ws = rot_weights(t, m, k, 512 << 20); x = torch.randn(1, k, device="cuda"); out = torch.empty(1, m, device="cuda")
        prod = torch.randn(1, k, device="cuda")
        rows = []
        for it in range(12):
            w = ws[it % len(ws)]; w2 = ws[(it + 1) % len(ws)]
            x.copy_(prod * 1.0001)                                   # activations freshly written by another kernel (cold for this one)
            lib.cdna4_exp_set_timeline(tl.data_ptr())
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            if fused == "norm": be.fused_up_gate_norm(t, w, w2, x, nw, 1e-5, out=out)
            elif fused: be.fused_up_gate(t, w, w2, x, out=out)
            else: be.mul_mat(t, w, x, out=out)
            e1.record(); torch.cuda.synchronize()
            lib.cdna4_exp_set_timeline(None)
            n = lib.cdna4_exp_timeline_wgs()
            a = tl[:8 * n].cpu().numpy().reshape(n, 8).astype(np.float64) * 0.01          # us
            if it == 11:
                t0 = a[:, 0].min(); dn = a[:, 3] - t0; st = a[:, 0] - t0
                print("    done percentiles p10/50/90/99/max: %s | start p50/90/max: %s | done by XCD (median): %s" % (
                    " ".join("%.2f" % np.percentile(dn, q) for q in (10, 50, 90, 99, 100)), " ".join("%.2f" % np.percentile(st, q) for q in (50, 90, 100)),
                    " ".join("%.2f" % np.median(dn[x::8]) for x in range(8))))
                print("    by XCD (median us): start %s | loads issued %s | prologue done %s" % (
                    " ".join("%.2f" % np.median(st[x::8]) for x in range(8)), " ".join("%.2f" % np.median((a[:, 1] - t0)[x::8]) for x in range(8)),
                    " ".join("%.2f" % np.median((a[:, 2] - t0)[x::8]) for x in range(8))))
                h = n // 2
                print("    first-half WGs (blockIdx < %d): start %.2f done %.2f | second half: start %.2f done %.2f | corr(start, done) %.2f | main-loop p10/50/90/max %s"
                      % (h, np.median(st[:h]), np.median(dn[:h]), np.median(st[h:]), np.median(dn[h:]), np.corrcoef(st, dn)[0, 1],
                         " ".join("%.2f" % np.percentile(a[:, 3] - a[:, 2], q) for q in (10, 50, 90, 100))))
            if it >= 4:
                t0 = a[:, 0].min()
                rows.append([a[:, 0].max() - t0, np.median(a[:, 1] - a[:, 0]), np.median(a[:, 2] - a[:, 0]), np.median(a[:, 3] - a[:, 2]),
                             np.median(a[:, 3] - t0), a[:, 3].max() - t0, e0.elapsed_time(e1) * 1e3, np.median(a[:, 4] - a[:, 0]), np.median(a[:, 5] - a[:, 0])])
        if os.environ.get("TL_GRAPH"):
            # the same launch as the LAST node of a HIP graph of 6 back-to-back launches (what the decode pass looks like): is the
            # per-XCD start skew the same when the previous kernel has just drained?
            lib.cdna4_exp_set_timeline(tl.data_ptr())
            def chain():
                for j in range(6):
                    if fused: be.fused_up_gate(t, ws[j % len(ws)], ws[(j + 1) % len(ws)], x, out=out)
                    else: be.mul_mat(t, ws[j % len(ws)], x, out=out)
            chain(); torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                chain()
            for _ in range(3):
                g.replay(); torch.cuda.synchronize()
            lib.cdna4_exp_set_timeline(None)
            n = lib.cdna4_exp_timeline_wgs()
            a = tl[:8 * n].cpu().numpy().reshape(n, 8).astype(np.float64) * 0.01
            t0 = a[:, 0].min()
            print("    IN GRAPH (6th of 6 back-to-back launches) by XCD: start %s | done %s | last done +%.2f" % (
                " ".join("%.2f" % np.median((a[:, 0] - t0)[x::8]) for x in range(8)), " ".join("%.2f" % np.median((a[:, 3] - t0)[x::8]) for x in range(8)), a[:, 3].max() - t0))
        r = np.median(np.array(rows), axis=0)
        print("%-28s wgs=%4d | last WG starts +%.2f | work decomposition done +%.2f | activation loads issued +%.2f | weight loads issued +%.2f | prologue done +%.2f | main loop %.2f | median WG done +%.2f | last WG done +%.2f us | event-to-event %.1f us"
              % (name, n, r[0], r[7], r[8], r[1], r[2], r[3], r[4], r[5], r[6]), flush=True)
