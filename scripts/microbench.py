#!/usr/bin/env python3
"""Per-kernel micro-benchmark on cuda:0: decode GEMV GB/s (algorithmic bytes, cold cache via rotating weight buffers)
and prefill GEMM TFLOP/s for model-shaped weights.  Usage: python scripts/microbench.py [gemv|fused|gemm|moe|all]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from __graft_entry__ import _load_package          # noqa: E402
from common import random_block_bytes               # noqa: E402
from oracle import bindings as ob                   # noqa: E402

HBM_PEAK = 8.0e12


def rot_weights(t, m, k, total_bytes=768 << 20):
    rs = ob.row_size(t, k); one = m * rs
    n = max(2, min(24, total_bytes // one))
    base = torch.from_numpy(random_block_bytes(t, min(m, 512), k, 5)).cuda()
    reps = (m + base.shape[0] - 1) // base.shape[0]
    w0 = base.repeat(reps, 1)[:m].contiguous()
    return [w0.clone() for _ in range(n)]


def main():
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    pkg = _load_package(); be = pkg.Cdna4Backend(0)
    print(be.description())
    shapes = [(4096, 4096), (14336, 4096), (4096, 14336), (128256, 4096)]
    if what in ("gemv", "all"):
        for t in ([ob.IQ2_S, ob.IQ3_S] if os.environ.get('MB_ONLY_IQ') else ob.BASE_TYPES):
            for (m, k) in shapes:
                if t != ob.Q4_K and (m, k) not in ((14336, 4096), (4096, 14336)):
                    continue
                ws = rot_weights(t, m, k)
                for n in ((1, 2, 4, 8) if (t == ob.Q4_K and m == 14336) else (1, 4) if (m == 14336 and os.environ.get("MB_GEMV_N4")) else (1,)):
                    x = torch.randn(n, k, device="cuda"); out = torch.empty(n, m, device="cuda")
                    ms = be.time_mul_mat(t, ws, x, out, warmup=5, iters=50)
                    by = m * ob.row_size(t, k) + 4 * k * n + 4 * m * n
                    print("gemv %-7s M=%6d K=%5d N=%d  %8.2f us  %7.1f GB/s  %5.1f%% of 8 TB/s" % (ob.NAMES[t], m, k, n, ms * 1e3, by / ms / 1e6, 100 * by / (ms * 1e-3) / HBM_PEAK))
                del ws
    if what in ("fused", "gemv", "all") and not os.environ.get("MB_ONLY_IQ"):          # fused up*gate decode GEMV, one HIP graph over rotating (cold) weight pairs
        for t in (ob.Q4_K, ob.Q6_K):
            m, k = 14336, 4096
            ws = rot_weights(t, m, k, 768 << 20); n_pairs = len(ws) // 2
            for n in (1, 2, 4):
                x = torch.randn(n, k, device="cuda"); out = torch.empty(n, m, device="cuda")
                def sweep():
                    for i in range(n_pairs):
                        be.fused_up_gate(t, ws[2 * i], ws[2 * i + 1], x, out=out)
                sweep(); torch.cuda.synchronize()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    sweep()
                g.replay(); torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10):
                    g.replay()
                e1.record(); torch.cuda.synchronize()
                ms = e0.elapsed_time(e1) / (10 * n_pairs)
                by = 2 * m * ob.row_size(t, k) + 4 * k * n + 4 * m * n
                print("fused up*gate %-5s M=%6d K=%5d N=%d  %8.2f us  %7.1f GB/s  %5.1f%% of 8 TB/s" % (ob.NAMES[t], m, k, n, ms * 1e3, by / ms / 1e6, 100 * by / (ms * 1e-3) / HBM_PEAK))
                del g
            del ws
    if what in ("gemm", "all"):
        for t in ([getattr(ob, v) for v in os.environ["MB_GEMM_TYPES"].split(",")] if os.environ.get("MB_GEMM_TYPES") else (ob.Q4_K, ob.Q6_K)):
            for (m, k) in shapes[:3]:
                ws = rot_weights(t, m, k, 256 << 20)
                for n in (32, 128, 512, 4096):
                    x = torch.randn(n, k, device="cuda"); out = torch.empty(n, m, device="cuda")
                    try:
                        ms = be.time_mul_mat(t, ws, x, out, warmup=2, iters=10)
                    except Exception as e:
                        print("gemm", ob.NAMES[t], m, k, n, "FAILED", e); continue
                    fl = 2.0 * m * k * n
                    print("gemm %-7s M=%6d K=%5d N=%4d  %9.2f us  %8.1f TFLOP/s  %5.1f%% of 2.5 PF" % (ob.NAMES[t], m, k, n, ms * 1e3, fl / ms / 1e9, 100 * fl / (ms * 1e-3) / 2.5e15))
    if what in ("moe", "all"):          # MUL_MAT_ID / MOE_FUSED_UP_GATE at Qwen3-30B-A3B expert shapes (128 experts, 8 used, 2048 -> 768)
        # second argument "mixtral": Mixtral-8x7B expert shapes (8 experts, 2 used, 4096 -> 14336)
        E, NU, K, FF = (8, 2, 4096, 14336) if (len(sys.argv) > 2 and sys.argv[2] == "mixtral") else (128, 8, 2048, 768)
        t = ob.Q4_K; torch.manual_seed(0)        # (the same expert choices in every run: tile counts, and with them the times, move by ~10 % between draws)
        def experts(m, k, seed):
            base = torch.from_numpy(random_block_bytes(t, m, k, seed)).cuda()
            return base.unsqueeze(0).repeat(E, 1, 1).contiguous()
        up, gate, down = experts(FF, K, 11), experts(FF, K, 12), experts(K, FF, 13)
        def graph_time(fn, reps=10):
            fn(); torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                for _ in range(reps):
                    fn()
            g.replay(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                g.replay()
            e1.record(); torch.cuda.synchronize()
            return e0.elapsed_time(e1) / (5 * reps)
        for T in ([int(v) for v in os.environ["MB_MOE_T"].split(",")] if os.environ.get("MB_MOE_T") else (1, 2, 4, 8, 16, 32, 64, 128, 512)):
            x = torch.randn(T, 1, K, device="cuda"); h = torch.randn(T, NU, FF, device="cuda")
            ids = torch.stack([torch.randperm(E, device="cuda")[:NU] for _ in range(T)]).to(torch.int32).contiguous()
            o1 = torch.empty(T, NU, FF, device="cuda"); o2 = torch.empty(T, NU, K, device="cuda")
            tu = graph_time(lambda: be.moe_fused_up_gate(t, up, gate, x, ids, out=o1))
            td = graph_time(lambda: be.mul_mat_id(t, down, h, ids, out=o2))
            nexp = len(torch.unique(ids))            # distinct experts touched
            by_u = nexp * 2 * FF * ob.row_size(t, K); by_d = nexp * K * ob.row_size(t, FF)
            fl_u = 2.0 * 2 * FF * K * T * NU; fl_d = 2.0 * K * FF * T * NU
            print("moe q4_K E=%d K=%d FF=%d T=%3d (%3d experts)  fused up*gate %8.2f us (%6.1f GB/s distinct weights, %6.1f TFLOP/s)   down %8.2f us (%6.1f GB/s, %6.1f TFLOP/s)" %
                  (E, K, FF, T, nexp, tu * 1e3, by_u / tu / 1e6, fl_u / tu / 1e9, td * 1e3, by_d / td / 1e6, fl_d / td / 1e9))
    be.close()


if __name__ == "__main__":
    main()
