"""Torch-free micro-benchmark / parity probe of the C ABI: ctypes + libamdhip64 only, so a `gpurun` call that runs it is charged ~20 s (box + push + a few
seconds of run) instead of the 3-5 minutes a call that imports torch costs (the first `import torch` on a fresh box pages in for 1-2 minutes).

    python scripts/nt_bench.py [--lib PATH ...] [--case TYPE:M:K:N ...] [--op upgate:TYPE:M:K | fa:N_HEAD:N_HEAD_KV:N_KV[:N_TOK] ...] [--iters 200] [--check]

* --lib: one or more builds of libggml-hip-cdna4.so (default: the in-tree one); with several, every case is timed on each in the SAME process, interleaved
  (A, B, A, B: the same-process A/B of bench.py --ab-lib without torch).
* --case: ggml type id : weight rows : row length : activation columns (default: the Llama-3-8B Q4_K shapes of the bench line).
* timing: cdna4_time_mul_mat (HIP events on the launch stream) over weight copies rotated through more than the 256 MiB of infinity cache (cold weights, as in a
  real token), reported as us per launch and as a fraction of the bound (8 TB/s for N <= 8, 2.5 PFLOP/s dense f16 above).
* --check: the result of the first copy against the CPU oracle (only sensible for small shapes: the oracle is a scalar restatement).
* --op: other entry points timed with HIP events on the null stream (back-to-back eager launches):
    upgate:TYPE:M:K            cdna4_fused_up_gate, one activation row (the dominant launch of a decode token: two M x K matrices), cold rotating weight pairs
    fa:NH:NHKV:NKV[:NTOK[:NVIS]]  (NVIS 0 = causal prompt mask)   cdna4_op_flash_attn, head size 128, f16 K / V of NKV keys, f32 q of NTOK tokens (default 1: the decode kernel), all-zero f16 mask
One JSON line per (case, lib)."""
import argparse
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from common import activations, random_block_bytes  # noqa: E402
from oracle import bindings as ob  # noqa: E402

P, I, L64 = C.c_void_p, C.c_int, C.c_long
HBM_PEAK_GBS, MFMA_F16_PEAK_TFLOPS = 8000.0, 2500.0


class Hip:
    def __init__(self):
        for name in ("libamdhip64.so", "/opt/rocm/lib/libamdhip64.so"):
            try:
                self.h = C.CDLL(name, mode=C.RTLD_GLOBAL); break
            except OSError:
                continue
        self.h.hipMalloc.argtypes = [C.POINTER(P), C.c_size_t]; self.h.hipMemcpy.argtypes = [P, P, C.c_size_t, I]; self.h.hipFree.argtypes = [P]; self.h.hipMemset.argtypes = [P, I, C.c_size_t]
        self.h.hipEventCreate.argtypes = [C.POINTER(P)]; self.h.hipEventRecord.argtypes = [P, P]; self.h.hipEventSynchronize.argtypes = [P]; self.h.hipEventElapsedTime.argtypes = [C.POINTER(C.c_float), P, P]
        self.ev = None

    def time_us(self, launch, iters, warmup):
        """average microseconds of `launch(i)` over `iters` back-to-back calls on the null stream"""
        if self.ev is None:
            self.ev = [P(), P()]
            for e in self.ev:
                self.check(self.h.hipEventCreate(C.byref(e)), "hipEventCreate")
        for i in range(warmup):
            launch(i)
        self.check(self.h.hipEventRecord(self.ev[0], None), "hipEventRecord")
        for i in range(iters):
            launch(i)
        self.check(self.h.hipEventRecord(self.ev[1], None), "hipEventRecord"); self.check(self.h.hipEventSynchronize(self.ev[1]), "hipEventSynchronize")
        ms = C.c_float(0); self.check(self.h.hipEventElapsedTime(C.byref(ms), self.ev[0], self.ev[1]), "hipEventElapsedTime")
        return ms.value * 1e3 / iters

    def check(self, rc, what):
        if rc != 0:
            raise RuntimeError("%s failed: hipError %d" % (what, rc))

    def malloc(self, n):
        p = P(); self.check(self.h.hipMalloc(C.byref(p), n), "hipMalloc"); return p

    def upload(self, arr):
        arr = np.ascontiguousarray(arr); p = self.malloc(arr.nbytes); self.check(self.h.hipMemcpy(p, arr.ctypes.data_as(P), arr.nbytes, 1), "hipMemcpy H2D"); return p

    def download(self, p, shape, dtype):
        out = np.empty(shape, dtype); self.check(self.h.hipMemcpy(out.ctypes.data_as(P), p, out.nbytes, 2), "hipMemcpy D2H"); return out


def load_lib(path):
    lib = C.CDLL(path)
    lib.cdna4_init.restype = P; lib.cdna4_init.argtypes = [I]; lib.cdna4_free.argtypes = [P]; lib.cdna4_last_error.restype = C.c_char_p
    lib.cdna4_mul_mat.argtypes = [P, L64, L64, L64, I, P, L64, I, P, L64, P, L64, P]
    lib.cdna4_time_mul_mat.argtypes = [P, L64, L64, L64, I, P, I, L64, P, L64, P, L64, I, I, P, C.POINTER(C.c_float)]
    lib.cdna4_reserve_workspace.argtypes = [P, C.c_size_t]
    lib.cdna4_fused_up_gate.argtypes = [P, L64, L64, L64, I, I, P, P, L64, I, P, L64, P, L64, P]
    lib.cdna4_op_flash_attn.argtypes = [P, P, P, P, P, P, C.c_float, C.c_float, C.c_float, P]
    return lib


class Tensor(C.Structure):          # cdna4_tensor {data, type, ne[4], nb[4]} (include/ggml_hip_cdna4.h)
    _fields_ = [("data", P), ("type", I), ("ne", C.c_int64 * 4), ("nb", C.c_int64 * 4)]


def tensor(data, t, ne, elem):
    nb = [elem, elem * ne[0], elem * ne[0] * ne[1], elem * ne[0] * ne[1] * ne[2]]
    return Tensor(data, t, (C.c_int64 * 4)(*ne), (C.c_int64 * 4)(*nb))


def run_op(spec, hip, built, ctxs, a):
    kind, *v = spec.split(":"); v = [int(x) for x in v]
    if kind == "upgate":
        t, m, k = v[:3]; n = v[3] if len(v) > 3 else 1           # upgate:TYPE:M:K[:N] -- N activation rows (N > 8: the prompt GEMM; reported against the MFMA roof)
        wu = random_block_bytes(t, m, k, 3); wg = random_block_bytes(t, m, k, 4); x = activations(n, k, 5)
        n_rot = max(2, min(32, (320 << 20) // (2 * wu.nbytes) + 1))
        du = [hip.upload(wu) for _ in range(n_rot)]; dg = [hip.upload(wg) for _ in range(n_rot)]; xd = hip.upload(x); cd = hip.malloc(4 * m * n)
        for (p, lib), ctx in zip(built, ctxs):
            def launch(i, lib=lib, ctx=ctx):
                rc = lib.cdna4_fused_up_gate(ctx, m, n, k, 10, t, du[i % n_rot], dg[i % n_rot], wu.shape[1], 0, xd, 4 * k, cd, m, None)      # 10 = GGML_UNARY_OP_SILU
                if rc != 0:
                    raise RuntimeError("cdna4_fused_up_gate rc %d: %s" % (rc, lib.cdna4_last_error()))
            us = min(hip.time_us(launch, a.iters, a.warmup) for _ in range(a.rounds))
            nbytes = 2 * wu.nbytes; rec = {"op": spec, "type": ob.NAMES.get(t, str(t)), "lib": os.path.relpath(p, ROOT), "us": round(us, 3), "rotating_pairs": n_rot}
            if n <= 8:
                rec.update({"gbs": round(nbytes / us / 1e3, 1), "frac_hbm": round(nbytes / us / 1e3 / HBM_PEAK_GBS, 4), "note": "eager back-to-back launches (a decode token replays them from a HIP graph)"})
            else:
                tf = 2.0 * 2 * m * k * n / us / 1e6; rec.update({"tflops": round(tf, 1), "frac_mfma": round(tf / MFMA_F16_PEAK_TFLOPS, 4), "note": "op = f16 activation image + GEMM, eager"})
            print(json.dumps(rec), flush=True)
        for d in du + dg + [xd, cd]:
            hip.h.hipFree(d)
    elif kind == "fa":
        nh, nhkv, nkv = v[:3]; ntok = v[3] if len(v) > 3 else 1; nvis0 = v[4] if len(v) > 4 else nkv; D = 128          # fa:NH:NHKV:NKV[:NTOK[:NVIS]] -- NVIS visible keys (mask -inf behind)
        rng = np.random.default_rng(6)
        q = rng.standard_normal((nh, ntok, D)).astype(np.float32); kk = rng.standard_normal((nhkv, nkv, D)).astype(np.float16); vv = rng.standard_normal((nhkv, nkv, D)).astype(np.float16)
        npad = (ntok + 31) // 32 * 32
        mask = np.zeros((npad, nkv), np.float16); mask[:, nvis0:] = -np.inf
        causal = nvis0 == 0                                                              # NVIS = 0: the causal mask of a prompt batch at the end of the window
        if causal:
            mask[:] = 0
            for t in range(ntok):
                mask[t, nkv - ntok + t + 1:] = -np.inf
            mask[ntok:] = -np.inf
        qd, kd, vd, md, od = hip.upload(q), hip.upload(kk), hip.upload(vv), hip.upload(mask), hip.malloc(4 * D * nh * ntok)
        tq = tensor(qd, 0, [D, ntok, nh, 1], 4); tk = tensor(kd, 1, [D, nkv, nhkv, 1], 2); tv = tensor(vd, 1, [D, nkv, nhkv, 1], 2)
        tm = tensor(md, 1, [nkv, npad, 1, 1], 2); to = tensor(od, 0, [D, nh, ntok, 1], 4)
        for (p, lib), ctx in zip(built, ctxs):
            def launch(i, lib=lib, ctx=ctx):
                rc = lib.cdna4_op_flash_attn(ctx, C.byref(tq), C.byref(tk), C.byref(tv), C.byref(tm), C.byref(to), 1.0 / np.sqrt(D), 0.0, 0.0, None)
                if rc != 0:
                    raise RuntimeError("cdna4_op_flash_attn rc %d: %s" % (rc, lib.cdna4_last_error()))
            us = min(hip.time_us(launch, a.iters, a.warmup) for _ in range(a.rounds))
            rec = {"op": spec, "lib": os.path.relpath(p, ROOT), "us": round(us, 3), "kv_bytes": int(kk.nbytes + vv.nbytes)}
            if a.check:       # float64 soft-max attention of the same inputs
                hip.check(hip.h.hipDeviceSynchronize(), "sync"); got = hip.download(od, (ntok, nh, D), np.float32)
                g = nh // nhkv; want = np.empty((ntok, nh, D))
                for h in range(nh):
                    s_ = q[h].astype(np.float64) @ kk[h // g, :nvis0 or nkv].astype(np.float64).T / np.sqrt(D)
                    if causal:
                        s_ = s_ + mask[:ntok].astype(np.float64)
                    s_ -= s_.max(axis=1, keepdims=True); pr = np.exp(s_); pr /= pr.sum(axis=1, keepdims=True)
                    want[:, h] = pr @ vv[h // g, :nvis0 or nkv].astype(np.float64)
                rec["nmse_vs_f64"] = float(np.sum((got - want) ** 2) / np.sum(want ** 2))
            if a.stress:      # every launch with a fresh q AND a fresh number of visible keys (mask -inf beyond), checked against float64: a partial of an EARLIER launch picked up by
                              # the combining workgroup (stale L2 line, lost write-through) shows as a wrong row; streaming load beside it (a copy kernel on a second stream) optional
                g = nh // nhkv; worst = 0.0; bad = 0; r2 = np.random.default_rng(11)
                for it in range(a.stress):
                    q2 = r2.standard_normal((nh, ntok, D)).astype(np.float32); nvis = int(r2.integers(1, nkv + 1))
                    m2 = np.zeros((npad, nkv), np.float16); m2[:, nvis:] = -np.inf
                    hip.check(hip.h.hipMemcpy(qd, q2.ctypes.data_as(P), q2.nbytes, 1), "H2D q"); hip.check(hip.h.hipMemcpy(md, m2.ctypes.data_as(P), m2.nbytes, 1), "H2D mask")
                    for rep in range(3):          # (back to back: the third launch's combine runs while nothing else changes)
                        launch(it)
                    hip.check(hip.h.hipDeviceSynchronize(), "sync"); got = hip.download(od, (ntok, nh, D), np.float32)
                    want = np.empty((ntok, nh, D))
                    for h in range(nh):
                        s_ = q2[h].astype(np.float64) @ kk[h // g, :nvis].astype(np.float64).T / np.sqrt(D); s_ -= s_.max(axis=1, keepdims=True); pr = np.exp(s_); pr /= pr.sum(axis=1, keepdims=True)
                        want[:, h] = pr @ vv[h // g, :nvis].astype(np.float64)
                    e = float(np.sum((got - want) ** 2) / np.sum(want ** 2)); worst = max(worst, e); bad += e > 1e-9 or not np.all(np.isfinite(got))
                rec["stress_launches"] = 3 * a.stress; rec["stress_bad"] = int(bad); rec["stress_worst_nmse"] = worst
                hip.check(hip.h.hipMemcpy(qd, q.ctypes.data_as(P), q.nbytes, 1), "H2D q"); hip.check(hip.h.hipMemcpy(md, mask.ctypes.data_as(P), mask.nbytes, 1), "H2D mask")
            print(json.dumps(rec), flush=True)
        for d in (qd, kd, vd, md, od):
            hip.h.hipFree(d)
    else:
        raise SystemExit("unknown --op %r" % spec)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lib", action="append"); ap.add_argument("--case", action="append"); ap.add_argument("--iters", type=int, default=200); ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--check", action="store_true"); ap.add_argument("--rounds", type=int, default=2); ap.add_argument("--op", action="append")
    ap.add_argument("--stress", type=int, default=0, help="fa ops: this many extra launches, each with fresh q / visible keys, each checked against float64")
    a = ap.parse_args()
    libs = a.lib or [os.environ.get("CDNA4_LIB", os.path.join(ROOT, "ik_llama.cpp_amd", "libggml-hip-cdna4.so"))]
    cases = a.case or ([] if a.op else ["12:14336:4096:1", "12:4096:4096:1", "12:14336:4096:512", "12:4096:4096:512", "14:4096:14336:512"])
    hip = Hip()
    built = [(p, load_lib(p)) for p in libs]
    ctxs = []
    for p, lib in built:
        ctx = lib.cdna4_init(0)
        if not ctx:
            raise RuntimeError("cdna4_init failed: %s" % lib.cdna4_last_error())
        lib.cdna4_reserve_workspace(ctx, 256 << 20); ctxs.append(ctx)
    orc = ob.Oracle() if a.check else None
    for case in cases:
        t, m, k, n = (int(v) for v in case.split(":"))
        w = random_block_bytes(t, m, k, 1); x = activations(n, k, 2)
        n_rot = max(2, min(64, (320 << 20) // w.nbytes + 1))
        wd = [hip.upload(w) for _ in range(n_rot)]; rot = (P * n_rot)(*wd)
        xd = hip.upload(x); cd = hip.malloc(4 * m * n)
        rs = w.shape[1]
        best = {p: None for p, _ in built}
        for _ in range(a.rounds):                                  # interleaved: A, B, A, B
            for (p, lib), ctx in zip(built, ctxs):
                ms = C.c_float(0)
                rc = lib.cdna4_time_mul_mat(ctx, m, n, k, t, rot, n_rot, rs, xd, k, cd, m, a.warmup, a.iters, None, C.byref(ms))
                if rc != 0:
                    raise RuntimeError("cdna4_time_mul_mat rc %d: %s" % (rc, lib.cdna4_last_error()))
                best[p] = ms.value if best[p] is None else min(best[p], ms.value)
        for (p, lib), ctx in zip(built, ctxs):
            us = best[p] * 1e3
            rec = {"case": case, "type": ob.NAMES.get(t, str(t)), "lib": os.path.relpath(p, ROOT), "us": round(us, 3), "rotating_copies": n_rot}
            if n <= 8:
                rec["gbs"] = round(w.nbytes / us / 1e3, 1); rec["frac_hbm"] = round(w.nbytes / us / 1e3 / HBM_PEAK_GBS, 4)
            else:
                tf = 2.0 * m * k * n / us / 1e6; rec["tflops"] = round(tf, 1); rec["frac_mfma"] = round(tf / MFMA_F16_PEAK_TFLOPS, 4)
            if orc is not None:
                hip.check(hip.h.hipMemset(cd, 0, 4 * m * n), "hipMemset")
                rc = lib.cdna4_mul_mat(ctx, m, n, k, t, wd[0], rs, 0, xd, k, cd, m, None); hip.check(hip.h.hipDeviceSynchronize(), "sync")
                got = hip.download(cd, (n, m), np.float32); want = orc.mul_mat(t, w, x)
                rec["nmse_vs_oracle"] = float(np.sum((got - want) ** 2) / max(float(np.sum(want.astype(np.float64) ** 2)), 1e-300)); rec["rc"] = rc
            print(json.dumps(rec), flush=True)
        for d in wd + [xd, cd]:
            hip.h.hipFree(d)
    for spec in a.op or []:
        run_op(spec, hip, built, ctxs, a)
    for (p, lib), ctx in zip(built, ctxs):
        lib.cdna4_free(ctx)


if __name__ == "__main__":
    main()
