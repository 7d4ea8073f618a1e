"""Stand-alone driver (child process of tests/test_shim_host_logic.py, started with LD_PRELOAD=<tests/fake_hip build>): consistency of the shim's supports_op with the C-ABI entry
points.  The shim aborts the process when an entry point refuses a node that supports_op accepted (outside a stream capture there is no other backend left to run it), so the two
must agree: randomized one-op graphs -- types, shapes, broadcast dims, strided / permuted / offset views, parameter values at and beyond the edges -- are offered to supports_op, and
every accepted one is computed (kernels do nothing on the stand-in runtime; the entry points' argument validation is what runs).  The case being computed is printed first, so an
abort names it.  Exit code 0 = every accepted node was also accepted by its entry point.

    python tests/shim_fuzz_case.py [n_per_op=150] [seed=1]"""
import ctypes as C
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE); sys.path.insert(0, os.path.dirname(HERE))
from oracle import bindings as ob  # noqa: E402

F32, F16, I32, BF16 = 0, 1, 26, 30
WTYPES = [ob.Q4_K, ob.Q5_K, ob.Q6_K, ob.IQ4_NL, ob.IQ2_S, ob.IQ3_S, ob.Q4_0, ob.Q8_0, ob.Q5_0, ob.Q4_1, ob.Q6_0, ob.Q2_K, ob.Q3_K, ob.IQ4_XS, ob.IQ2_XXS, ob.IQ2_XS, ob.IQ3_XXS, ob.IQ2_K, ob.IQ3_K, ob.IQ4_K, ob.IQ5_K,
          ob.IQ4_KS, ob.IQ5_KS, ob.IQ2_KS, ob.IQ3_KS, ob.IQ4_KSS, ob.IQ2_KL, ob.IQ6_K, ob.IQ1_S, ob.IQ1_M, ob.MXFP4, ob.IQ1_BN, ob.IQ2_BN, ob.IQ1_KT, ob.IQ2_KT, ob.IQ3_KT, ob.IQ4_KT, F32, F16, BF16, 15, 9]      # + Q8_K, Q8_1: not weight types


def main():
    n_per_op = int(sys.argv[1]) if len(sys.argv) > 1 else 150; seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    from ggml_host import GgmlHost
    h = GgmlHost(); g = h.g; rng = np.random.default_rng(seed)
    assert h.shim.ggml_backend_cuda_get_device_count() >= 1
    gpu = h.shim.ggml_backend_cuda_init(0, None, None); assert gpu
    g.ggml_blck_size.restype = C.c_int64; g.ggml_blck_size.argtypes = [C.c_int]
    g.ggml_view_4d.restype = C.c_void_p; g.ggml_view_4d.argtypes = [C.c_void_p, C.c_void_p] + [C.c_int64] * 4 + [C.c_size_t] * 4
    g.ggml_mul_mat_id.restype = C.c_void_p
    g.ggml_backend_buffer_clear.restype = None; g.ggml_backend_buffer_clear.argtypes = [C.c_void_p, C.c_uint8]
    stats = {}

    def pick(*v):
        return v[int(rng.integers(0, len(v)))]

    def offer(kind, desc, build):
        """build(ctx) -> output tensor; counts (offered, accepted) per op kind, computes accepted nodes"""
        ctx = g.ggml_init(h.ref.InitParams(g.ggml_tensor_overhead() * 32 + g.ggml_graph_overhead() + (1 << 16), None, True))
        out = build(ctx)
        st = stats.setdefault(kind, [0, 0]); st[0] += 1
        if out:
            buf = g.ggml_backend_alloc_ctx_tensors(ctx, gpu)
            if buf:
                g.ggml_backend_buffer_clear(buf, 0)         # on a real GPU the kernels run: zero weights, zero row / expert ids (valid everywhere)
            if buf and g.ggml_backend_supports_op(gpu, out):
                st[1] += 1
                print("computing %s %s" % (kind, json.dumps(desc)), flush=True)
                gf = g.ggml_new_graph(ctx); g.ggml_build_forward_expand(gf, out)
                rc = g.ggml_backend_graph_compute(gpu, gf)
                assert rc == 0, (kind, desc, rc)
            if buf:
                g.ggml_backend_buffer_free(buf)
        g.ggml_free(ctx)

    def t4(ctx, t, ne):
        return g.ggml_new_tensor_4d(ctx, t, *ne)

    for _ in range(n_per_op):
        # MUL_MAT: weight types x shapes (row lengths off the block grid are refused by ggml itself: keep multiples of the block size), broadcast over ne2 / ne3, column counts
        t = pick(*WTYPES); bs = max(1, g.ggml_blck_size(t)); k = int(bs * pick(1, 2, 3, 4, 8, 16, 17)) if bs > 1 else pick(64, 96, 128, 1000, 4096)
        m, n = pick(1, 2, 4, 6, 64, 66, 256), pick(1, 2, 5, 8, 9, 33, 64); w2, w3, r2, r3 = pick(1, 1, 2), pick(1, 1, 3), pick(1, 2), pick(1, 2)
        d = dict(t=t, k=k, m=m, n=n, w=[w2, w3], r=[r2, r3])
        offer("MUL_MAT", d, lambda ctx: g.ggml_mul_mat(ctx, t4(ctx, t, [k, m, w2, w3]), t4(ctx, pick(F32, F32, F32, F16), [k, n, w2 * r2, w3 * r3])))
        # MUL_MAT_ID
        t = pick(*WTYPES[:38]); bs = max(1, g.ggml_blck_size(t)); k = int(bs * pick(1, 2, 4)); m, ne_, nu, ntok = pick(4, 64, 66), pick(1, 4, 8), pick(1, 2), pick(1, 3, 40)
        d = dict(t=t, k=k, m=m, n_expert=ne_, n_used=nu, n_tok=ntok)
        offer("MUL_MAT_ID", d, lambda ctx: g.ggml_mul_mat_id(ctx, g.ggml_new_tensor_3d(ctx, t, k, m, ne_), g.ggml_new_tensor_3d(ctx, F32, k, pick(1, nu), ntok), g.ggml_new_tensor_2d(ctx, I32, nu, ntok)))
        # ROPE: modes, rotated dims (odd, beyond the head, zero), strided views
        hd, nh, ntok = pick(64, 128, 96, 127), pick(1, 4, 32), pick(1, 5, 64); mode = pick(0, 2, 2, 0); nd = pick(hd, hd // 2, hd - 1, hd + 2, 0, 2)         # (NORM / NEOX; the multi-section modes have constructor asserts of their own)
        d = dict(hd=hd, nh=nh, ntok=ntok, mode=mode, n_dims=nd, view=False)

        def rope(ctx, hd=hd, nh=nh, ntok=ntok, mode=mode, nd=nd):
            x = g.ggml_new_tensor_3d(ctx, pick(F32, F32, F16), hd, nh, ntok); p = g.ggml_new_tensor_1d(ctx, I32, ntok)
            if nd <= 0 or nd > hd or (mode & 1):          # (ggml asserts on these itself)
                return None
            return g.ggml_rope_ext(ctx, x, p, None, nd, mode, 8192, 10000.0, 1.0, 0.0, 1.0, 32.0, 1.0)
        offer("ROPE", d, rope)
        # SOFT_MAX: mask types / shapes (narrower, shorter, broadcast), non-contiguous input
        ne0, ne1, ne2 = pick(1, 31, 64, 1000), pick(1, 7, 32), pick(1, 4); mt = pick(None, F16, F32); mne0, mne1 = pick(ne0, ne0 + 32, max(1, ne0 - 1)), pick(ne1, ne1 + 31, max(1, ne1 - 1))
        d = dict(ne=[ne0, ne1, ne2], mask=mt, mne=[mne0, mne1])

        def soft_max(ctx, ne0=ne0, ne1=ne1, ne2=ne2, mt=mt, mne0=mne0, mne1=mne1):
            x = g.ggml_new_tensor_3d(ctx, F32, ne0, ne1, ne2)
            if mt is not None and (mne0 < ne0 or mne1 < ne1):
                return None                                 # (ggml_soft_max_ext asserts mask->ne[0] == a->ne[0] / ne[1] >= a->ne[1] itself)
            msk = g.ggml_new_tensor_2d(ctx, mt, ne0, mne1) if mt is not None else None
            return g.ggml_soft_max_ext(ctx, x, msk, 0.125, pick(0.0, 8.0) if msk else 0.0)
        offer("SOFT_MAX", d, soft_max)
        # FLASH_ATTN_EXT: head sizes, GQA ratios, contexts, token counts, K / V as views of a wider cache (offsets, strides), mask padding
        hd, nhkv, gq, ntok, nkv = pick(64, 128, 256, 96), pick(1, 2, 8), pick(1, 4, 8), pick(1, 2, 33), pick(1, 32, 100, 256, 1500)
        off = pick(0, 0, 1, 3); d = dict(hd=hd, nhkv=nhkv, gqa=gq, ntok=ntok, nkv=nkv, kv_row_offset=off)

        def fa(ctx, hd=hd, nhkv=nhkv, gq=gq, ntok=ntok, nkv=nkv, off=off):
            nh = nhkv * gq
            q = g.ggml_permute(ctx, g.ggml_new_tensor_3d(ctx, F32, hd, nh, ntok), 0, 2, 1, 3)
            kc = g.ggml_new_tensor_2d(ctx, F16, hd * nhkv, nkv + off); vc = g.ggml_new_tensor_2d(ctx, F16, hd * nhkv, nkv + off)
            row = hd * nhkv * 2
            kk = g.ggml_view_3d(ctx, kc, hd, nkv, nhkv, row, hd * 2, off * row); vv = g.ggml_view_3d(ctx, vc, hd, nkv, nhkv, row, hd * 2, off * row)
            npad = (ntok + 31) // 32 * 32
            msk = g.ggml_new_tensor_2d(ctx, F16, (nkv + 31) // 32 * 32 if pick(0, 1) else nkv, npad)
            return g.ggml_flash_attn_ext(ctx, q, kk, vv, msk, 0.088, pick(0.0, 8.0), pick(0.0, 30.0))
        offer("FLASH_ATTN_EXT", d, fa)
        # ARGSORT / SUM_ROWS / GET_ROWS
        ne0, ne1 = pick(1, 8, 60, 64, 128, 20000), pick(1, 5)
        offer("ARGSORT", dict(ne=[ne0, ne1]), lambda ctx: g.ggml_argsort(ctx, g.ggml_new_tensor_2d(ctx, F32, ne0, ne1), pick(0, 1)))
        offer("SUM_ROWS", dict(ne=[ne0, ne1]), lambda ctx: g.ggml_sum_rows(ctx, g.ggml_new_tensor_2d(ctx, pick(F32, F32, F16), ne0, ne1)))
        t = pick(*WTYPES); bs = max(1, g.ggml_blck_size(t)); k = int(bs * pick(1, 2, 5)); nv = pick(1, 7, 96)
        offer("GET_ROWS", dict(t=t, k=k, n_vocab=nv), lambda ctx: g.ggml_get_rows(ctx, g.ggml_new_tensor_2d(ctx, t, k, nv), g.ggml_new_tensor_1d(ctx, I32, pick(1, 13))))
        # ADD / MUL / DIV with broadcast shapes and mixed types, RMS_NORM / FUSED_RMS_NORM, CPY between f32 / f16 incl. permuted sources
        a_ne = [pick(1, 32, 4096), pick(1, 5), pick(1, 3), pick(1, 2)]; b_ne = [pick(a_ne[0], 1), pick(a_ne[1], 1), pick(a_ne[2], 1), pick(a_ne[3], 1)]; opn = pick("add", "mul", "div")
        offer(opn.upper(), dict(a=a_ne, b=b_ne), lambda ctx: getattr(g, "ggml_" + opn)(ctx, t4(ctx, pick(F32, F32, F16), a_ne), t4(ctx, pick(F32, F32, F16), b_ne)))
        ne0 = pick(1, 64, 1000, 4096, 8192, 20000); fused = pick(0, 1)
        offer("RMS_NORM", dict(ne0=ne0, fused=fused), lambda ctx: (g.ggml_fused_rms_norm(ctx, g.ggml_new_tensor_2d(ctx, pick(F32, F32, F16), ne0, 3), g.ggml_new_tensor_1d(ctx, F32, ne0), 1e-5) if fused
                                                                    else g.ggml_rms_norm(ctx, g.ggml_new_tensor_2d(ctx, F32, ne0, 3), 1e-5)))
        st_, dt_ = pick(F32, F16), pick(F32, F16, ob.Q8_0); c0, c1, c2 = pick(32, 64, 100), pick(1, 4), pick(1, 3); perm = pick(0, 1)

        def cpy(ctx, st_=st_, dt_=dt_, c0=c0, c1=c1, c2=c2, perm=perm):
            s = g.ggml_new_tensor_3d(ctx, st_, c0, c1, c2)
            if perm:
                s = g.ggml_permute(ctx, s, 0, 2, 1, 3)
            return g.ggml_cpy(ctx, s, g.ggml_new_tensor_1d(ctx, dt_, c0 * c1 * c2))
        offer("CPY", dict(src=st_, dst=dt_, ne=[c0, c1, c2], permuted=perm), cpy)
    print(json.dumps({"offered / accepted per op": stats}))
    g.ggml_backend_free(gpu)
    return 0


if __name__ == "__main__":
    sys.exit(main())
