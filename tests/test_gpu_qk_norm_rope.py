"""-m gpu: cdna4_op_norm_rope_store_kv -- FUSED_RMS_NORM(q) + ROPE(q) + FUSED_RMS_NORM(k) + ROPE(k) + CPY(k -> f16 K cache) + CPY(v -> f16 V cache) of an attention block with
per-head norms (llm_build_mul_mat_qkv with q_norm / k_norm, llama-build-context.cpp:2481-2490: Qwen3, ...) as ONE launch (csrc/ops.hip norm_rope_store_kv_kernel) -- against the six
launches it replaces, through the C ABI: rotated Q, rotated K and both cache rows must be BIT-IDENTICAL (the sums of squares and the rotations run in the order of the single kernels),
and the result must be float64 arithmetic within f32 rounding.  Q / K / V are slices of one fused q,k,v result or three separate mat-mul results, one token up to a prompt batch, with
and without the slot indirection of a captured graph, in place (rotated Q over un-normed Q, as the graph allocator places it) and apart."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scripts"))

pytestmark = pytest.mark.gpu
P, I, L64, F = C.c_void_p, C.c_int, C.c_long, C.c_float
ROPE = dict(n_ctx_orig=40960, base=1000000.0, scale=1.0, ext=0.0, attn=1.0, fast=32.0, slow=1.0)


@pytest.fixture(scope="module")
def env():
    import nt_bench as nb
    hip = nb.Hip(); lib = nb.load_lib(os.path.join(ROOT, "ik_llama.cpp_amd", "libggml-hip-cdna4.so"))
    TP = C.POINTER(nb.Tensor)
    lib.cdna4_op_norm_rope_store_kv.argtypes = [P, TP, TP, F, TP, TP, TP, F, TP, TP, P, TP, TP, P, P, P, I, I, I, F, F, F, F, F, F, P]
    lib.cdna4_op_rms_norm.argtypes = [P, TP, TP, F, TP, P]
    lib.cdna4_op_rope.argtypes = [P, TP, P, P, TP, I, I, I, F, F, F, F, F, F, P]
    lib.cdna4_op_rope_cache.argtypes = [P, P, L64, P, I, I, F, F, F, F, F, F, P]
    lib.cdna4_op_rope_cache_reset.argtypes = [P]
    lib.cdna4_op_cpy_indirect.argtypes = [P, TP, TP, P, P]
    ctx = lib.cdna4_init(0)
    assert ctx, lib.cdna4_last_error()
    yield nb, hip, lib, ctx
    lib.cdna4_free(ctx)


class _IntPointers:
    """the Hip helper of scripts/nt_bench.py with device addresses as plain ints (slices of one allocation are address arithmetic)"""
    def __init__(self, hip):
        self.hip = hip; self.h = hip.h; self.check = hip.check; self.download = hip.download

    def malloc(self, n):
        return self.hip.malloc(n).value

    def upload(self, arr):
        return self.hip.upload(arr).value


def rope_f64(x, pos, mode, base):
    """x [tok, head, hd] (float64) rotated over the whole head: NORM pairs (2i, 2i+1), NEOX pairs (i, i + hd/2); ggml_rope_cache_init without YaRN"""
    hd = x.shape[-1]; half = hd // 2
    theta = pos[:, None].astype(np.float64) * (np.float32(base) ** (-2.0 / hd)).astype(np.float64) ** np.arange(half)[None, :]
    c, s = np.cos(theta)[:, None, :], np.sin(theta)[:, None, :]
    a, b = (x[..., :half], x[..., half:]) if mode == 2 else (x[..., 0::2], x[..., 1::2])
    y = np.empty_like(x)
    if mode == 2:
        y[..., :half] = a * c - b * s; y[..., half:] = a * s + b * c
    else:
        y[..., 0::2] = a * c - b * s; y[..., 1::2] = a * s + b * c
    return y


@pytest.mark.parametrize("in_place", [False, True], ids=["apart", "in_place"])
@pytest.mark.parametrize("layout", ["fused_qkv", "separate"])
@pytest.mark.parametrize("mode", [0, 2], ids=["norm", "neox"])
@pytest.mark.parametrize("hd,n_head,n_head_kv,n_tok", [(128, 16, 8, 1), (128, 16, 8, 7), (128, 32, 8, 512), (64, 12, 4, 3), (256, 8, 2, 5), (128, 5, 1, 2)])
def test_norm_rope_store_kv_matches_the_six_launches_bit_for_bit(hd, n_head, n_head_kv, n_tok, mode, layout, in_place, env):
    if in_place and layout == "fused_qkv":
        pytest.skip("the allocator never places a rotated [hd, heads, tokens] block over a slice of the wider q,k,v rows")
    nb, hip_, lib, ctx = env
    hip = _IntPointers(hip_)
    nq, nk = hd * n_head, hd * n_head_kv; n_ctx, head = n_tok + 40, 17
    rng = np.random.default_rng(hd + n_tok + mode)
    row = nq + 2 * nk if layout == "fused_qkv" else 0
    if layout == "fused_qkv":
        src = (rng.standard_normal((n_tok, row)) * 2.5).astype(np.float32); sd = hip.upload(src)
        qp, kp, vp = sd, sd + 4 * nq, sd + 4 * (nq + nk); q_row = k_row = v_row = 4 * row
        xq, xk, xv = src[:, :nq], src[:, nq:nq + nk], src[:, nq + nk:]
    else:
        xq = (rng.standard_normal((n_tok, nq)) * 2.5).astype(np.float32); xk = (rng.standard_normal((n_tok, nk)) * 0.3).astype(np.float32); xv = rng.standard_normal((n_tok, nk)).astype(np.float32)
        qp, kp, vp = hip.upload(xq), hip.upload(xk), hip.upload(xv); q_row, k_row, v_row = 4 * nq, 4 * nk, 4 * nk
    wq = (1 + 0.2 * rng.standard_normal(hd)).astype(np.float32); wk = (1 + 0.2 * rng.standard_normal(hd)).astype(np.float32); pos = (np.arange(n_tok) + head).astype(np.int32)
    wqd, wkd, posd = hip.upload(wq), hip.upload(wk), hip.upload(pos)
    eps_q, eps_k = 1e-6, 2e-6

    def t3(p, heads, rowb):      # [hd, heads, n_tok] f32 rows of a (possibly wider) token row
        t = nb.tensor(p, 0, [hd, heads, n_tok, 1], 4); t.nb[2] = rowb; t.nb[3] = rowb * n_tok; return t
    tq, tk = t3(qp, n_head, q_row), t3(kp, n_head_kv, k_row)
    tv = nb.tensor(vp, 0, [nk, n_tok, 1, 1], 4); tv.nb[1] = v_row; tv.nb[2] = tv.nb[3] = v_row * n_tok
    twq, twk = nb.tensor(wqd, 0, [hd, 1, 1, 1], 4), nb.tensor(wkd, 0, [hd, 1, 1, 1], 4)
    rp = (hd, mode, ROPE["n_ctx_orig"], ROPE["base"], ROPE["scale"], ROPE["ext"], ROPE["attn"], ROPE["fast"], ROPE["slow"])
    assert lib.cdna4_op_rope_cache(ctx, posd, n_tok, None, hd, ROPE["n_ctx_orig"], ROPE["base"], ROPE["scale"], ROPE["ext"], ROPE["attn"], ROPE["fast"], ROPE["slow"], None) == 0, lib.cdna4_last_error()

    # ---- the six launches (never in place: they are the reference)
    nqd, nkd, rq1, rk1 = hip.malloc(4 * nq * n_tok), hip.malloc(4 * nk * n_tok), hip.malloc(4 * nq * n_tok), hip.malloc(4 * nk * n_tok)
    kc1, vc1, kc2, vc2 = (hip.malloc(2 * nk * n_ctx) for _ in range(4))
    for b in (kc1, vc1, kc2, vc2):
        hip.check(hip.h.hipMemset(b, 0x5a, 2 * nk * n_ctx), "memset")
    tnq, tnk, trq1, trk1 = t3(nqd, n_head, 4 * nq), t3(nkd, n_head_kv, 4 * nk), t3(rq1, n_head, 4 * nq), t3(rk1, n_head_kv, 4 * nk)
    view = lambda base: nb.tensor(base + 2 * nk * head, 1, [nk, n_tok, 1, 1], 2)
    assert lib.cdna4_op_rms_norm(ctx, C.byref(tq), C.byref(twq), eps_q, C.byref(tnq), None) == 0, lib.cdna4_last_error()
    assert lib.cdna4_op_rope(ctx, C.byref(tnq), posd, None, C.byref(trq1), *rp, None) == 0, lib.cdna4_last_error()
    assert lib.cdna4_op_rms_norm(ctx, C.byref(tk), C.byref(twk), eps_k, C.byref(tnk), None) == 0, lib.cdna4_last_error()
    assert lib.cdna4_op_rope(ctx, C.byref(tnk), posd, None, C.byref(trk1), *rp, None) == 0, lib.cdna4_last_error()
    tkr = nb.tensor(rk1, 0, [nk, n_tok, 1, 1], 4)
    assert lib.cdna4_op_cpy_indirect(ctx, C.byref(tkr), C.byref(view(kc1)), None, None) == 0, lib.cdna4_last_error()
    assert lib.cdna4_op_cpy_indirect(ctx, C.byref(tv), C.byref(view(vc1)), None, None) == 0, lib.cdna4_last_error()
    hip.check(hip.h.hipDeviceSynchronize(), "sync")
    want_q, want_k = hip.download(rq1, (n_tok, nq), np.float32), hip.download(rk1, (n_tok, nk), np.float32)
    want_kc, want_vc = hip.download(kc1, (n_ctx, nk), np.uint16), hip.download(vc1, (n_ctx, nk), np.uint16)

    # ---- the one launch: once with the rotated K in f32 and direct cache addresses, once without it through slots (a captured graph's form)
    slots = hip.upload(np.array([kc2 + 2 * nk * head, vc2 + 2 * nk * head], np.uint64))
    for use_slots in (False, True):
        if in_place and layout == "separate":
            qd2, kd2 = qp, kp; qrow2, krow2 = q_row, k_row          # rotated rows over the un-normed ones (each lane writes what it read)
        else:
            qd2, kd2 = hip.malloc(4 * nq * n_tok), hip.malloc(4 * nk * n_tok); qrow2, krow2 = 4 * nq, 4 * nk
            hip.check(hip.h.hipMemset(kd2, 0xff, 4 * nk * n_tok), "memset")
        trq2, trk2 = t3(qd2, n_head, qrow2), t3(kd2, n_head_kv, krow2)
        for b in (kc2, vc2):
            hip.check(hip.h.hipMemset(b, 0x5a, 2 * nk * n_ctx), "memset")
        fake = nb.tensor(kc2, 1, [nk, n_tok, 1, 1], 2), nb.tensor(vc2, 1, [nk, n_tok, 1, 1], 2)         # with slots the tensors' own addresses must not be used: row 0 stays untouched
        tkc, tvc = (fake if use_slots else (view(kc2), view(vc2)))
        rc = lib.cdna4_op_norm_rope_store_kv(ctx, C.byref(tq), C.byref(twq), eps_q, C.byref(trq2), C.byref(tk), C.byref(twk), eps_k, None if use_slots else C.byref(trk2), C.byref(tkc),
                                             slots if use_slots else None, C.byref(tv), C.byref(tvc), slots + 8 if use_slots else None, posd, None, *rp, None)
        assert rc == 0, lib.cdna4_last_error()
        hip.check(hip.h.hipDeviceSynchronize(), "sync")
        got_q = hip.download(qd2, (n_tok, qrow2 // 4), np.float32)[:, :nq]
        np.testing.assert_array_equal(got_q.view(np.uint32), want_q.view(np.uint32), err_msg="rotated Q")
        got_k = hip.download(kd2, (n_tok, krow2 // 4), np.float32)[:, :nk]
        if not use_slots:
            np.testing.assert_array_equal(got_k.view(np.uint32), want_k.view(np.uint32), err_msg="rotated K")
        elif not (in_place and layout == "separate"):
            assert np.all(got_k.view(np.uint32) == 0xffffffff), "the rotated K was written although no reader was declared"
        np.testing.assert_array_equal(hip.download(kc2, (n_ctx, nk), np.uint16), want_kc, err_msg="K cache")
        np.testing.assert_array_equal(hip.download(vc2, (n_ctx, nk), np.uint16), want_vc, err_msg="V cache")
        if in_place and layout == "separate":
            break           # (the inputs are gone)
    # ---- and the arithmetic: float64 norm + rotation
    def f64(x, w, eps, heads):
        x = x.astype(np.float64).reshape(n_tok, heads, hd)
        n = x / np.sqrt(np.mean(x * x, axis=-1, keepdims=True) + eps) * w.astype(np.float64)
        return rope_f64(n, pos, mode, ROPE["base"]).reshape(n_tok, heads * hd)
    eq, ek = f64(xq, wq, eps_q, n_head), f64(xk, wk, eps_k, n_head_kv)
    assert np.sum((want_q - eq) ** 2) / np.sum(eq ** 2) < 1e-9 and np.sum((want_k - ek) ** 2) / np.sum(ek ** 2) < 1e-9
    kc = want_kc[head:head + n_tok].view(np.float16).astype(np.float64)
    assert np.max(np.abs(kc - ek)) <= 2 ** -10 * np.max(np.abs(ek)) * 1.01 and np.all(want_kc[:head] == 0x5a5a) and np.all(want_kc[head + n_tok:] == 0x5a5a)
    np.testing.assert_array_equal(want_vc[head:head + n_tok].view(np.float16), xv.astype(np.float16))


def test_norm_rope_store_kv_declines_without_launching(env):
    """partial rotation, a head size the kernel has no lane grouping for, a stale rope cache: CDNA4_E_UNSUPPORTED (-1) and nothing written"""
    nb, hip_, lib, ctx = env
    hip = _IntPointers(hip_)

    def call(hd, n_dims, fresh_cache=True, n_head=4, n_head_kv=2, n_tok=2):
        nq, nk = hd * n_head, hd * n_head_kv
        qd, kd, vd, od = hip.malloc(4 * nq * n_tok), hip.malloc(4 * nk * n_tok), hip.malloc(4 * nk * n_tok), hip.malloc(4 * nq * n_tok)
        kc, vc = hip.malloc(2 * nk * n_tok), hip.malloc(2 * nk * n_tok); w = hip.upload(np.ones(hd, np.float32)); posd = hip.upload(np.arange(n_tok, dtype=np.int32))
        hip.check(hip.h.hipMemset(od, 0x11, 4 * nq * n_tok), "memset")
        if fresh_cache:
            assert lib.cdna4_op_rope_cache(ctx, posd, n_tok, None, n_dims, 4096, 10000.0, 1.0, 0.0, 1.0, 32.0, 1.0, None) == 0
        else:
            assert lib.cdna4_op_rope_cache_reset(ctx) == 0
        t3 = lambda p, heads: nb.tensor(p, 0, [hd, heads, n_tok, 1], 4)
        tw = nb.tensor(w, 0, [hd, 1, 1, 1], 4); tkc, tvc = nb.tensor(kc, 1, [nk, n_tok, 1, 1], 2), nb.tensor(vc, 1, [nk, n_tok, 1, 1], 2); tv = nb.tensor(vd, 0, [nk, n_tok, 1, 1], 4)
        rc = lib.cdna4_op_norm_rope_store_kv(ctx, C.byref(t3(qd, n_head)), C.byref(tw), 1e-6, C.byref(t3(od, n_head)), C.byref(t3(kd, n_head_kv)), C.byref(tw), 1e-6, None, C.byref(tkc), None,
                                             C.byref(tv), C.byref(tvc), None, posd, None, n_dims, 2 if n_dims == hd else 0, 4096, 10000.0, 1.0, 0.0, 1.0, 32.0, 1.0, None)
        hip.check(hip.h.hipDeviceSynchronize(), "sync")
        untouched = bool(np.all(hip.download(od, (nq * n_tok,), np.uint32) == 0x11111111))
        for d in (qd, kd, vd, od, kc, vc, w, posd):
            hip.h.hipFree(d)
        return rc, untouched
    assert call(128, 128) == (0, False)
    assert call(128, 64) == (-1, True)           # partial rotation
    assert call(96, 96) == (-1, True)            # 24 lanes per row: no DPP grouping
    assert call(128, 128, fresh_cache=False) == (-1, True)
