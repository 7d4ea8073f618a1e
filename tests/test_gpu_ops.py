"""-m gpu: the non-mat-mul ops of a Llama / Mixtral decode + prompt graph (norm, rope, KV copy, attention, router) through the
backend shim, one-op graphs built with the reference's own graph API and compared with the reference CPU backend on the same inputs -- the
`ggml_backend_compare_graph_backend` flow of tests/test-backend-ops.cpp (test_rms_norm :1297, test_rope :1420, test_cpy :1132, test_get_rows :905,
test_soft_max :1370, test_flash_attn_ext :1765, test_argsort :1600, test_sum_rows :1650).

Bars: f32 element-wise arithmetic (add / mul / div / cpy / get_rows / argsort / mul_multi_add) is bit-exact or 1 ulp (division); reductions
(rms_norm, soft_max, sum_rows, dense mul_mat) differ only by summation order: NMSE < 1e-10; rope uses device sinf / cosf: NMSE < 1e-9;
flash attention: q, scores, probabilities and accumulators in f32 over the f16 K / V (the CPU kernels round some of these to f16 for batches):
NMSE < 1e-5 vs the CPU backend and never further from exact f64 attention than the CPU backend is."""
import ctypes as C
import os

import numpy as np
import pytest

from common import nmse
from oracle import bindings as ob

pytestmark = pytest.mark.gpu
F32, F16, I32 = 0, 1, 26


@pytest.fixture(scope="module")
def host():
    from ggml_host import SHIM, GgmlHost
    if ob.ref_path() is None or not os.path.exists(SHIM):
        pytest.skip("needs oracle/_ref (reference libggml) and the prebuilt backend shim")
    h = GgmlHost()
    cpu_only = os.environ.get("TEST_OPS_CPU_DRY_RUN")        # no GPU: run every case on the reference CPU backend twice (checks the cases themselves)
    gpu = h.g.ggml_backend_cpu_init() if cpu_only else h.shim.ggml_backend_cuda_init(0, None, None); cpu = h.g.ggml_backend_cpu_init(); h.g.ggml_backend_cpu_set_n_threads(cpu, 8)
    yield h, gpu, cpu
    h.g.ggml_backend_free(gpu); h.g.ggml_backend_free(cpu)


def both(host, build, inputs):
    h, gpu, cpu = host
    got, sup = h.run(gpu, build, inputs); want, _ = h.run(cpu, build, inputs)
    assert sup, "the backend declined the op"
    return got, want


def new(h, ctx, t, *ne):
    return getattr(h.g, "ggml_new_tensor_%dd" % len(ne))(ctx, t, *ne)


def rnd(seed, *shape):
    return np.random.default_rng(seed).standard_normal(shape).astype(np.float32)


@pytest.mark.parametrize("op", ["add", "mul", "div"])
@pytest.mark.parametrize("a_ne,b_ne", [((4096, 7), (4096, 7)), ((4096, 7), (4096, 1)), ((64, 5, 3), (64, 1, 1)), ((64, 5, 3), (64, 5, 1)), ((1, 2, 9), (1, 1, 9)), ((33, 3), (33, 3))])
def test_binary_broadcast(op, a_ne, b_ne, host):
    h = host[0]
    a = rnd(1, *a_ne[::-1]); b = rnd(2, *b_ne[::-1]) + (3.0 if op == "div" else 0.0)

    def build(ctx):
        ta = new(h, ctx, F32, *a_ne); tb = new(h, ctx, F32, *b_ne)
        return {"a": ta, "b": tb}, getattr(h.g, "ggml_" + op)(ctx, ta, tb)
    got, want = both(host, build, {"a": a, "b": b})
    if op == "div":
        np.testing.assert_allclose(got, want, rtol=2e-7, atol=0)
    else:
        np.testing.assert_array_equal(got, want)


@pytest.mark.parametrize("fused", [False, True])
@pytest.mark.parametrize("ne", [(4096, 7), (128, 8, 5), (512, 1), (1000, 3)])
def test_rms_norm(fused, ne, host):
    h = host[0]
    x = rnd(3, *ne[::-1]) * 3; w = 1 + 0.1 * rnd(4, ne[0])

    def build(ctx):
        tx = new(h, ctx, F32, *ne); tw = new(h, ctx, F32, ne[0])
        if fused:
            return {"x": tx, "w": tw}, h.g.ggml_fused_rms_norm(ctx, tx, tw, 1e-5)
        return {"x": tx}, h.g.ggml_rms_norm(ctx, tx, 1e-5)
    got, want = both(host, build, {"x": x, "w": w} if fused else {"x": x})
    assert nmse(got, want) < 1e-10


@pytest.mark.parametrize("mode,n_dims,ff,ext", [(0, 128, False, 0.0), (0, 64, False, 0.0), (2, 128, False, 0.0), (0, 128, True, 0.0), (0, 128, False, 1.0), (2, 128, True, 1.0),
                                                (2, 64, False, 0.0), (2, 32, True, 0.0), (2, 96, False, 1.0), (2, 2, False, 0.0)])
def test_rope(mode, n_dims, ff, ext, host):
    """x [head_dim, n_head, n_tok] as the Q / K of a Llama layer (mode 0) or a NEOX-style model (mode 2); Llama-3 freq factors; YaRN (ext_factor 1); partial rotation in both
    modes (round 6: NEOX with n_dims < head size -- Phi-2 rotates 32 of 80, GPT-NeoX 25 % of the head: pairs (i, i + n_dims / 2), the rest copied, ggml-cuda/rope.cu:156-243)"""
    h = host[0]
    hd, n_head, n_tok = 128, 8, 37
    x = rnd(5, n_tok, n_head, hd); pos = (np.arange(n_tok) * 53 + 11).astype(np.int32); fac = (1 + 7 * np.random.default_rng(6).random(n_dims // 2)).astype(np.float32)

    def build(ctx):
        tx = new(h, ctx, F32, hd, n_head, n_tok); tp = new(h, ctx, I32, n_tok); tf = new(h, ctx, F32, n_dims // 2)
        t = {"x": tx, "p": tp}
        if ff:
            t["f"] = tf
        return t, h.g.ggml_rope_ext(ctx, tx, tp, tf if ff else None, n_dims, mode, 8192, 500000.0, 0.25 if ext else 1.0, ext, 1.0, 32.0, 1.0)
    inp = {"x": x, "p": pos}
    if ff:
        inp["f"] = fac
    got, want = both(host, build, inp)
    assert nmse(got, want) < 1e-9
    np.testing.assert_allclose(got, want, atol=2e-4 * np.abs(want).max())      # large angles: device sinf vs libm differ by a few ulp of the ARGUMENT


def test_rope_on_a_strided_view(host):
    """the Q / K slices of a fused QKV result: rows of a wider tensor"""
    h = host[0]
    hd, n_head, n_tok = 128, 4, 9
    x = rnd(7, n_tok, 3 * n_head * hd); pos = np.arange(n_tok, dtype=np.int32) + 100

    def build(ctx):
        tx = new(h, ctx, F32, 3 * n_head * hd, n_tok); tp = new(h, ctx, I32, n_tok)
        v = h.g.ggml_view_3d(ctx, tx, hd, n_head, n_tok, hd * 4, 3 * n_head * hd * 4, n_head * hd * 4)      # the K third
        return {"x": tx, "p": tp}, h.g.ggml_rope_ext(ctx, v, tp, None, hd, 0, 8192, 10000.0, 1.0, 0.0, 1.0, 32.0, 1.0)
    got, want = both(host, build, {"x": x, "p": pos})
    assert nmse(got, want) < 1e-9


def test_cpy_f32_into_a_q8_0_cache_view_and_attention_on_it(host):
    """-ctk q8_0 -ctv q8_0 at the op level (round 6): CPY of f32 K / V rows into Q8_0 cache views (blocks of 32: d = amax / 127 as f16, q = round(x / d) -- the bytes of the CPU
    backend's quantize_row_q8_0 up to rounding ties), then FLASH_ATTN_EXT on the Q8_0 views (de-quantized into f16 copies by the library) against the CPU backend on the same graph."""
    h = host[0]
    hd, n_head, n_head_kv, n_ctx, n_tok, head = 128, 8, 2, 256, 5, 64
    nk = hd * n_head_kv
    Q8 = 8
    xk = rnd(81, n_ctx, nk); xv = rnd(82, n_ctx, nk); xq = rnd(83, n_tok, n_head, hd)
    mask = np.zeros((32, n_ctx), np.float16); mask[:, head + n_tok:] = -np.inf

    def build(ctx):
        tk = new(h, ctx, F32, nk, n_ctx); tv = new(h, ctx, F32, nk, n_ctx); tq = new(h, ctx, F32, hd, n_head, n_tok); tm = new(h, ctx, F16, n_ctx, 32)
        kc = new(h, ctx, Q8, nk, n_ctx); vc = new(h, ctx, Q8, nk, n_ctx)
        ck = h.g.ggml_cpy(ctx, tk, kc); cv = h.g.ggml_cpy(ctx, tv, vc)
        row = nk // 32 * 34
        k3 = h.g.ggml_view_3d(ctx, ck, hd, n_ctx, n_head_kv, row, hd // 32 * 34, 0); v3 = h.g.ggml_view_3d(ctx, cv, hd, n_ctx, n_head_kv, row, hd // 32 * 34, 0)
        qp = h.g.ggml_permute(ctx, tq, 0, 2, 1, 3)
        fa = h.g.ggml_flash_attn_ext(ctx, qp, k3, v3, tm, 1.0 / np.sqrt(hd), 0.0, 0.0)
        return {"k": tk, "v": tv, "q": tq, "m": tm}, [ck, fa]
    (gk, gfa), (wk, wfa) = both(host, build, {"k": xk, "v": xv, "q": xq, "m": mask})
    a = gk.view(np.uint8).reshape(-1, 34); b = wk.view(np.uint8).reshape(-1, 34)
    assert np.array_equal(a[:, :2], b[:, :2])                                   # the block scales: identical f16 bits
    dq = np.abs(a[:, 2:].astype(np.int8).astype(np.int32) - b[:, 2:].astype(np.int8).astype(np.int32))
    assert dq.max() <= 1 and (dq > 0).mean() < 1e-3                              # quants: at most a rounding tie apart
    # attention: float64 on the de-quantized cache the DEVICE wrote (the truth for its own bytes), and the CPU backend (whose Q8_0 attention also quantizes q: ~3e-5 from exact)
    blocks = a.reshape(n_ctx, nk // 32, 34)
    dsc = blocks[:, :, :2].copy().view(np.float16).astype(np.float64)[..., 0]; kq = blocks[:, :, 2:].astype(np.int8).astype(np.float64)
    Kd = (dsc[:, :, None] * kq).reshape(n_ctx, n_head_kv, hd)
    # (V is read back through a second graph output in `both`; here the K cache stands for both layouts -- V's bytes follow the same kernel) -> attention truth needs V too:
    xvq = xv.reshape(n_ctx, nk // 32, 32); amax = np.abs(xvq).max(axis=2); dv = (amax / 127.0).astype(np.float32)
    with np.errstate(divide="ignore", invalid="ignore"):
        qv = np.rint(np.where(dv[:, :, None] > 0, xvq / dv[:, :, None], 0.0)).astype(np.float64)
    Vd = (dv.astype(np.float16).astype(np.float64)[:, :, None] * qv).reshape(n_ctx, n_head_kv, hd)
    nvis = head + n_tok; g_ = n_head // n_head_kv; want = np.zeros((n_tok, n_head, hd))
    for t_ in range(n_tok):
        for hh in range(n_head):
            sc_ = Kd[:nvis, hh // g_] @ xq[t_, hh].astype(np.float64) / np.sqrt(hd); p_ = np.exp(sc_ - sc_.max()); p_ /= p_.sum()
            want[t_, hh] = p_ @ Vd[:nvis, hh // g_]
    assert nmse(gfa.reshape(n_tok, n_head, hd), want) < 2e-6
    assert nmse(gfa, wfa) < 2e-4


@pytest.mark.parametrize("st,dt", [(F32, F16), (F32, F32), (F16, F32), (F16, F16)])
def test_cpy_into_a_cache_view(st, dt, host):
    """K-cache write: n_tok rows of n_embd_gqa values land at row `head` of a [n_embd_gqa, n_ctx] cache (llm_build_kv_store); the cache is read back whole"""
    h = host[0]
    n_embd, n_ctx, n_tok, head = 256, 64, 5, 17
    es = {F32: 4, F16: 2}
    src = rnd(8, n_tok, n_embd).astype(np.float16 if st == F16 else np.float32)
    cache = np.zeros((n_ctx, n_embd), np.float16 if dt == F16 else np.float32)

    def build(ctx):
        ts = new(h, ctx, st, n_embd, n_tok); tc = new(h, ctx, dt, n_embd, n_ctx)
        view = h.g.ggml_view_2d(ctx, tc, n_embd, n_tok, n_embd * es[dt], head * n_embd * es[dt])
        return {"s": ts, "c": tc}, h.g.ggml_cpy(ctx, ts, view)
    hh, gpu, cpu = host

    def run(backend):       # read the cache tensor itself after the copy node ran
        import ctypes as C
        g = hh.g
        ctx = g.ggml_init(hh.ref.InitParams(g.ggml_tensor_overhead() * 16 + g.ggml_graph_overhead() + (1 << 16), None, True))
        t, out = build(ctx); gf = g.ggml_new_graph(ctx); g.ggml_build_forward_expand(gf, out)
        buf = g.ggml_backend_alloc_ctx_tensors(ctx, backend)
        for k, a in (("s", src), ("c", cache)):
            g.ggml_backend_tensor_set(t[k], a.ctypes.data_as(C.c_void_p), 0, a.nbytes)
        assert g.ggml_backend_supports_op(backend, out)
        assert g.ggml_backend_graph_compute(backend, gf) == 0
        res = np.empty_like(cache); g.ggml_backend_tensor_get(t["c"], res.ctypes.data_as(C.c_void_p), 0, res.nbytes)
        g.ggml_backend_buffer_free(buf); g.ggml_free(ctx)
        return res
    got, want = run(gpu), run(cpu)
    np.testing.assert_array_equal(got.view(np.uint16 if dt == F16 else np.uint32), want.view(np.uint16 if dt == F16 else np.uint32))
    assert np.any(got[head:head + n_tok] != 0) and not np.any(got[:head]) and not np.any(got[head + n_tok:])


def test_cont_of_a_permuted_tensor(host):
    h = host[0]
    x = rnd(9, 6, 5, 64)

    def build(ctx):
        tx = new(h, ctx, F32, 64, 5, 6)
        return {"x": tx}, h.g.ggml_cont(ctx, h.g.ggml_permute(ctx, tx, 0, 2, 1, 3))
    got, want = both(host, build, {"x": x})
    np.testing.assert_array_equal(got, want)


@pytest.mark.parametrize("t", [F32, F16] + list(ob.BASE_TYPES), ids=lambda t: {F32: "f32", F16: "f16"}.get(t) or ob.NAMES[t])
def test_get_rows(t, host):
    """token-embedding lookup: rows of a (quantized) [n_embd, n_vocab] matrix selected by i32 ids, dequantized to f32"""
    h = host[0]
    n_embd, n_vocab, n_tok = 512, 96, 13
    w32 = rnd(10, n_vocab, n_embd)
    w = w32 if t == F32 else w32.astype(np.float16) if t == F16 else h.ref.quantize(t, w32)
    ids = np.random.default_rng(11).integers(0, n_vocab, n_tok).astype(np.int32)

    def build(ctx):
        tw = new(h, ctx, t, n_embd, n_vocab); ti = new(h, ctx, I32, n_tok)
        return {"w": tw, "i": ti}, h.g.ggml_get_rows(ctx, tw, ti)
    got, want = both(host, build, {"w": w, "i": ids})
    np.testing.assert_array_equal(got, want)


def test_get_rows_3d_router_weights(host):
    """MoE: the routing weights of the selected experts -- probs viewed as [1, n_expert, n_tok], ids [n_used, n_tok] (llm_build_moe_ffn)"""
    h = host[0]
    n_expert, n_used, n_tok = 8, 2, 7
    probs = np.random.default_rng(12).random((n_tok, n_expert)).astype(np.float32)
    ids = np.stack([np.random.default_rng(13 + i).permutation(n_expert)[:n_used] for i in range(n_tok)]).astype(np.int32)

    def build(ctx):
        tp = new(h, ctx, F32, 1, n_expert, n_tok); ti = new(h, ctx, I32, n_used, n_tok)
        return {"p": tp, "i": ti}, h.g.ggml_get_rows(ctx, tp, ti)
    got, want = both(host, build, {"p": probs, "i": ids})
    np.testing.assert_array_equal(got, want)
    np.testing.assert_array_equal(got.reshape(n_tok, n_used), np.take_along_axis(probs, ids, 1))


@pytest.mark.parametrize("ne,mask_t,scale,max_bias", [((8, 5), None, 1.0, 0.0), ((64, 7), None, 0.5, 0.0), ((256, 32, 4), F16, 0.088, 0.0), ((256, 32, 4), F32, 0.088, 0.0),
                                                     ((1000, 3, 8), F16, 0.1, 8.0), ((5000, 2), None, 1.0, 0.0)])
def test_soft_max(ne, mask_t, scale, max_bias, host):
    h = host[0]
    x = rnd(14, *ne[::-1]) * 4
    m = rnd(15, ne[1], ne[0]); m[m > 1.0] = -np.inf; m[:, 0] = 0          # some masked positions, never a fully masked row

    def build(ctx):
        tx = new(h, ctx, F32, *ne); t = {"x": tx}; tm = None
        if mask_t is not None:
            tm = new(h, ctx, mask_t, ne[0], ne[1]); t["m"] = tm
        return t, h.g.ggml_soft_max_ext(ctx, tx, tm, scale, max_bias)
    inp = {"x": x}
    if mask_t is not None:
        inp["m"] = m.astype(np.float16 if mask_t == F16 else np.float32)
    got, want = both(host, build, inp)
    assert nmse(got, want) < 1e-10
    np.testing.assert_allclose(got.reshape(-1, ne[0]).sum(1), 1.0, rtol=1e-5)


@pytest.mark.parametrize("hd,n_head,n_head_kv,n_tok,n_kv,softcap,max_bias", [(64, 8, 2, 1, 256, 0.0, 0.0), (64, 16, 4, 5, 512, 0.0, 0.0), (64, 8, 8, 40, 256, 30.0, 8.0),      # head size 64 (ggml-cuda.cu:5152-5157)
                                                                             (128, 8, 2, 1, 256, 0.0, 0.0), (128, 8, 2, 1, 768, 0.0, 0.0), (128, 4, 4, 19, 256, 0.0, 0.0),
                                                                             (128, 32, 8, 48, 256, 0.0, 0.0), (256, 4, 2, 3, 512, 0.0, 0.0), (128, 8, 2, 5, 256, 30.0, 0.0),
                                                                             (128, 8, 8, 4, 256, 0.0, 8.0), (128, 8, 2, 1, 4096, 0.0, 0.0), (128, 32, 8, 1, 8192, 0.0, 0.0), (128, 8, 8, 2, 1024, 0.0, 0.0), (128, 16, 2, 3, 512, 0.0, 0.0),
                                                                             # prompt batches (>= 16 queries): the matrix-core kernel (csrc/flash_attn.hip)
                                                                             (128, 8, 2, 64, 256, 0.0, 0.0), (128, 4, 1, 130, 512, 0.0, 0.0), (128, 32, 8, 512, 768, 0.0, 0.0),
                                                                             (128, 8, 2, 33, 256, 30.0, 0.0), (128, 8, 8, 40, 320, 0.0, 8.0), (128, 4, 4, 16, 1024, 0.0, 0.0),
                                                                             # the keys of a query block spread over four waves: several chunks of eight blocks per wave, mostly masked / mostly visible
                                                                             (128, 8, 2, 64, 2048, 0.0, 0.0), (128, 4, 4, 32, 1280, 0.0, 0.0), (128, 8, 4, 100, 1536, 20.0, 4.0)])
def test_flash_attn_ext(hd, n_head, n_head_kv, n_tok, n_kv, softcap, max_bias, host):
    """the attention of llm_build_kqv with -fa (n_kv a multiple of 256 as llama.cpp pads it: the reference CPU kernel does not terminate for e.g. n_kv = 592, n_tok = 1): Q f32 permuted to [hd, n_tok, n_head], K / V f16 views of the cache [hd, n_kv, n_head_kv] (strided:
    one cache row holds all KV heads), causal f16 mask padded to GGML_KQ_MASK_PAD rows"""
    h = host[0]
    q = rnd(16, n_tok, n_head, hd); k = rnd(17, n_kv, n_head_kv, hd).astype(np.float16); v = rnd(18, n_kv, n_head_kv, hd).astype(np.float16)
    n_pad = (n_tok + 15) // 16 * 16
    mask = np.zeros((n_pad, n_kv), np.float16); past = n_kv - n_tok - 3        # 3 unused cells at the end of the window
    for t in range(n_tok):
        mask[t, past + t + 1:] = -np.inf
    mask[n_tok:] = -np.inf

    def build(ctx):
        tq = new(h, ctx, F32, hd, n_head, n_tok); tk = new(h, ctx, F16, hd * n_head_kv, n_kv); tv = new(h, ctx, F16, hd * n_head_kv, n_kv); tm = new(h, ctx, F16, n_kv, n_pad)
        qp = h.g.ggml_permute(ctx, tq, 0, 2, 1, 3)
        kv3 = lambda t_: h.g.ggml_view_3d(ctx, t_, hd, n_kv, n_head_kv, hd * n_head_kv * 2, hd * 2, 0)
        return {"q": tq, "k": tk, "v": tv, "m": tm}, h.g.ggml_flash_attn_ext(ctx, qp, kv3(tk), kv3(tv), tm, 1.0 / np.sqrt(hd), max_bias, softcap)
    got, want = both(host, build, {"q": q, "k": k, "v": v, "m": mask})
    # against exact f64 attention first: the bar against the CPU kernels follows from how far THEY are from it (soft-cap + ALiBi over 1536 keys: 3e-5)
    kk = np.repeat(k.astype(np.float64), n_head // n_head_kv, 1); vv = np.repeat(v.astype(np.float64), n_head // n_head_kv, 1)
    s = np.einsum("thd,jhd->htj", q.astype(np.float64), kk) / np.sqrt(hd)
    if softcap:
        s = softcap * np.tanh(s / softcap)
    slope = np.ones(n_head)
    if max_bias:
        nl2 = 1 << int(np.floor(np.log2(n_head))); m0 = 2.0 ** (-max_bias / nl2); m1 = 2.0 ** (-max_bias / 2 / nl2)
        slope = np.array([m0 ** (i + 1) if i < nl2 else m1 ** (2 * (i - nl2) + 1) for i in range(n_head)])
    s = s + slope[:, None, None] * mask[:n_tok].astype(np.float64)[None]
    p = np.exp(s - s.max(-1, keepdims=True)); p /= p.sum(-1, keepdims=True)
    exact = np.einsum("htj,jhd->thd", p, vv).reshape(-1)
    assert nmse(got, exact) < 1e-5, nmse(got, exact)
    assert nmse(got, want) < max(1e-5, 2 * nmse(want, exact)), (nmse(got, want), nmse(want, exact))
    # single rows: f32 throughout; batches: Q and the probabilities are f16 MFMA operands (the CPU kernels round the same two to f16)
    assert nmse(got, exact) <= max(nmse(want, exact) * 1.5, 1e-11 if n_tok < 16 else 2e-7)


@pytest.mark.parametrize("n_kv,visible", [(256, [(0, 5)]), (256, [(0, 40)]), (256, [(0, 100)]), (256, [(0, 256)]), (768, [(0, 600)]), (768, [(300, 420)]), (1024 - 256, [(10, 20), (500, 700)]),
                                          (512, [(448, 512)]), (256, [(17, 18)]), (512, None), (2048, [(1500, 1600)])],
                         ids=["5of256", "40of256", "100of256", "all256", "600of768", "window300-420of768", "two_islands", "last_tile_of512", "one_cell", "no_mask", "split_form_window"])
@pytest.mark.parametrize("n_head,n_head_kv", [(8, 2), (6, 6)])
def test_flash_attn_decode_windows(n_head, n_head_kv, n_kv, visible, host):
    """one decoded token over KV windows of every kind the per-head kernel treats differently (csrc/fa_decode.cuh, round 5): the visible cells a short prefix of the padded window
    (dead tiles are skipped from a bit set built in one look-ahead), a sliding window whose first tiles are dead (a wave's speculative first tile is not the one it needs), two islands,
    a single cell, no mask at all, 4 waves per head (< 512 cells) and 8; the last case takes the split-KV form.  Masked cells of the cache hold NaN-free garbage of large magnitude:
    they must not leak.  Bars: exact f64 attention and the CPU backend, as test_flash_attn_ext."""
    h = host[0]
    hd = 128
    q = rnd(61, 1, n_head, hd); k = rnd(62, n_kv, n_head_kv, hd).astype(np.float16); v = rnd(63, n_kv, n_head_kv, hd).astype(np.float16)
    mask = np.full((16, n_kv), -np.inf, np.float16)
    if visible is None:
        mask[:] = 0
    else:
        for a, b in visible:
            mask[0, a:b] = 0
        dead = np.isinf(mask[0]); k[dead] = (k[dead].astype(np.float32) * 200).astype(np.float16); v[dead] = (v[dead].astype(np.float32) * 200).astype(np.float16)

    def build(ctx):
        tq = new(h, ctx, F32, hd, n_head, 1); tk = new(h, ctx, F16, hd * n_head_kv, n_kv); tv = new(h, ctx, F16, hd * n_head_kv, n_kv)
        qp = h.g.ggml_permute(ctx, tq, 0, 2, 1, 3)
        kv3 = lambda t_: h.g.ggml_view_3d(ctx, t_, hd, n_kv, n_head_kv, hd * n_head_kv * 2, hd * 2, 0)
        inp = {"q": tq, "k": tk, "v": tv}; tm = None
        if visible is not None:
            tm = new(h, ctx, F16, n_kv, 16); inp["m"] = tm
        return inp, h.g.ggml_flash_attn_ext(ctx, qp, kv3(tk), kv3(tv), tm, 1.0 / np.sqrt(hd), 0.0, 0.0)
    inputs = {"q": q, "k": k, "v": v}
    if visible is not None:
        inputs["m"] = mask
    got, want = both(host, build, inputs)
    kk = np.repeat(k.astype(np.float64), n_head // n_head_kv, 1); vv = np.repeat(v.astype(np.float64), n_head // n_head_kv, 1)
    s_ = np.einsum("thd,jhd->htj", q.astype(np.float64), kk) / np.sqrt(hd) + mask[:1].astype(np.float64)[None]
    pr = np.exp(s_ - s_.max(-1, keepdims=True)); pr /= pr.sum(-1, keepdims=True)
    exact = np.einsum("htj,jhd->thd", pr, np.where(np.isinf(mask[0])[:, None, None], 0.0, vv)).reshape(-1)
    assert np.all(np.isfinite(got))
    assert nmse(got, exact) < 1e-10, nmse(got, exact)
    # The reference's CPU kernel is the second bar -- when its own result is usable.  Round 6: for the mask-less 6-head case the CPU backend's row came back with garbage in its
    # last values (5.7e+36, -2.1e+17, 1e-35 ...) in two of three runs of this file while the device result matched the f64 attention to 1e-13: a result that is itself
    # non-finite or off the exact attention by more than 1e-3 cannot judge anything and is reported instead of compared.
    ref_err = nmse(want, exact) if np.all(np.isfinite(want)) else float("inf")
    if not ref_err < 1e-3:
        import warnings
        warnings.warn("reference CPU flash attention returned an unusable row for this case (nmse vs f64 attention: %r); device result checked against the f64 attention only" % (ref_err,))
    else:
        assert nmse(got, want) < max(1e-5, 2 * ref_err), (nmse(got, want), ref_err)


@pytest.mark.parametrize("order", [0, 1])
@pytest.mark.parametrize("ne", [(8, 7), (64, 3), (160, 2), (1000, 4)])
def test_argsort(order, ne, host):
    h = host[0]
    x = np.stack([np.random.default_rng(19 + i).permutation(ne[0]) for i in range(ne[1])]).astype(np.float32) * 0.37      # distinct values: the order is unique

    def build(ctx):
        tx = new(h, ctx, F32, *ne)
        return {"x": tx}, h.g.ggml_argsort(ctx, tx, order)
    got, want = both(host, build, {"x": x})
    np.testing.assert_array_equal(got.view(np.int32), want.view(np.int32))


def test_sum_rows(host):
    h = host[0]
    x = rnd(20, 5, 3, 300)

    def build(ctx):
        tx = new(h, ctx, F32, 300, 3, 5)
        return {"x": tx}, h.g.ggml_sum_rows(ctx, tx)
    got, want = both(host, build, {"x": x})
    np.testing.assert_allclose(got, want, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("n_embd,n_used", [(4096, 2), (4096, 8), (1002, 2), (4096, 9)], ids=["2_experts", "8_experts", "odd_row", "9_experts"])
def test_mul_multi_add(n_embd, n_used, host):
    """the weighted sum of the selected experts' outputs: a [n_embd, n_used, n_tok] * b [1, n_used, n_tok] summed over n_used"""
    h = host[0]
    n_tok = 9
    a = rnd(21, n_tok, n_used, n_embd); b = np.random.default_rng(22).random((n_tok, n_used, 1)).astype(np.float32)

    def build(ctx):
        ta = new(h, ctx, F32, n_embd, n_used, n_tok); tb = new(h, ctx, F32, 1, n_used, n_tok)
        return {"a": ta, "b": tb}, h.g.ggml_mul_multi_add(ctx, ta, tb)
    got, want = both(host, build, {"a": a, "b": b})
    assert nmse(got, want) < 1e-12


@pytest.mark.parametrize("n_tok", [1, 9])
def test_mul_multi_add_plus_residual_one_launch(n_tok, host):
    """the experts' weighted sum followed by the block's residual ADD (llm_build_moe_ffn -> ffn_out + ffn_inp): one launch in the shim (cdna4_op_mul_multi_add_res)"""
    h = host[0]
    n_embd, n_used = 4096, 2
    a = rnd(25, n_tok, n_used, n_embd); b = np.random.default_rng(26).random((n_tok, n_used, 1)).astype(np.float32); r = rnd(27, n_tok, n_embd)

    def build(ctx):
        ta = new(h, ctx, F32, n_embd, n_used, n_tok); tb = new(h, ctx, F32, 1, n_used, n_tok); tr = new(h, ctx, F32, n_embd, n_tok)
        return {"a": ta, "b": tb, "r": tr}, h.g.ggml_add(ctx, h.g.ggml_mul_multi_add(ctx, ta, tb), tr)
    got, want = both(host, build, {"a": a, "b": b, "r": r})
    assert nmse(got, want) < 1e-12


@pytest.mark.parametrize("wt", [F32, F16])
@pytest.mark.parametrize("n", [1, 5, 48])
def test_router_mul_mat(wt, n, host):
    """ffn_gate_inp: an f32 [n_embd, n_expert] weight x f32 activations"""
    h = host[0]
    n_embd, n_expert = 4096, 8
    w = (rnd(23, n_expert, n_embd) / 64).astype(np.float16 if wt == F16 else np.float32); x = rnd(24, n, n_embd)

    def build(ctx):
        tw = new(h, ctx, wt, n_embd, n_expert); tx = new(h, ctx, F32, n_embd, n)
        return {"w": tw, "x": tx}, h.g.ggml_mul_mat(ctx, tw, tx)
    got, want = both(host, build, {"w": w, "x": x})
    assert nmse(got, want) < (1e-10 if wt == F32 else 1e-6)        # f16 weights: the CPU path rounds the activations to f16 as well


def test_add_then_rms_norm_fused_pair(host):
    """the residual add followed by the next norm: the shim runs the two nodes as one kernel (cdna4_op_add_rms_norm); both results are graph outputs here"""
    h = host[0]
    n_embd, n_tok = 4096, 7
    a = rnd(30, n_tok, n_embd); b = rnd(31, n_tok, n_embd); w = 1 + 0.1 * rnd(32, n_embd)

    def build(ctx):
        ta = new(h, ctx, F32, n_embd, n_tok); tb = new(h, ctx, F32, n_embd, n_tok); tw = new(h, ctx, F32, n_embd)
        s = h.g.ggml_add(ctx, ta, tb)
        return {"a": ta, "b": tb, "w": tw}, [s, h.g.ggml_fused_rms_norm(ctx, s, tw, 1e-5)]
    (gs, gn), (ws, wn) = both(host, build, {"a": a, "b": b, "w": w})
    np.testing.assert_array_equal(gs, ws)
    assert nmse(gn, wn) < 1e-10


@pytest.mark.parametrize("n_tok,n_dims", [(1, 128), (7, 128), (512, 128), (33, 64)])
@pytest.mark.parametrize("mode", [0, 2])
def test_rope_q_k_and_kv_cache_store_fused_separate_tensors(n_tok, n_dims, mode, host):
    """the same four nodes on the layout a Llama graph hands over: Qcur / Kcur / Vcur are separate contiguous mat-mul results ([hd, n_head, n_tok] views), 32 / 8 heads of 128,
    a 512-token ubatch among the cases (the (work item, token) grid of rope_store_kv_fast_kernel), partial rotation (n_dims < head size) in both modes (NEOX since round 6: Phi / GPT-NeoX style graphs, ggml-cuda/rope.cu:156-243)"""
    h = host[0]
    hd, n_head, n_head_kv, n_ctx, head = 128, 32, 8, 640, 37
    nq, nk = hd * n_head, hd * n_head_kv
    xq = rnd(71, n_tok, nq); xk = rnd(72, n_tok, nk); xv = rnd(73, n_tok, nk); pos = (np.arange(n_tok) + head).astype(np.int32)

    def build(ctx):
        tq = new(h, ctx, F32, nq, n_tok); tk = new(h, ctx, F32, nk, n_tok); tv = new(h, ctx, F32, nk, n_tok); tp = new(h, ctx, I32, n_tok)
        kc = new(h, ctx, F16, nk, n_ctx); vc = new(h, ctx, F16, nk, n_ctx)
        q = h.g.ggml_view_3d(ctx, tq, hd, n_head, n_tok, hd * 4, nq * 4, 0); k = h.g.ggml_view_3d(ctx, tk, hd, n_head_kv, n_tok, hd * 4, nk * 4, 0)
        rope = lambda x: h.g.ggml_rope_ext(ctx, x, tp, None, n_dims, mode, 8192, 500000.0, 1.0, 0.0, 1.0, 32.0, 1.0)
        qr = rope(q); kr = rope(k)
        ck = h.g.ggml_cpy(ctx, kr, h.g.ggml_view_2d(ctx, kc, nk, n_tok, nk * 2, head * nk * 2))
        cv = h.g.ggml_cpy(ctx, tv, h.g.ggml_view_2d(ctx, vc, nk, n_tok, nk * 2, head * nk * 2))
        return {"q": tq, "k": tk, "v": tv, "p": tp}, [qr, ck, cv]
    (gq, gk, gv), (wq, wk, wv) = both(host, build, {"q": xq, "k": xk, "v": xv, "p": pos})
    assert nmse(gq, wq) < 1e-9
    gk16, wk16 = gk.view(np.float16).astype(np.float32), wk.view(np.float16).astype(np.float32)
    assert nmse(gk16, wk16) < 1e-6 and np.max(np.abs(gk16 - wk16)) <= 2 ** -9 * np.max(np.abs(wk16))
    np.testing.assert_array_equal(gv.view(np.uint16), wv.view(np.uint16))


@pytest.mark.parametrize("n_tok", [1, 5, 64])
@pytest.mark.parametrize("mode", [0, 2])
def test_rope_q_k_and_kv_cache_store_fused(n_tok, mode, host):
    """ROPE(q), ROPE(k), CPY(k -> f16 K cache rows), CPY(v -> f16 V cache rows) as llm_build_kv_store emits them: one launch in the shim
    (cdna4_op_rope_store_kv).  Q / K / V are column slices of one fused QKV result."""
    h = host[0]
    hd, n_head, n_head_kv, n_ctx, head = 128, 8, 2, 96, 11
    nq, nk = hd * n_head, hd * n_head_kv
    qkv = rnd(33, n_tok, nq + 2 * nk); pos = (np.arange(n_tok) + head).astype(np.int32)

    def build(ctx):
        t = new(h, ctx, F32, nq + 2 * nk, n_tok); tp = new(h, ctx, I32, n_tok); kc = new(h, ctx, F16, nk, n_ctx); vc = new(h, ctx, F16, nk, n_ctx)
        row = (nq + 2 * nk) * 4
        q = h.g.ggml_view_3d(ctx, t, hd, n_head, n_tok, hd * 4, row, 0); k = h.g.ggml_view_3d(ctx, t, hd, n_head_kv, n_tok, hd * 4, row, nq * 4)
        v = h.g.ggml_view_2d(ctx, t, nk, n_tok, row, (nq + nk) * 4)
        rope = lambda x: h.g.ggml_rope_ext(ctx, x, tp, None, hd, mode, 8192, 500000.0, 1.0, 0.0, 1.0, 32.0, 1.0)
        qr = rope(q); kr = rope(k)
        ck = h.g.ggml_cpy(ctx, kr, h.g.ggml_view_2d(ctx, kc, nk, n_tok, nk * 2, head * nk * 2))
        cv = h.g.ggml_cpy(ctx, v, h.g.ggml_view_2d(ctx, vc, nk, n_tok, nk * 2, head * nk * 2))
        return {"x": t, "p": tp}, [qr, ck, cv]
    (gq, gk, gv), (wq, wk, wv) = both(host, build, {"x": qkv, "p": pos})
    assert nmse(gq, wq) < 1e-9
    gk16, wk16 = gk.view(np.float16).astype(np.float32), wk.view(np.float16).astype(np.float32)
    assert nmse(gk16, wk16) < 1e-6 and np.max(np.abs(gk16 - wk16)) <= 2 ** -9 * np.max(np.abs(wk16))        # f16 roundings of values that differ in the last f32 bits
    np.testing.assert_array_equal(gv.view(np.uint16), wv.view(np.uint16))


@pytest.mark.parametrize("n_tok", [1, 5, 512])
@pytest.mark.parametrize("mode", [0, 2])
@pytest.mark.parametrize("hd,n_head,n_head_kv", [(128, 16, 8), (64, 12, 4)])
def test_per_head_q_k_norms_rope_and_kv_cache_store_fused(hd, n_head, n_head_kv, mode, n_tok, host):
    """Qwen3-style attention (llm_build_mul_mat_qkv with attn_q_norm / attn_k_norm, then ROPE, then llm_build_kv which expands q, k, v in this order): FUSED_RMS_NORM(q) ROPE(q)
    FUSED_RMS_NORM(k) ROPE(k) CPY(k) CPY(v) -- one launch in the shim (cdna4_op_norm_rope_store_kv; tests/test_gpu_qk_norm_rope.py holds the bit-for-bit comparison with the six)"""
    h = host[0]
    n_ctx, head = n_tok + 64, 23
    nq, nk = hd * n_head, hd * n_head_kv
    xq = rnd(81, n_tok, nq) * 2; xk = rnd(82, n_tok, nk) * 0.5; xv = rnd(83, n_tok, nk); wq = 1 + 0.2 * rnd(84, hd); wk = 1 + 0.2 * rnd(85, hd); pos = (np.arange(n_tok) + head).astype(np.int32)

    def build(ctx):
        tq = new(h, ctx, F32, nq, n_tok); tk = new(h, ctx, F32, nk, n_tok); tv = new(h, ctx, F32, nk, n_tok); tp = new(h, ctx, I32, n_tok); twq = new(h, ctx, F32, hd); twk = new(h, ctx, F32, hd)
        kc = new(h, ctx, F16, nk, n_ctx); vc = new(h, ctx, F16, nk, n_ctx)
        q = h.g.ggml_fused_rms_norm(ctx, h.g.ggml_view_3d(ctx, tq, hd, n_head, n_tok, hd * 4, nq * 4, 0), twq, 1e-6); k = h.g.ggml_fused_rms_norm(ctx, h.g.ggml_view_3d(ctx, tk, hd, n_head_kv, n_tok, hd * 4, nk * 4, 0), twk, 1e-6)
        rope = lambda x: h.g.ggml_rope_ext(ctx, x, tp, None, hd, mode, 40960, 1000000.0, 1.0, 0.0, 1.0, 32.0, 1.0)
        qr = rope(q); kr = rope(k)
        ck = h.g.ggml_cpy(ctx, kr, h.g.ggml_view_2d(ctx, kc, nk, n_tok, nk * 2, head * nk * 2))
        cv = h.g.ggml_cpy(ctx, tv, h.g.ggml_view_2d(ctx, vc, nk, n_tok, nk * 2, head * nk * 2))
        return {"q": tq, "k": tk, "v": tv, "p": tp, "wq": twq, "wk": twk}, [qr, ck, cv]
    (gq, gk, gv), (wq_, wk_, wv_) = both(host, build, {"q": xq, "k": xk, "v": xv, "p": pos, "wq": wq, "wk": wk})
    assert nmse(gq, wq_) < 1e-9
    gk16, wk16 = gk.view(np.float16).astype(np.float32), wk_.view(np.float16).astype(np.float32)
    assert nmse(gk16, wk16) < 1e-6 and np.max(np.abs(gk16 - wk16)) <= 2 ** -9 * np.max(np.abs(wk16))
    np.testing.assert_array_equal(gv.view(np.uint16), wv_.view(np.uint16))


# ---- decode-token fusions across mat-mul boundaries (cdna4_mul_mat_multi_fused / cdna4_fused_up_gate_fused)
@pytest.mark.parametrize("t,m,k", [(ob.Q4_K, 512, 4096), (ob.Q6_K, 256, 1024), (ob.Q4_K, 256, 14336), (ob.IQ4_NL, 128, 2048)], ids=["q4_K", "q6_K", "q4_K_long", "iq4_nl"])
def test_mul_mat_plus_residual_add_one_launch(t, m, k, host):
    """attn_output / ffn_down followed by the residual ADD of one decoded token: the shim folds the ADD into the mat-mul's epilogue"""
    h = host[0]
    w = h.ref.quantize(t, rnd(50, m, k) * 0.02); x = rnd(51, 1, k); r = rnd(52, 1, m)

    def build(ctx):
        tw = new(h, ctx, t, k, m); tx = new(h, ctx, F32, k, 1); tr = new(h, ctx, F32, m, 1)
        return {"w": tw, "x": tx, "r": tr}, h.g.ggml_add(ctx, h.g.ggml_mul_mat(ctx, tw, tx), tr)
    got, want = both(host, build, {"w": w, "x": x, "r": r})
    assert nmse(got, want) < 1e-10


@pytest.mark.parametrize("k", [4096, 1024])
def test_rms_norm_folded_into_qkv_mat_muls(k, host):
    """attn_norm followed by the q, k, v mat-muls of one decoded token (Q4_K, Q4_K, Q6_K as in Q4_K_M): the norm runs in the launch's prologue"""
    h = host[0]
    mq, mk = 512, 128
    wq = h.ref.quantize(ob.Q4_K, rnd(53, mq, k) * 0.02); wk = h.ref.quantize(ob.Q4_K, rnd(54, mk, k) * 0.02); wv = h.ref.quantize(ob.Q6_K, rnd(55, mk, k) * 0.02)
    x = rnd(56, 1, k) * 3; nw = 1 + 0.1 * rnd(57, k)

    def build(ctx):
        tq = new(h, ctx, ob.Q4_K, k, mq); tk = new(h, ctx, ob.Q4_K, k, mk); tv = new(h, ctx, ob.Q6_K, k, mk); tx = new(h, ctx, F32, k, 1); tn = new(h, ctx, F32, k)
        nrm = h.g.ggml_fused_rms_norm(ctx, tx, tn, 1e-5)
        return {"q": tq, "k": tk, "v": tv, "x": tx, "n": tn}, [h.g.ggml_mul_mat(ctx, tq, nrm), h.g.ggml_mul_mat(ctx, tk, nrm), h.g.ggml_mul_mat(ctx, tv, nrm)]
    got, want = both(host, build, {"q": wq, "k": wk, "v": wv, "x": x, "n": nw})
    for a, b in zip(got, want):
        assert nmse(a, b) < 1e-8        # (the row's 1 / rms differs in the last bit -> a few int8 activations round the other way)


@pytest.mark.parametrize("tv", [ob.Q4_K, ob.Q6_K], ids=["v_q4_K", "v_q6_K"])
def test_norm_qkv_rope_kv_store_one_launch(tv, host):
    """one decoded token: FUSED_RMS_NORM -> q, k, v MUL_MATs -> ROPE(q), ROPE(k) -> CPY(k -> K cache), CPY(v -> V cache) is ONE launch in the shim (rotation and f16 cache writes in the
    mat-mul's epilogue, cdna4_fusion.qkv); V in Q6_K takes the two-type-group kernel (Q4_K_M layers with a Q6_K attn_v)"""
    h = host[0]
    for name, res, args in [("ggml_reshape_3d", C.c_void_p, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64])]:
        f = getattr(h.g, name); f.restype = res; f.argtypes = args
    k, hd, n_head, n_head_kv, n_ctx, head = 4096, 128, 8, 2, 32, 5
    nq, nk = hd * n_head, hd * n_head_kv
    wq = h.ref.quantize(ob.Q4_K, rnd(80, nq, k) * 0.02); wk = h.ref.quantize(ob.Q4_K, rnd(81, nk, k) * 0.02); wv = h.ref.quantize(tv, rnd(82, nk, k) * 0.02)
    x = rnd(83, 1, k) * 3; nw = 1 + 0.1 * rnd(84, k); pos = np.array([head], np.int32)

    def build(ctx):
        tq = new(h, ctx, ob.Q4_K, k, nq); tk = new(h, ctx, ob.Q4_K, k, nk); tvv = new(h, ctx, tv, k, nk); tx = new(h, ctx, F32, k, 1); tn = new(h, ctx, F32, k)
        tp = new(h, ctx, I32, 1); kc = new(h, ctx, F16, nk, n_ctx); vc = new(h, ctx, F16, nk, n_ctx)
        nrm = h.g.ggml_fused_rms_norm(ctx, tx, tn, 1e-5)
        q = h.g.ggml_mul_mat(ctx, tq, nrm); kk = h.g.ggml_mul_mat(ctx, tk, nrm); v = h.g.ggml_mul_mat(ctx, tvv, nrm)
        rope = lambda t: h.g.ggml_rope_ext(ctx, t, tp, None, hd, 0, 8192, 500000.0, 1.0, 0.0, 1.0, 32.0, 1.0)
        qr = rope(h.g.ggml_reshape_3d(ctx, q, hd, n_head, 1)); kr = rope(h.g.ggml_reshape_3d(ctx, kk, hd, n_head_kv, 1))
        ck = h.g.ggml_cpy(ctx, kr, h.g.ggml_view_2d(ctx, kc, nk, 1, nk * 2, head * nk * 2))
        cv = h.g.ggml_cpy(ctx, v, h.g.ggml_view_2d(ctx, vc, nk, 1, nk * 2, head * nk * 2))
        # (the three mat-muls are expanded first, as llm_build_context does: adjacent graph nodes; their own results are dead once the chain is fused and are not compared)
        return {"q": tq, "k": tk, "v": tvv, "x": tx, "n": tn, "p": tp}, [q, kk, v, qr, ck, cv]
    (_, _, _, gq, gk, gv), (_, _, _, wq_, wk_, wv_) = both(host, build, {"q": wq, "k": wk, "v": wv, "x": x, "n": nw, "p": pos})
    assert nmse(gq, wq_) < 1e-8        # (bars of test_rms_norm_folded_into_qkv_mat_muls: the row's 1 / rms differs in the last bit)
    for a, b in ((gk, wk_), (gv, wv_)):
        a16, b16 = a.view(np.float16).astype(np.float32), b.view(np.float16).astype(np.float32)
        assert nmse(a16, b16) < 1e-6


def test_rms_norm_folded_into_fused_up_gate(host):
    h = host[0]
    t, m, k = ob.Q4_K, 1024, 4096
    wu = h.ref.quantize(t, rnd(58, m, k) * 0.02); wg = h.ref.quantize(t, rnd(59, m, k) * 0.02); x = rnd(60, 1, k) * 2; nw = 1 + 0.1 * rnd(61, k)

    def build(ctx):
        tu = new(h, ctx, t, k, m); tg = new(h, ctx, t, k, m); tx = new(h, ctx, F32, k, 1); tn = new(h, ctx, F32, k)
        return {"u": tu, "g": tg, "x": tx, "n": tn}, h.g.ggml_fused_up_gate(ctx, tu, tg, h.g.ggml_fused_rms_norm(ctx, tx, tn, 1e-5), 10)
    got, want = both(host, build, {"u": wu, "g": wg, "x": x, "n": nw})
    assert nmse(got, want) < 1e-8


def test_norm_result_with_a_second_consumer_is_still_written(host):
    """the fusion must not fire when the normed row is read by another node (here: also a graph output)"""
    h = host[0]
    t, m, k = ob.Q4_K, 256, 1024
    w = h.ref.quantize(t, rnd(62, m, k) * 0.02); x = rnd(63, 1, k); nw = 1 + 0.1 * rnd(64, k)

    def build(ctx):
        tw = new(h, ctx, t, k, m); tx = new(h, ctx, F32, k, 1); tn = new(h, ctx, F32, k)
        nrm = h.g.ggml_fused_rms_norm(ctx, tx, tn, 1e-5)
        return {"w": tw, "x": tx, "n": tn}, [h.g.ggml_mul_mat(ctx, tw, nrm), h.g.ggml_add(ctx, nrm, nrm)]
    (g0, g1), (w0, w1) = both(host, build, {"w": w, "x": x, "n": nw})
    assert nmse(g0, w0) < 1e-8 and nmse(g1, w1) < 1e-10


def test_moe_router_tied_probabilities_pick_the_reference_experts(host):
    """experts with IDENTICAL router rows have identical probabilities: the reference orders (value, index) pairs with std::greater (iqk_argsort), i.e. the HIGHER
    index first -- the fused router launch and the stand-alone ARGSORT must both do so, or fused and unfused graphs would route tied tokens to different experts."""
    h = host[0]
    n_embd, n_expert, n_used, n_tok = 256, 8, 2, 4
    for name, res, args in [("ggml_soft_max", C.c_void_p, [C.c_void_p, C.c_void_p]), ("ggml_top_k", C.c_void_p, [C.c_void_p, C.c_void_p, C.c_int]),
                            ("ggml_reshape_3d", C.c_void_p, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64]), ("ggml_reshape_2d", C.c_void_p, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64])]:
        f = getattr(h.g, name); f.restype = res; f.argtypes = args
    w = (rnd(80, n_expert, n_embd) / 8).astype(np.float32)
    w[1] = w[6]; w[3] = w[6]; w[0] = w[5]                      # ties: {1, 3, 6} and {0, 5}
    x = rnd(81, n_tok, n_embd)

    def build_fused(ctx):          # the exact chain of llm_build_moe_ffn (ggml_top_k = ARGSORT + VIEW): one cdna4_op_moe_router launch
        tw = new(h, ctx, F32, n_embd, n_expert); tx = new(h, ctx, F32, n_embd, n_tok)
        logits = h.g.ggml_mul_mat(ctx, tw, tx); probs = h.g.ggml_soft_max(ctx, logits)
        sel = h.g.ggml_top_k(ctx, probs, n_used)                 # a [n_used, n_tok] view of the ARGSORT result (row stride n_expert)
        wsel = h.g.ggml_get_rows(ctx, h.g.ggml_reshape_3d(ctx, probs, 1, n_expert, n_tok), sel)
        w2 = h.g.ggml_reshape_2d(ctx, wsel, n_used, n_tok); ws = h.g.ggml_sum_rows(ctx, w2)
        return {"w": tw, "x": tx}, [logits, sel, h.g.ggml_div(ctx, w2, ws)]

    def build_plain(ctx):          # a bare ARGSORT of the same probabilities: argsort_kernel
        tw = new(h, ctx, F32, n_embd, n_expert); tx = new(h, ctx, F32, n_embd, n_tok)
        logits = h.g.ggml_mul_mat(ctx, tw, tx); probs = h.g.ggml_soft_max(ctx, logits)
        return {"w": tw, "x": tx}, [logits, h.g.ggml_argsort(ctx, probs, 1)]
    if os.environ.get("TEST_OPS_CPU_DRY_RUN"):
        return
    (lg, srt_f, _), _ = h.run(host[1], build_fused, {"w": w, "x": x})
    (_, srt_p), _ = h.run(host[1], build_plain, {"w": w, "x": x})
    lg = lg.reshape(n_tok, n_expert)
    for t in range(n_tok):
        assert lg[t, 1] == lg[t, 3] == lg[t, 6] and lg[t, 0] == lg[t, 5]            # the ties are exact
        want = sorted(range(n_expert), key=lambda e: (lg[t, e], e), reverse=True)    # std::greater on (value, index)
        np.testing.assert_array_equal(srt_p.view(np.int32).reshape(n_tok, n_expert)[t], want)
        np.testing.assert_array_equal(srt_f.view(np.int32)[t * n_expert:t * n_expert + n_used], want[:n_used])       # (the strided slab of the view)


@pytest.mark.parametrize("n_embd,n_tok", [(4096, 1), (4096, 3), (1024, 1), (2048, 8)])
def test_rms_norm_rides_in_the_moe_router_launch(n_embd, n_tok, host):
    """llm_build_moe_ffn: ffn_norm's row feeds the router AND the experts.  For up to 8 tokens the shim folds the FUSED_RMS_NORM into the router launch (cdna4_op_moe_router_norm), which
    writes the normed row as well.  The launch sums the row's squares in the partition and order of the stand-alone norm kernel: the normed row, the logits and the selection are
    BIT-IDENTICAL to the two launches -- here: to a graph whose norm stands alone (the router then reads it as an input of a second graph)."""
    h = host[0]
    n_expert, n_used = 8, 2
    for name, res, args in [("ggml_soft_max", C.c_void_p, [C.c_void_p, C.c_void_p]), ("ggml_top_k", C.c_void_p, [C.c_void_p, C.c_void_p, C.c_int]),
                            ("ggml_reshape_3d", C.c_void_p, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64]), ("ggml_reshape_2d", C.c_void_p, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64])]:
        f = getattr(h.g, name); f.restype = res; f.argtypes = args
    w = (rnd(90, n_expert, n_embd) / 8).astype(np.float32); x = rnd(91, n_tok, n_embd) * 3; nw = (1 + 0.1 * rnd(92, n_embd)).astype(np.float32)

    def router(ctx, tw, xn):
        logits = h.g.ggml_mul_mat(ctx, tw, xn); probs = h.g.ggml_soft_max(ctx, logits)
        sel = h.g.ggml_top_k(ctx, probs, n_used)
        wsel = h.g.ggml_get_rows(ctx, h.g.ggml_reshape_3d(ctx, probs, 1, n_expert, n_tok), sel)
        w2 = h.g.ggml_reshape_2d(ctx, wsel, n_used, n_tok); ws = h.g.ggml_sum_rows(ctx, w2)
        return [logits, sel, h.g.ggml_div(ctx, w2, ws)]

    def build_fused(ctx):
        tw = new(h, ctx, F32, n_embd, n_expert); tx = new(h, ctx, F32, n_embd, n_tok); tn = new(h, ctx, F32, n_embd)
        xn = h.g.ggml_fused_rms_norm(ctx, tx, tn, 1e-5)
        return {"w": tw, "x": tx, "n": tn}, [xn] + router(ctx, tw, xn)

    def build_norm(ctx):
        tx = new(h, ctx, F32, n_embd, n_tok); tn = new(h, ctx, F32, n_embd)
        return {"x": tx, "n": tn}, h.g.ggml_fused_rms_norm(ctx, tx, tn, 1e-5)

    def build_router(ctx):
        tw = new(h, ctx, F32, n_embd, n_expert); txn = new(h, ctx, F32, n_embd, n_tok)
        return {"w": tw, "xn": txn}, router(ctx, tw, txn)
    (xn_f, lg_f, sel_f, wn_f), (xn_c, lg_c, _, _) = both(host, build_fused, {"w": w, "x": x, "n": nw})
    assert nmse(xn_f, xn_c) < 1e-10 and nmse(lg_f, lg_c) < 1e-9
    if os.environ.get("TEST_OPS_CPU_DRY_RUN"):
        return
    xn_s, _ = h.run(host[1], build_norm, {"x": x, "n": nw})
    (lg_s, sel_s, wn_s), _ = h.run(host[1], build_router, {"w": w, "xn": xn_s.reshape(n_tok, n_embd)})
    np.testing.assert_array_equal(xn_f.view(np.uint32), xn_s.view(np.uint32), err_msg="normed row")
    np.testing.assert_array_equal(lg_f.view(np.uint32), lg_s.view(np.uint32), err_msg="router logits")
    np.testing.assert_array_equal(sel_f.view(np.int32), sel_s.view(np.int32), err_msg="selected experts")
    np.testing.assert_array_equal(wn_f.view(np.uint32), wn_s.view(np.uint32), err_msg="normalized weights")


@pytest.mark.parametrize("n_expert,n_used,n_tok,wt", [(8, 2, 1, F32), (8, 2, 7, F32), (64, 8, 5, F32), (16, 4, 48, F16), (4, 2, 3, F32)])
def test_moe_router_chain_one_launch(n_expert, n_used, n_tok, wt, host):
    """the router of llm_build_moe_ffn (softmax gating, normalized weights): MUL_MAT(f32) -> SOFT_MAX -> top-k (ARGSORT + view) -> GET_ROWS -> SUM_ROWS -> DIV,
    which the shim runs as one kernel (cdna4_op_moe_router)."""
    h = host[0]
    n_embd = 1024
    for name, res, args in [("ggml_soft_max", C.c_void_p, [C.c_void_p, C.c_void_p]), ("ggml_top_k", C.c_void_p, [C.c_void_p, C.c_void_p, C.c_int]),
                            ("ggml_reshape_3d", C.c_void_p, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64]), ("ggml_reshape_2d", C.c_void_p, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64])]:
        f = getattr(h.g, name); f.restype = res; f.argtypes = args
    w = (rnd(70, n_expert, n_embd) / 8).astype(np.float16 if wt == F16 else np.float32); x = rnd(71, n_tok, n_embd)

    def build(ctx):
        tw = new(h, ctx, wt, n_embd, n_expert); tx = new(h, ctx, F32, n_embd, n_tok)
        logits = h.g.ggml_mul_mat(ctx, tw, tx); probs = h.g.ggml_soft_max(ctx, logits)
        sel = h.g.ggml_top_k(ctx, probs, n_used)
        wsel = h.g.ggml_get_rows(ctx, h.g.ggml_reshape_3d(ctx, probs, 1, n_expert, n_tok), sel)
        w2 = h.g.ggml_reshape_2d(ctx, wsel, n_used, n_tok); ws = h.g.ggml_sum_rows(ctx, w2)
        return {"w": tw, "x": tx}, [logits, probs, wsel, ws, h.g.ggml_div(ctx, w2, ws)]
    got, want = both(host, build, {"w": w, "x": x})
    tol = 1e-10 if wt == F32 else 1e-6
    # The reference CPU backend runs its own fused form of this chain, which is only valid inside the full MoE block (stand-alone it returns e.g.
    # [0.5, 0.5] for the normalized weights): the logits are compared with it, everything after them with numpy.  In model context the chain is
    # covered against the CPU backend by the Mixtral-shaped logits tests (tests/test_gpu_llama.py).
    assert nmse(got[0], want[0]) < tol, nmse(got[0], want[0])
    if os.environ.get("TEST_OPS_CPU_DRY_RUN"):
        return
    lg = got[0].reshape(n_tok, n_expert).astype(np.float64); pr = np.exp(lg - lg.max(1, keepdims=True)); pr /= pr.sum(1, keepdims=True)
    top = np.sort(pr, 1)[:, ::-1][:, :n_used]
    assert nmse(got[1], pr.reshape(-1)) < 1e-10 and nmse(got[2], top.reshape(-1)) < 1e-10 and nmse(got[3], top.sum(1)) < 1e-10
    assert nmse(got[4], (top / top.sum(1, keepdims=True)).reshape(-1)) < 1e-10
    np.testing.assert_allclose(got[4].reshape(n_tok, n_used).sum(1), 1.0, rtol=1e-5)
