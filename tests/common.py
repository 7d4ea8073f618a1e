"""Shared test helpers: deterministic synthetic weights / activations (SURVEY 8d recipe) and tolerances."""
import numpy as np

from oracle import bindings as ob

# parity bars (BASELINE.md section 2 / SURVEY 8c)
TOL_INT8_PATH = 2e-6      # |gpu - oracle| / sum|w*x| for the int8-activation (decode) kernels: same integers, f32 order differs
TOL_FP_ACCUM = 1e-3       # north_star: "within 1e-3 relative on the fp accumulate" (vs fp64 accumulate, relative to sum|w*x|)
NMSE_VS_CPU = 5e-4        # reference's own backend-op tolerance (tests/test-backend-ops.cpp:979-981)

D_OFFSETS = {  # (offset of fp16 fields inside one block) used to sanitise random-byte blocks
    ob.Q4_K: [0, 2], ob.Q5_K: [0, 2], ob.Q6_K: [208], ob.IQ4_NL: [0], ob.IQ2_S: [0], ob.IQ3_S: [0], ob.Q4_0: [0], ob.Q8_0: [0], ob.IQ4_XS: [0],
    ob.Q5_0: [0], ob.IQ2_XXS: [0], ob.IQ2_XS: [0], ob.IQ3_XXS: [0],
    ob.Q4_1: [0, 2], ob.Q5_1: [0, 2], ob.Q6_0: [0], ob.Q2_K: [80, 82], ob.Q3_K: [108],
    ob.IQ2_K: [0], ob.IQ3_K: [0], ob.IQ4_K: [0], ob.IQ5_K: [0], ob.IQ4_KS: [], ob.IQ5_KS: [], ob.IQ2_KS: [], ob.IQ3_KS: [], ob.IQ4_KSS: [], ob.IQ2_KL: [], ob.IQ6_K: [0], ob.IQ1_S: [0], ob.IQ1_M: [], ob.MXFP4: [], ob.IQ1_BN: [], ob.IQ2_BN: [], ob.IQ1_KT: [], ob.IQ2_KT: [], ob.IQ3_KT: [], ob.IQ4_KT: [],
}


def random_block_bytes(t, m, k, seed):
    """Every byte uniformly random (covers all bit patterns of scales / codebook indices / signs); the fp16
    super-block scales are replaced by finite values in [-0.02, 0.02]."""
    rng = np.random.default_rng(seed)
    rs = ob.row_size(t, k); ts = ob.TYPE_SIZE[t]; meta = ob.ROW_META.get(t, 0)
    w = rng.integers(0, 256, size=(m, rs), dtype=np.uint8)
    if meta == 4 and t in ob.KT_TYPES:      # trellis types: row scales of real quantizer output on N(0, 0.02^2) weights sit around 1e-4 ... 1e-3 (values reach +-126 x +-127)
        w[:, :4] = (rng.uniform(1e-4, 6e-4, size=(m, 1)) * rng.choice([-1.0, 1.0], size=(m, 1))).astype(np.float32).view(np.uint8)
    elif meta == 4:     # f32 row scale in front of the blocks (IQ4_KS, IQ5_KS)
        w[:, :4] = rng.uniform(-2e-4, 2e-4, size=(m, 1)).astype(np.float32).view(np.uint8)
    elif meta == 2:   # f16 row scale (IQ2_KS, IQ3_KS)
        w[:, :2] = rng.uniform(-2e-3, 2e-3, size=(m, 1)).astype(np.float16).view(np.uint8)
    blocks = w[:, meta:].reshape(m, (rs - meta) // ts, ts)
    for off in D_OFFSETS[t]:
        d = rng.uniform(-0.02, 0.02, size=blocks.shape[:2]).astype(np.float16)
        if off == 2:
            d = np.abs(d)
        blocks[:, :, off:off + 2] = d.view(np.uint8).reshape(m, (rs - meta) // ts, 2)
    if t == ob.MXFP4:       # E8M0 exponent byte: keep the block scales in 2^-20 .. 2^-5
        blocks[:, :, 0] = rng.integers(108, 123, size=blocks.shape[:2], dtype=np.uint8)
    if t == ob.IQ1_M:       # the f16 super-block scale lives in the top nibbles of the four scale words (iq1m_scale_t)
        d = rng.uniform(-0.02, 0.02, size=blocks.shape[:2]).astype(np.float16).view(np.uint16)
        for i in range(4):
            blocks[:, :, 49 + 2 * i] = (blocks[:, :, 49 + 2 * i] & 0x0f) | (((d >> (4 * i)) & 0xf) << 4).astype(np.uint8)
    return w


def gaussian_weights_f32(m, k, seed):
    return (np.random.default_rng(seed).standard_normal((m, k)) * 0.02).astype(np.float32)


def activations(n, k, seed, outliers=False):
    x = np.random.default_rng(seed).standard_normal((n, k)).astype(np.float32)
    if outliers:                      # one 1e3 outlier per 256 (SURVEY 8d "adversarial set")
        x[:, ::256] = 1e3 * np.sign(x[:, ::256])
    return x


def make_weights(t, m, k, seed, oracle, ref=None):
    """Quantized weights of (possibly _R4) type t.  With `ref` they are real quantizer output of N(0, 0.02^2); otherwise random bytes."""
    base = ob.BASE_OF.get(t, t)
    w = ref.quantize(base, gaussian_weights_f32(m, k, seed)) if ref is not None else random_block_bytes(base, m, k, seed)
    if t in ob.BASE_OF:
        w = oracle.repack_r4(base, w, k)
    return w


def rel_err_vs_terms(a, b, sum_abs_terms):
    return float(np.max(np.abs(a.astype(np.float64) - b.astype(np.float64)) / np.maximum(sum_abs_terms, 1e-30)))


def nmse(a, ref):
    a = a.astype(np.float64); ref = ref.astype(np.float64)
    return float(np.sum((a - ref) ** 2) / max(np.sum(ref ** 2), 1e-300))


def np_up_gate_combine(op, up, gate, up_b=None, gate_b=None, limit=0.0):
    """float64 statement of the fused up*gate epilogue (mul_mat_up_gate_NxM, iqk_mul_mat.cpp:146-175); biases broadcast on the last axis."""
    g = gate.astype(np.float64) + (0 if gate_b is None else gate_b.astype(np.float64))
    u = up.astype(np.float64) + (0 if up_b is None else up_b.astype(np.float64))
    if op == 6:
        t = np.maximum(g, 0)
    elif op == 10:
        t = g * 0.5 * (1 + np.tanh(0.5 * g))
    elif op == 15:
        t = 0.5 * g * (1 + np.tanh(0.797884560802865 * g * (1 + 0.044715 * g * g)))
    elif op == 14:
        xi = np.minimum(g, 7.0); t = xi * 0.5 * (1 + np.tanh(0.5 * 1.702 * xi))
    else:
        raise ValueError(op)
    if limit > 1e-6:
        t = np.minimum(t, limit)
    if op == 14:
        u = 1 + np.clip(u, -7.0, 7.0)
    elif limit > 1e-6:
        u = np.clip(u, -limit, limit)
    return u * t
