#!/usr/bin/env python3
"""Generates tests/golden/iqk_golden.npz by running the REAL reference library (oracle/_ref, built from /root/reference
by oracle/Makefile) on small seeded inputs.  Only runs in the build container; the .npz is committed and travels.

    python tests/golden/make_golden.py

Contents per weight type T (base and _R4): w_T (quantized bytes from the reference quantizer [+ oracle repack, itself
pinned against the reference's R4 dequantizer]), wb_T (random-byte blocks), deq_T / deqb_T (reference to_float),
mm_T_n{1,2,8} (reference iqk_mul_mat results for x[:n]).  Plus x (f32 activations incl. an outlier row) and the
reference's Q8_2_X4 / Q8_K / Q8_K32 quantizations of x."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import bindings as ob          # noqa: E402
from common import activations, gaussian_weights_f32, random_block_bytes   # noqa: E402

M, K = 16, 512


def main():
    ref = ob.Ref(); orc = ob.Oracle()
    out = {"meta": np.array([M, K]), "ref_variant": np.array(ref.variant)}
    x = activations(8, K, 42, outliers=False); x[1] = activations(1, K, 43, outliers=True)[0]; x[2, :64] = 0
    out["x"] = x
    for vdt in (ob.Q8_2_X4, ob.Q8_K, ob.Q8_K32):
        out["xq_%d" % vdt] = ref.quantize_activations(vdt, x)
    for t in ob.BASE_TYPES:
        w = ref.quantize(t, gaussian_weights_f32(M, K, 1000 + t)); wb = random_block_bytes(t, M, K, 2000 + t)
        for tt, ww, wwb in ((t, w, wb), (ob.R4_OF[t], orc.repack_r4(t, w, K), orc.repack_r4(t, wb, K))):
            out["w_%d" % tt] = ww; out["wb_%d" % tt] = wwb
            out["deq_%d" % tt] = ref.dequantize(tt, ww, K); out["deqb_%d" % tt] = ref.dequantize(tt, wwb, K)
            for n in (1, 2, 8):
                out["mm_%d_n%d" % (tt, n)] = ref.mul_mat(tt, ww, x[:n])
    path = os.path.join(ROOT, "tests", "golden", "iqk_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes; reference variant:", ref.variant)
    # ---- the further weight types: same contents per type, in a file of their own.  Activations `xs` for the types whose reference AVX-512 kernel
    # saturates int16 pair sums on full-range int8 activations (tests/test_oracle_vs_ref.py SATURATING): one outlier per 256 sets the block scale, the other int8
    # values stay small, so the reference's kernel and its exact form coincide and the golden vector pins the EXACT arithmetic.
    f3 = {"meta": np.array([M, K]), "ref_variant": np.array(ref.variant)}
    xs = activations(8, K, 44, outliers=True)
    f3["x"] = x; f3["xs"] = xs
    saturating = (ob.IQ4_XS, ob.IQ4_K, ob.IQ5_K, ob.IQ4_KS, ob.IQ5_KS, ob.IQ4_KSS, ob.IQ6_K)
    for t in ob.LEGACY_TYPES:
        w = ref.quantize(t, gaussian_weights_f32(M, K, 3000 + t)); wb = random_block_bytes(t, M, K, 4000 + t)
        f3["w_%d" % t] = w; f3["wb_%d" % t] = wb
        f3["deq_%d" % t] = ref.dequantize(t, w, K); f3["deqb_%d" % t] = ref.dequantize(t, wb, K)
        xx = xs if t in saturating else x
        for n in (1, 2, 8):
            f3["mm_%d_n%d" % (t, n)] = ref.mul_mat(t, w, xx[:n])
    path = os.path.join(ROOT, "tests", "golden", "iqk_golden_f3.npz")
    np.savez_compressed(path, **f3)
    print("wrote", path, os.path.getsize(path), "bytes")
    # ---- trellis types: a file of their own (real quantizer output: 4 rows -- the trellis search is slow --, random-bit blocks: 16 rows)
    kt = {"meta": np.array([M, K]), "ref_variant": np.array(ref.variant), "x": x}
    for t in ob.KT_TYPES:
        w = ref.quantize(t, gaussian_weights_f32(4, K, 5000 + t)); wb = random_block_bytes(t, M, K, 6000 + t)
        kt["w_%d" % t] = w; kt["wb_%d" % t] = wb
        kt["deq_%d" % t] = ref.dequantize(t, w, K); kt["deqb_%d" % t] = ref.dequantize(t, wb, K)
        for n in (1, 2, 8):
            kt["mm_%d_n%d" % (t, n)] = ref.mul_mat(t, w, x[:n]); kt["mmb_%d_n%d" % (t, n)] = ref.mul_mat(t, wb, x[:n])
    path = os.path.join(ROOT, "tests", "golden", "iqk_golden_kt.npz")
    np.savez_compressed(path, **kt)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
