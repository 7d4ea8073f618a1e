#!/usr/bin/env python3
"""Generates tests/golden/r4_host_golden.npz: for each of the 18 host re-tiled row-interleaved types, base-type rows (real quantizer output and random-byte blocks)
and the bytes the REAL reference's own repacker (iqk_repack_tensor, iqk_quantize.cpp:8535-8583 -- what `llama-quantize --repack` and -rtr run) makes of them.
Only runs in the build container (needs oracle/_ref); the .npz is committed and travels, so the layout pin does not depend on the reference library being loadable.

    python tests/golden/make_golden_r4_host.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import bindings as ob          # noqa: E402
from common import gaussian_weights_f32, random_block_bytes   # noqa: E402
from test_retile_host import R4_HOST       # noqa: E402

M, K = 16, 512


def main():
    ref = ob.Ref()
    out = {"meta": np.array([M, K]), "ref_variant": np.array(ref.variant)}
    for base, r in R4_HOST.items():
        for tag, w in (("q", ref.quantize(base, gaussian_weights_f32(M, K, 3000 + base))), ("b", random_block_bytes(base, M, K, 4000 + base))):
            new_t, rep = ref.repack_tensor(base, w, K)
            assert new_t == r, (base, new_t, r)
            out["base_%s_%d" % (tag, r)] = w; out["r_%s_%d" % (tag, r)] = rep.reshape(w.shape)
    path = os.path.join(ROOT, "tests", "golden", "r4_host_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes; reference variant:", ref.variant)


if __name__ == "__main__":
    main()
