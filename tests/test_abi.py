"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every symbol the header declares,
type traits match the reference's block geometry, and entry points fail loudly (no fallback) without a GPU."""
import os
import re

import pytest

from conftest import ROOT, load_package


def header_symbols():
    text = open(os.path.join(ROOT, "include", "ggml_hip_cdna4.h")).read()
    return sorted(set(re.findall(r"CDNA4_API[^;(]*?\b(cdna4_\w+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    pkg = load_package()
    lib = pkg.load_library()
    syms = header_symbols()
    assert len(syms) >= 27
    for s in syms:
        assert hasattr(lib, s), "missing export %s" % s
    # and the Python binding covers exactly the declared set
    from ik_llama_cpp_amd.cdna4 import SIGNATURES
    assert sorted(SIGNATURES) == syms


def header_prototypes():
    """{symbol: number of parameters} parsed from the header's prototypes (a lone `void` parameter list counts 0)."""
    text = open(os.path.join(ROOT, "include", "ggml_hip_cdna4.h")).read()
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    out = {}
    for m in re.finditer(r"CDNA4_API[^;(]*?\b(cdna4_\w+)\s*\(([^;]*?)\)\s*;", text, flags=re.S):
        params = m.group(2).strip()
        out[m.group(1)] = 0 if params in ("", "void") else params.count(",") + 1
    return out


def test_binding_argument_counts_match_the_header_prototypes():
    # cdna4_op_norm_rope_store_kv was bound with 27 argtypes against a 26-parameter prototype and nothing noticed
    load_package()
    from ik_llama_cpp_amd.cdna4 import SIGNATURES
    protos = header_prototypes()
    assert sorted(protos) == header_symbols()
    bad = {s: (len(SIGNATURES[s][1]), n) for s, n in protos.items() if len(SIGNATURES[s][1]) != n}
    assert not bad, "binding argtypes vs header parameters: %r" % bad


def test_type_traits_match_reference_block_geometry():
    # reference: ggml/src/ggml-common.h:348-353,367-373,388-394,468-474,503-510,586-590 ; ggml.c:4808-4811
    pkg = load_package(); lib = pkg.load_library()
    expect = {12: (256, 144), 13: (256, 176), 14: (256, 210), 20: (32, 18), 22: (256, 82), 21: (256, 110)}
    for t, (bs, ts) in expect.items():
        for tt in (t, pkg.R4_OF[t]):
            assert lib.cdna4_type_supported(tt) == 1
            assert lib.cdna4_blck_size(tt) == bs and lib.cdna4_type_size(tt) == ts
            assert lib.cdna4_row_size(tt, 4096) == ts * 4096 // bs == pkg.row_size(tt, 4096)
    assert lib.cdna4_row_size(12, 4096) == 2304 and lib.cdna4_row_size(12, 14336) == 8064
    assert lib.cdna4_type_supported(23) == 1 and lib.cdna4_blck_size(23) == 256 and lib.cdna4_type_size(23) == 136 and lib.cdna4_vec_dot_type(23) == 15      # IQ4_XS
    for t, (bs, ts, vd) in {6: (32, 22, 99), 16: (256, 66, 15), 17: (256, 74, 15), 18: (256, 98, 15), 3: (32, 20, 99), 7: (32, 24, 99), 133: (32, 26, 99), 10: (256, 84, 15), 11: (256, 110, 15),
                          137: (256, 76, 15), 138: (256, 110, 15), 139: (256, 144, 15), 140: (256, 176, 15), 144: (256, 136, 15), 152: (256, 168, 15), 145: (256, 70, 15), 156: (256, 102, 15), 146: (256, 128, 15), 157: (256, 86, 15), 141: (256, 212, 15), 19: (256, 50, 15), 29: (256, 56, 15), 39: (32, 17, 99)}.items():
        # Q5_0, IQ2_XXS, IQ2_XS, IQ3_XXS, Q4_1, Q5_1, Q6_0, Q2_K, Q3_K, IQ2_K ... IQ5_K, IQ4_KS, IQ5_KS (decode kernels + f16 prompt route)
        assert lib.cdna4_type_supported(t) == 1 and lib.cdna4_blck_size(t) == bs and lib.cdna4_type_size(t) == ts and lib.cdna4_vec_dot_type(t) == vd
    assert lib.cdna4_row_size(144, 4096) == 4 + 16 * 136 == pkg.row_size(144, 4096) and lib.cdna4_row_size(152, 512) == 4 + 2 * 168 and lib.cdna4_row_size(145, 512) == 2 + 2 * 70      # f32 row scale of the _KS types
    for t, (bs, ts) in {2: (32, 18), 8: (32, 34)}.items():       # legacy 32-block types Q4_0 / Q8_0 (ggml-common.h block_q4_0, block_q8_0)
        assert lib.cdna4_type_supported(t) == 1 and lib.cdna4_blck_size(t) == bs and lib.cdna4_type_size(t) == ts and lib.cdna4_vec_dot_type(t) == 99
    assert lib.cdna4_type_supported(9) == 0          # Q8_1 is not a weight type of the path
    # vec_dot_type table (ggml.c:963-1053,1837-1857)
    assert [lib.cdna4_vec_dot_type(t) for t in (12, 13, 14, 20, 22, 21, 212, 213, 214, 220, 222, 221)] == \
           [99, 99, 99, 99, 15, 15, 148, 148, 15, 99, 15, 15]


def test_no_cpu_fallback_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    pkg = load_package()
    with pytest.raises(RuntimeError):
        pkg.Cdna4Backend(0)
    lib = pkg.load_library()
    assert lib.cdna4_init(0) is None
    assert b"invalid device" in lib.cdna4_last_error()


def test_missing_library_fails_loudly(tmp_path):
    pkg = load_package()
    with pytest.raises((FileNotFoundError, OSError)):
        pkg.load_library(str(tmp_path / "nope.so"))
