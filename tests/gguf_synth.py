"""Synthetic GGUF files for driving the UNMODIFIED reference binaries (llama-bench, libllama) through the ggml-hip-cdna4 shim.

A minimal GGUF v3 writer (numpy only; the format is ggml/src/ggml.c gguf_init_from_file: magic, version, counts, KV pairs, tensor infos,
aligned data) plus two model builders:
  * tiny_model():  a few layers of a Llama / Mixtral-shaped model whose weights are N(0, sigma^2) quantized by the REAL reference quantizer
                   (oracle/_ref, test infrastructure) -- well-conditioned logits for the -ngl 99 vs -ngl 0 parity tests;
  * bench_model(): full-size Llama-3-8B / Mixtral shapes with random-bit blocks and controlled scales, streamed to disk tensor by tensor
                   (4.6 GB for Q4_K_M): throughput runs of llama-bench only.
No tokenizer is needed: `tokenizer.ggml.model = "no_vocab"` + `<arch>.vocab_size` (src/llama-vocab.cpp:1759-1778); llama-bench feeds random ids."""
import struct

import numpy as np

F32, F16, Q4_K, Q5_K, Q6_K, IQ4_NL, IQ3_S, IQ2_S = 0, 1, 12, 13, 14, 20, 21, 22
TYPE_SIZE = {F32: 4, F16: 2, Q4_K: 144, Q5_K: 176, Q6_K: 210, IQ4_NL: 18, IQ3_S: 110, IQ2_S: 82}
BLCK = {F32: 1, F16: 1, Q4_K: 256, Q5_K: 256, Q6_K: 256, IQ4_NL: 32, IQ3_S: 256, IQ2_S: 256}
D_OFFS = {Q4_K: (0, 2), Q5_K: (0, 2), Q6_K: (208,), IQ4_NL: (0,), IQ3_S: (0,), IQ2_S: (0,)}
ROW_META = {}          # bytes in front of a row's blocks (type traits row_meta_size): the _KS / _KL types keep a row scale there


def add_types(ob):
    """make every weight type of oracle.bindings known to the writer (tiny_model() with real quantizer output only)"""
    for t in ob.LEGACY_TYPES + ob.KT_TYPES:
        TYPE_SIZE[t] = ob.TYPE_SIZE[t]; BLCK[t] = ob.BLCK[t]
    ROW_META.update(ob.ROW_META)
ALIGN = 32
# GGUF value types
T_U32, T_F32, T_STR, T_ARR, T_U64 = 4, 6, 8, 9, 10


def _s(x):
    b = x.encode(); return struct.pack("<Q", len(b)) + b


def _kv(key, val):
    if isinstance(val, str):
        return _s(key) + struct.pack("<I", T_STR) + _s(val)
    if isinstance(val, float):
        return _s(key) + struct.pack("<If", T_F32, val)
    if isinstance(val, int):
        return _s(key) + struct.pack("<II", T_U32, val)
    if isinstance(val, list):        # arrays of str / float / int (GGUF element types STRING 8, FLOAT32 6, INT32 5): the tokenizer tables of spm_vocab()
        head = _s(key) + struct.pack("<I", T_ARR)
        if isinstance(val[0], str):
            return head + struct.pack("<IQ", T_STR, len(val)) + b"".join(_s(v) for v in val)
        if isinstance(val[0], float):
            return head + struct.pack("<IQ", T_F32, len(val)) + struct.pack("<%df" % len(val), *val)
        return head + struct.pack("<IQ", 5, len(val)) + struct.pack("<%di" % len(val), *val)
    raise TypeError(type(val))


def nbytes(t, ne):
    n = 1
    for d in ne[1:]:
        n *= d
    return n * (ROW_META.get(t, 0) + (ne[0] // BLCK[t]) * TYPE_SIZE[t])


def write_gguf(path, kv, tensors):
    """kv: dict key -> str | int | float.  tensors: list of (name, ggml_type, ne (ggml order, ne[0] = row length), data) with data = bytes-like /
    numpy array / zero-argument callable returning one (called at write time: big files are streamed)."""
    infos, off = [], 0
    for name, t, ne, _ in tensors:
        infos.append((name, t, ne, off))
        off = (off + nbytes(t, ne) + ALIGN - 1) // ALIGN * ALIGN
    with open(path, "wb") as f:
        f.write(struct.pack("<IIQQ", 0x46554747, 3, len(tensors), len(kv)))
        for k, v in kv.items():
            f.write(_kv(k, v))
        for name, t, ne, o in infos:
            f.write(_s(name) + struct.pack("<I", len(ne)) + b"".join(struct.pack("<Q", d) for d in ne) + struct.pack("<IQ", t, o))
        f.write(b"\0" * (-f.tell() % ALIGN))
        base = f.tell()
        for (name, t, ne, data), (_, _, _, o) in zip(tensors, infos):
            assert f.tell() == base + o, name
            d = data() if callable(data) else data
            b = np.ascontiguousarray(d).tobytes() if isinstance(d, np.ndarray) else bytes(d)
            assert len(b) == nbytes(t, ne), (name, len(b), nbytes(t, ne))
            f.write(b); f.write(b"\0" * (-len(b) % ALIGN))
    return path


def llama_kv(name, n_embd, n_ff, n_head, n_head_kv, n_layer, n_vocab, n_ctx=4096, n_expert=0, n_used=0, arch="llama", head_dim=None):
    """arch = "llama" | "qwen3" (src/llama-arch.cpp:29; qwen3 = llama + per-head q / k RMS norms and an explicit head size, src/llama-load-tensors.cpp:1459-1490)"""
    a = arch; hd = head_dim or n_embd // n_head
    kv = {"general.architecture": a, "general.name": name, a + ".context_length": n_ctx, a + ".embedding_length": n_embd,
          a + ".block_count": n_layer, a + ".feed_forward_length": n_ff, a + ".attention.head_count": n_head,
          a + ".attention.head_count_kv": n_head_kv, a + ".attention.layer_norm_rms_epsilon": 1e-5 if a == "llama" else 1e-6, a + ".rope.dimension_count": hd,
          a + ".rope.freq_base": 500000.0 if a == "llama" else 1000000.0, a + ".vocab_size": n_vocab, "tokenizer.ggml.model": "no_vocab"}
    if head_dim:
        kv[a + ".attention.key_length"] = hd; kv[a + ".attention.value_length"] = hd
    if n_expert:
        kv[a + ".expert_count"] = n_expert; kv[a + ".expert_used_count"] = n_used
    return kv


def use_more_bits(i, n):       # src/llama-quantize.cpp:312-314
    return i < n // 8 or i >= 7 * n // 8 or (i - n // 8) % 3 == 2


def q4_k_m(name, il, nl):
    if name == "output" or (name in ("attn_v", "ffn_down") and use_more_bits(il, nl)):
        return Q6_K
    return Q4_K


def iq4_nl_mix(name, il, nl):
    """LLAMA_FTYPE_MOSTLY_IQ4_NL: IQ4_NL everywhere (row lengths of Qwen3-0.6B are multiples of 32 but not all of 256), output / token_embd -> Q6_K
    (src/llama-quantize.cpp: output.weight of every ftype below Q8_0 is Q6_K unless overridden)"""
    return Q6_K if name in ("output", "token_embd") else IQ4_NL


def llama_tensors(n_embd, n_ff, n_head, n_head_kv, n_layer, n_vocab, make, types=q4_k_m, n_expert=0, embd_type=None, head_dim=None, qk_norm=False, tied=False):
    """tensor list in file order; `make(name, type, ne)` returns the data (or a callable producing it).
    head_dim: explicit head size (qwen3: n_head * head_dim != n_embd); qk_norm: per-head attn_q_norm / attn_k_norm (qwen3); tied: no output.weight"""
    hd = head_dim or n_embd // n_head; kvd = hd * n_head_kv; qd = hd * n_head
    out = []

    def add(name, t, ne):
        out.append((name, t, ne, make(name, t, ne)))
    add("token_embd.weight", embd_type if embd_type is not None else types("token_embd", 0, n_layer), [n_embd, n_vocab])
    for il in range(n_layer):
        p = "blk.%d." % il
        add(p + "attn_norm.weight", F32, [n_embd])
        add(p + "attn_q.weight", types("attn_q", il, n_layer), [n_embd, qd])
        add(p + "attn_k.weight", types("attn_k", il, n_layer), [n_embd, kvd])
        add(p + "attn_v.weight", types("attn_v", il, n_layer), [n_embd, kvd])
        add(p + "attn_output.weight", types("attn_output", il, n_layer), [qd, n_embd])
        if qk_norm:
            add(p + "attn_k_norm.weight", F32, [hd]); add(p + "attn_q_norm.weight", F32, [hd])
        add(p + "ffn_norm.weight", F32, [n_embd])
        if n_expert:
            add(p + "ffn_gate_inp.weight", F32, [n_embd, n_expert])
            add(p + "ffn_gate_exps.weight", types("ffn_gate", il, n_layer), [n_embd, n_ff, n_expert])
            add(p + "ffn_down_exps.weight", types("ffn_down", il, n_layer), [n_ff, n_embd, n_expert])
            add(p + "ffn_up_exps.weight", types("ffn_up", il, n_layer), [n_embd, n_ff, n_expert])
        else:
            add(p + "ffn_gate.weight", types("ffn_gate", il, n_layer), [n_embd, n_ff])
            add(p + "ffn_down.weight", types("ffn_down", il, n_layer), [n_ff, n_embd])
            add(p + "ffn_up.weight", types("ffn_up", il, n_layer), [n_embd, n_ff])
    add("output_norm.weight", F32, [n_embd])
    if not tied:
        add("output.weight", types("output", 0, n_layer), [n_embd, n_vocab])
    return out


def random_blocks(t, ne, rng, d_scale=0.004):
    """random-bit quant blocks with finite, small fp16 super-block scales (every byte pattern is a valid block)"""
    rows = int(np.prod(ne[1:])); nb = ne[0] // BLCK[t]; ts = TYPE_SIZE[t]
    w = rng.integers(0, 256, size=(rows, nb, ts), dtype=np.uint8)
    for off in D_OFFS[t]:
        d = (rng.random((rows, nb), dtype=np.float32) * d_scale + d_scale * 0.1).astype(np.float16)
        if off != 2:
            d = d * (rng.integers(0, 2, (rows, nb)) * 2 - 1).astype(np.float16)
        w[:, :, off:off + 2] = d.view(np.uint8).reshape(rows, nb, 2)
    return w.reshape(-1)


def spm_vocab(n_vocab):
    """a minimal SentencePiece-style vocabulary (tokenizer.ggml.model = "llama", src/llama-vocab.cpp): <unk> <s> </s>, the 256 byte tokens, then normal pieces -- enough for
    llama-server to tokenize / detokenize (the throughput tools run on `no_vocab` files and feed token ids)"""
    assert n_vocab >= 3 + 256 + 1
    toks = ["<unk>", "<s>", "</s>"] + ["<0x%02X>" % b for b in range(256)] + ["\u2581t%d" % i for i in range(n_vocab - 259)]
    types = [2, 3, 3] + [6] * 256 + [1] * (n_vocab - 259)
    return {"tokenizer.ggml.model": "llama", "tokenizer.ggml.tokens": toks, "tokenizer.ggml.scores": [0.0] * 259 + [-float(i) for i in range(n_vocab - 259)],
            "tokenizer.ggml.token_type": types, "tokenizer.ggml.bos_token_id": 1, "tokenizer.ggml.eos_token_id": 2, "tokenizer.ggml.unknown_token_id": 0}


def tiny_model(path, ref, n_embd=512, n_ff=1024, n_head=4, n_head_kv=2, n_layer=2, n_vocab=512, types=q4_k_m, n_expert=0, n_used=0, seed=0, vocab=False,
               arch="llama", head_dim=None, qk_norm=False, tied=False):
    """small model with REAL quantizer output (ref = oracle.bindings.Ref): weights N(0, (1/sqrt(fan_in))^2), norms ~1; vocab: carry spm_vocab() instead of `no_vocab`;
    arch / head_dim / qk_norm / tied: as bench_model (a qwen3 model: per-head q / k norms, explicit head size, NEOX rotation, tied embeddings)"""
    rng = np.random.default_rng(seed)

    def make(name, t, ne):
        if t == F32:
            if name.endswith("norm.weight"):
                return (1.0 + 0.1 * rng.standard_normal(ne[0])).astype(np.float32)
            return (rng.standard_normal(int(np.prod(ne))) / np.sqrt(ne[0])).astype(np.float32)
        rows = int(np.prod(ne[1:]))
        w = (rng.standard_normal((rows, ne[0])) / np.sqrt(ne[0])).astype(np.float32)
        return ref.quantize(t, w)
    kv = llama_kv("tiny-synth", n_embd, n_ff, n_head, n_head_kv, n_layer, n_vocab, n_ctx=512, n_expert=n_expert, n_used=n_used, arch=arch, head_dim=head_dim)
    if vocab:
        kv.update(spm_vocab(n_vocab))
    return write_gguf(path, kv, llama_tensors(n_embd, n_ff, n_head, n_head_kv, n_layer, n_vocab, make, types=types, n_expert=n_expert, head_dim=head_dim, qk_norm=qk_norm, tied=tied))


def bench_model(path, n_embd=4096, n_ff=14336, n_head=32, n_head_kv=8, n_layer=32, n_vocab=128256, types=q4_k_m, n_expert=0, n_used=0, seed=1, name="Llama-3-8B-synth",
                arch="llama", head_dim=None, qk_norm=False, tied=False):
    """full-size shapes, random-bit blocks (throughput runs): streamed, one tensor in memory at a time"""
    rng = np.random.default_rng(seed)

    def make(name, t, ne):
        if t == F32:
            if name.endswith("norm.weight"):
                return lambda: np.ones(ne[0], np.float32)
            return lambda: (rng.standard_normal(int(np.prod(ne))) / np.sqrt(ne[0])).astype(np.float32)
        return lambda: random_blocks(t, ne, rng)
    kv = llama_kv(name, n_embd, n_ff, n_head, n_head_kv, n_layer, n_vocab, n_ctx=8192, n_expert=n_expert, n_used=n_used, arch=arch, head_dim=head_dim)
    return write_gguf(path, kv, llama_tensors(n_embd, n_ff, n_head, n_head_kv, n_layer, n_vocab, make, types=types, n_expert=n_expert, head_dim=head_dim,
                                              qk_norm=qk_norm, tied=tied))


def qwen3_06b_model(path, seed=2):
    """BASELINE.json configs[0]: Qwen3-0.6B shapes (n_embd 1024, n_ff 3072, 16 heads / 8 KV heads x 128, 28 layers, vocab 151936, tied embeddings) in the
    IQ4_NL mix, random-bit blocks -- for `llama-bench -p 128 -n 32` (CPU reference and through the shim)"""
    return bench_model(path, n_embd=1024, n_ff=3072, n_head=16, n_head_kv=8, n_layer=28, n_vocab=151936, types=iq4_nl_mix, seed=seed, name="Qwen3-0.6B-synth",
                       arch="qwen3", head_dim=128, qk_norm=True, tied=True)


if __name__ == "__main__":
    import sys
    bench_model(sys.argv[1], n_layer=int(sys.argv[2]) if len(sys.argv) > 2 else 32)
    print(sys.argv[1])
