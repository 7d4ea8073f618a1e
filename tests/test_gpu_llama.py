"""-m gpu: the drop-in boundary end to end with the reference's OWN consumers: the unmodified libllama / llama-bench of /root/reference,
built by ik_llama.cpp_amd/backend/Makefile.llama with -DGGML_USE_CUDA and linked against the shim (libggml-cuda-cdna4.so) instead of
ggml-cuda, load synthetic GGUF files (tests/gguf_synth.py) and run with every layer offloaded (-ngl 99).  Ops outside the hot path run on
the reference CPU backend (the scheduler splits the graph, ggml-backend.cpp:1314-1360).

Parity bar: logits of the offloaded run vs the pure-CPU run (-ngl 0) of the same binary, NMSE <= 5e-4 (the reference's own MUL_MAT
tolerance, tests/test-backend-ops.cpp:979-981; the CPU path itself is lossy at N >= 32)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from common import NMSE_VS_CPU, nmse
from oracle import bindings as ob

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "oracle", "_ref", "llama", "bin")
BENCH, LOGITS = os.path.join(BIN, "llama-bench"), os.path.join(BIN, "llama_logits")
N_VOCAB = 512


def run(cmd, env=None, timeout=180):
    e = dict(os.environ); e.update(env or {})
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=e, timeout=timeout)
    assert r.returncode == 0, "%s\nrc=%d\n%s\n%s" % (" ".join(cmd), r.returncode, r.stdout.decode(errors="replace")[-2000:], r.stderr.decode(errors="replace")[-3000:])
    return r.stdout.decode(errors="replace"), r.stderr.decode(errors="replace")


@pytest.fixture(scope="module")
def models(tmp_path_factory):
    if not (os.path.exists(BENCH) and os.path.exists(LOGITS)):
        pytest.skip("oracle/_ref/llama not built (needs /root/reference: make -C ik_llama.cpp_amd/backend -f Makefile.llama)")
    if ob.ref_path() is None:
        pytest.skip("needs the reference quantizer (oracle/_ref) to synthesize well-conditioned models")
    import gguf_synth as gs
    ref = ob.Ref(); d = tmp_path_factory.mktemp("gguf")

    def iq_mix(name, il, nl):       # the sub-4-bit / non-linear types of the hot path in one model
        return {"attn_q": gs.IQ2_S, "attn_k": gs.IQ4_NL, "attn_v": gs.Q6_K, "attn_output": gs.IQ3_S, "ffn_gate": gs.IQ2_S, "ffn_up": gs.IQ2_S,
                "ffn_down": gs.Q5_K, "output": gs.Q6_K, "token_embd": gs.Q4_K}[name]
    def r4_mix(name, il, nl):       # row-interleaved tensors as an offline-repacked GGUF carries them; token_embd stays plain (GET_ROWS)
        return {"token_embd": gs.Q4_K, "output": gs.Q6_K}.get(name, gs.Q4_K + 200 if name != "attn_v" else gs.Q6_K + 200)
    gs.add_types(ob)
    def iqk_mix(name, il, nl):      # ik's non-linear and row-scaled types in one model
        return {"attn_q": ob.IQ4_K, "attn_k": ob.IQ4_KS, "attn_v": ob.IQ5_K, "attn_output": ob.IQ3_K, "ffn_gate": ob.IQ2_K, "ffn_up": ob.IQ2_K,
                "ffn_down": ob.IQ5_KS if il == 0 else ob.IQ4_KSS, "output": ob.IQ6_K, "token_embd": ob.IQ4_XS}[name]
    def legacy_mix(name, il, nl):   # legacy 32-blocks, the small K-quants, the remaining IQ types
        return {"attn_q": ob.Q5_0, "attn_k": ob.Q4_1, "attn_v": ob.Q8_0, "attn_output": ob.Q6_0, "ffn_gate": ob.Q3_K if il == 0 else ob.IQ2_XS, "ffn_up": ob.Q3_K if il == 0 else ob.IQ2_XS,
                "ffn_down": ob.Q2_K if il == 0 else ob.IQ3_XXS, "output": ob.Q5_1, "token_embd": ob.Q4_0}[name]
    def onebit_mix(name, il, nl):   # the ternary-codebook types (decode units + the f16 prompt route) and MXFP4
        return {"attn_q": ob.IQ1_M, "attn_k": ob.MXFP4, "attn_v": ob.Q8_0 if il == 0 else ob.MXFP4, "attn_output": ob.IQ1_S, "ffn_gate": ob.IQ1_S, "ffn_up": ob.IQ1_S,
                "ffn_down": ob.IQ1_M, "output": ob.Q6_K, "token_embd": ob.IQ1_S if il == 0 else ob.Q4_0}[name]
    def kt_mix(name, il, nl):       # the trellis types (decode units + the f16 prompt route); token_embd through GET_ROWS
        return {"attn_q": ob.IQ2_KT, "attn_k": ob.IQ4_KT, "attn_v": ob.IQ4_KT, "attn_output": ob.IQ3_KT, "ffn_gate": ob.IQ2_KT if il == 0 else ob.IQ1_KT, "ffn_up": ob.IQ2_KT if il == 0 else ob.IQ1_KT,
                "ffn_down": ob.IQ3_KT, "output": ob.Q6_K, "token_embd": ob.IQ4_KT if il == 0 else ob.Q4_K}[name]

    class KtRef:        # the reference's trellis search takes minutes for a model's worth of rows: random generator seeds / block scales instead (every bit pattern is a valid
        orc = ob.Oracle()      # block), with the ROW scale chosen so that the de-quantized row has the standard deviation the synthetic model asks for
        def quantize(self, t, w):
            if t not in ob.KT_TYPES:
                return ref.quantize(t, w)
            from common import random_block_bytes
            q = random_block_bytes(t, w.shape[0], w.shape[1], 77 + t); q[:, :4] = np.ones((w.shape[0], 1), np.float32).view(np.uint8)
            sd = self.orc.dequantize(t, q, w.shape[1]).std(axis=1, keepdims=True)
            q[:, :4] = (w.std(axis=1, keepdims=True) / sd).astype(np.float32).view(np.uint8)
            return q
    m = {"dense": gs.tiny_model(str(d / "dense.gguf"), ref, n_vocab=N_VOCAB),
         "kt": gs.tiny_model(str(d / "kt.gguf"), KtRef(), n_vocab=N_VOCAB, types=kt_mix, seed=7),
         "iqk": gs.tiny_model(str(d / "iqk.gguf"), ref, n_vocab=N_VOCAB, types=iqk_mix, seed=4),
         "legacy": gs.tiny_model(str(d / "legacy.gguf"), ref, n_vocab=N_VOCAB, types=legacy_mix, seed=5),
         "onebit": gs.tiny_model(str(d / "onebit.gguf"), ref, n_vocab=N_VOCAB, types=onebit_mix, seed=6),
         "iq": gs.tiny_model(str(d / "iq.gguf"), ref, n_vocab=N_VOCAB, types=iq_mix, seed=1),
         "moe": gs.tiny_model(str(d / "moe.gguf"), ref, n_vocab=N_VOCAB, n_expert=4, n_used=2, seed=2),
         # rows of 4096 weights, 32 q heads / 8 KV heads of 128: the shapes at which the decode launches of an 8B model take their fused forms (q,k,v epilogue, attention +
         # attn_output in one launch, 64 lanes per row)
         "wide": gs.tiny_model(str(d / "wide.gguf"), ref, n_embd=4096, n_ff=1024, n_head=32, n_head_kv=8, n_layer=2, n_vocab=N_VOCAB, seed=9),
         # a MoE model whose rows are long enough (1024) for the ffn_norm to ride in the router's launch (cdna4_op_moe_router_norm)
         "moe1k": gs.tiny_model(str(d / "moe1k.gguf"), ref, n_embd=1024, n_ff=512, n_head=8, n_head_kv=4, n_layer=2, n_vocab=N_VOCAB, n_expert=4, n_used=2, seed=12),
         # a Qwen3-style model (llm_build_mul_mat_qkv with attn_q_norm / attn_k_norm, NEOX rotation, explicit head size 128, tied embeddings): the q / k norm + ROPE + KV-store
         # launch of round 5 inside a real graph, prompt and decode
         "qwen3": gs.tiny_model(str(d / "qwen3.gguf"), ref, n_embd=1024, n_ff=1536, n_head=8, n_head_kv=4, n_layer=3, n_vocab=N_VOCAB, seed=10, arch="qwen3", head_dim=128, qk_norm=True,
                                tied=True, types=lambda name, il, nl: gs.Q6_K if name == "token_embd" else gs.q4_k_m(name, il, nl))}
    gs.TYPE_SIZE.update({gs.Q4_K + 200: 144, gs.Q6_K + 200: 210}); gs.BLCK.update({gs.Q4_K + 200: 256, gs.Q6_K + 200: 256})
    orc = ob.Oracle()

    class R4Ref:        # quantize with the reference, then interleave 4 rows like `llama-quantize --repack` (oracle restatement, pinned vs iqk_repack_tensor)
        def quantize(self, t, w):
            return orc.repack_r4(t - 200, ref.quantize(t - 200, w), w.shape[1]) if t >= 200 else ref.quantize(t, w)
    m["r4"] = gs.tiny_model(str(d / "r4.gguf"), R4Ref(), n_vocab=N_VOCAB, types=r4_mix, seed=3)
    return m


_CPU_LOGITS = {}      # the reference CPU backend's logits of (model, prompt length, decode steps): the same for every test that compares against them


def logits(model, ngl, n_tokens, n_decode, sm="none", env=None, tmp="/tmp", kv_offload=True):
    if ngl == 0 and not env and sm == "none":
        key = (model, n_tokens, n_decode)
        if key not in _CPU_LOGITS:
            _CPU_LOGITS[key] = _logits(model, ngl, n_tokens, n_decode, sm, env, tmp, kv_offload)
        return _CPU_LOGITS[key].copy()
    return _logits(model, ngl, n_tokens, n_decode, sm, env, tmp, kv_offload)


def _logits(model, ngl, n_tokens, n_decode, sm, env, tmp, kv_offload):
    out = os.path.join(tmp, "logits_%d_%s_%d.bin" % (ngl, sm, os.getpid()))
    env = dict(env or {})
    if kv_offload and ngl > 0:
        env["LLAMA_LOGITS_KV_OFFLOAD"] = "1"            # KV cache in device memory: KV writes + attention run on the device
    run([LOGITS, model, str(ngl), str(n_tokens), "8", sm, out, str(n_decode)], env=env)
    a = np.fromfile(out, np.float32).reshape(1 + n_decode, N_VOCAB); os.remove(out)
    assert np.all(np.isfinite(a))
    return a


def test_llama_bench_links_and_lists_the_device():
    """the unmodified llama-bench resolved every ggml_backend_cuda_* symbol from the shim and runs"""
    if not os.path.exists(BENCH):
        pytest.skip("oracle/_ref/llama not built")
    out, _ = run([BENCH, "--help"])
    assert "usage" in out.lower()
    ldd, _ = run(["ldd", BENCH])
    assert "libggml-cuda-cdna4.so" in ldd and "libggml-hip-cdna4.so" in ldd and "not found" not in ldd


@pytest.mark.parametrize("name", ["dense", "iq", "moe", "r4"])
def test_llama_bench_runs_offloaded(name, models):
    # KV cache in HBM, flash attention on: the whole graph (norms, rope, KV writes, attention, router) runs on the device
    out, err = run([BENCH, "-m", models[name], "-p", "64", "-n", "8", "-ngl", "99", "-fa", "1", "-t", "8", "-r", "2", "-o", "json"])
    res = json.loads(out[out.index("["):])
    assert len(res) == 2 and all(r["avg_ts"] > 0 for r in res)
    assert all(r["n_gpu_layers"] == 99 for r in res) and "gfx950" in json.dumps(res)          # device description comes from the shim


@pytest.mark.parametrize("name", ["iqk", "legacy", "onebit", "kt"])
def test_logits_more_weight_types_vs_cpu(name, models, tmp_path):
    """the SURVEY 8 f3 types end to end through libllama: a prompt batch (MFMA tiles) + decode steps (GEMV units) of models mixing them, -ngl 99 vs the reference CPU backend.
    (iqk, decode rows: the CPU's AVX-512 kernel for N < 32 saturates int16 pair sums for IQ4_K / IQ5_K / IQ4_KS / IQ5_KS / IQ4_KSS / IQ6_K on full-range int8 activations and is
    itself 6e-4 ... 5e-3 NMSE per mat-mul from its exact form, tests/test_oracle_vs_ref.py; the device computes the exact sums -- measured 4e-3 on the logits -- so only a sanity bar
    holds there; the prompt row, where the CPU repacks to Q8_K_R8 instead, keeps the reference's backend-op tolerance)"""
    gpu = logits(models[name], 99, 48, 3, tmp=str(tmp_path)); cpu = logits(models[name], 0, 48, 3, tmp=str(tmp_path))
    bars = [NMSE_VS_CPU if name == "legacy" else (4 * NMSE_VS_CPU if i == 0 else 2e-2) for i in range(gpu.shape[0])]
    bad = [(i, nmse(gpu[i], cpu[i])) for i in range(gpu.shape[0]) if not nmse(gpu[i], cpu[i]) < bars[i]]
    # (round 3 saw iqk / decode row 1 at NMSE 0.41 once and retried here; round 4's soak reproduced and fixed it -- the fused ROPE + ROPE + KV-store launch wrote the rotated K
    #  over the un-rotated Q, where the graph allocator had placed it, while other workgroups were still reading Q: test_soak_repetitions_are_bit_identical below)
    assert not bad, dict(model=name, rows=bad)


SOAK = os.path.join(BIN, "llama_soak")


@pytest.mark.parametrize("mode,sm", [("fresh", "none"), ("reuse", "none"), ("reuse", "graph")])
@pytest.mark.parametrize("name", ["iqk", "dense", "wide", "qwen3", "moe1k"])
def test_soak_repetitions_are_bit_identical(name, mode, sm, models, tmp_path):
    """200 repetitions of (48-token prompt + 3 decode steps) in ONE process: every logits row must hash like the first repetition's.  `fresh` = a new context (backend) per
    repetition (eager walk, capture, replay); `reuse` = one context, KV cache cleared (every graph, the prompt's included, replayed from its HIP graph); `graph` = two logical
    devices with -sm graph (the decode steps take the ROPE + KV-store launch there).  The launch-order race this guards against failed 2 ... 20 % of the repetitions
    (scripts/soak_logits.py, profiles/r04_soak*.json); the overlap assertions of the shim are on, so an unsafe operand layout aborts instead of racing."""
    if not os.path.exists(SOAK):
        pytest.skip("oracle/_ref/llama/bin/llama_soak not built")
    if sm == "graph" and name == "moe1k":
        pytest.skip("a LLAMA-arch MoE model is not a -sm graph case of the reference")
    env = {"LLAMA_LOGITS_KV_OFFLOAD": "1", "CDNA4_DETERMINISTIC": "1", "GGML_CDNA4_CHECK_OVERLAP": "1"}
    if sm == "graph":
        env["GGML_CDNA4_FAKE_DEVICES"] = "2"
    if mode == "reuse":
        env["GGML_CDNA4_GRAPH_MAX_BATCH"] = "1000000"       # (prompt-size graphs run eagerly by default: keep the captured-prompt path under the soak)
    out, _ = run([SOAK, models[name], "99", "48", "3", "200", "8", sm, mode], env=env, timeout=300)
    rec = json.loads([ln for ln in out.splitlines() if ln.startswith("{")][-1])
    assert rec["mismatched_rows"] == 0 and rec["nonfinite_rows"] == 0, rec


@pytest.mark.parametrize("kv_offload", [True, False], ids=["kv_hbm", "kv_host"])
@pytest.mark.parametrize("name", ["dense", "iq", "moe", "wide", "qwen3", "moe1k"])
def test_logits_offloaded_vs_cpu(name, kv_offload, models, tmp_path):
    """prompt batch of 48 tokens (prefill kernels) + 3 decode steps (GEMV kernels): -ngl 99 through the shim vs -ngl 0 on the reference CPU backend.
    kv_host: the KV cache stays in host memory, so the scheduler splits every layer at the attention (ggml-backend.cpp:1314-1360)."""
    gpu = logits(models[name], 99, 48, 3, tmp=str(tmp_path), kv_offload=kv_offload); cpu = logits(models[name], 0, 48, 3, tmp=str(tmp_path))
    for i in range(gpu.shape[0]):
        assert nmse(gpu[i], cpu[i]) < NMSE_VS_CPU, (name, i, nmse(gpu[i], cpu[i]))


def test_moe_graph_folds_ffn_norm_into_the_router_launch(models, tmp_path):
    """one decoded token of a MoE layer: FUSED_RMS_NORM(ffn_norm) + the six router nodes are ONE launch (the normed row is still written: the experts read it); 2 layers x 3 decode
    steps; with the fold off (GGML_CDNA4_FUSION_OFF bit 1024) no logit bit changes"""
    import re
    out = os.path.join(str(tmp_path), "m.bin"); rows = []
    for extra in ({}, {"GGML_CDNA4_FUSION_OFF": "1024"}):
        env = {"LLAMA_LOGITS_KV_OFFLOAD": "1", "GGML_CDNA4_STATS": "1", "GGML_CDNA4_PARAMS": "graphs=0"}; env.update(extra)
        _, err = run([LOGITS, models["moe1k"], "99", "48", "8", "none", out, "3"], env=env)
        m = re.search(r"RMS_NORM in MoE router (\d+)", err)
        assert m, err[-1500:]
        rows.append((int(m.group(1)), np.fromfile(out, np.float32).reshape(4, N_VOCAB)))
    assert rows[0][0] == 2 * 3 and rows[1][0] == 0, (rows[0][0], rows[1][0])
    np.testing.assert_array_equal(rows[0][1].view(np.uint32), rows[1][1].view(np.uint32))


def test_qwen3_graph_takes_the_q_k_norm_rope_kv_store_launch(models, tmp_path):
    """the six nodes between the q,k,v mat-muls and the attention of a Qwen3-style layer (FUSED_RMS_NORM(q) ROPE(q) FUSED_RMS_NORM(k) ROPE(k) CPY(k) CPY(v), in the order llm_build_kv
    expands them) are ONE launch in the prompt graph and in every decode graph (3 layers x (1 prompt + 3 decode steps)), and switching the launch off (GGML_CDNA4_FUSION_OFF bit 512)
    changes no logit bit (tests/test_gpu_qk_norm_rope.py: bit-identical to the six launches)"""
    import re
    out = os.path.join(str(tmp_path), "q.bin"); rows = []
    for extra in ({}, {"GGML_CDNA4_FUSION_OFF": "512"}):
        env = {"LLAMA_LOGITS_KV_OFFLOAD": "1", "GGML_CDNA4_STATS": "1", "GGML_CDNA4_PARAMS": "graphs=0"}; env.update(extra)
        _, err = run([LOGITS, models["qwen3"], "99", "48", "8", "none", out, "3"], env=env)
        m = re.search(r"q/k norms\+ROPE\+KV stores (\d+)", err)
        assert m, err[-1500:]
        rows.append((int(m.group(1)), np.fromfile(out, np.float32).reshape(4, N_VOCAB)))
    assert rows[0][0] == 3 * 4 and rows[1][0] == 0, (rows[0][0], rows[1][0])
    np.testing.assert_array_equal(rows[0][1].view(np.uint32), rows[1][1].view(np.uint32))


def test_logits_r4_model(models, tmp_path):
    """a GGUF of row-interleaved (_R4) tensors: re-tiled at upload, computed with the _R4 kernels' activation arithmetic.
    Token by token (1-token prompt + 3 decode steps: the GEMV kernels, same int8 activations and block sums as the CPU _R4 kernels) the offloaded
    run reproduces the CPU logits to rounding.  The _R4 CPU kernels quantize activations with ONE scale per 256 values (Q8_K32 / Q8_K,
    ggml.c:989,1019,1050), so the CPU logits themselves sit ~1e-3 (NMSE, this 2-layer model) from exact arithmetic and any difference in the
    prompt batch (f16 MFMA path, or the CPU's N >= 32 re-quantization of weights to Q8) moves the roundings of every later
    activation: with a 48-token prompt both prompt modes are held to 4x the reference's MUL_MAT bar (measured 3e-4 ... 1e-3)."""
    cpu1 = logits(models["r4"], 0, 1, 3, tmp=str(tmp_path)); gpu1 = logits(models["r4"], 99, 1, 3, tmp=str(tmp_path))
    for i in range(cpu1.shape[0]):
        assert nmse(gpu1[i], cpu1[i]) < 1e-8, ("token by token", i, nmse(gpu1[i], cpu1[i]))
    cpu = logits(models["r4"], 0, 48, 3, tmp=str(tmp_path))
    par = logits(models["r4"], 99, 48, 3, env={"GGML_CDNA4_PREFILL_INT8": "1"}, tmp=str(tmp_path))
    mfma = logits(models["r4"], 99, 48, 3, tmp=str(tmp_path))
    for i in range(cpu.shape[0]):
        assert nmse(par[i], cpu[i]) < 4 * NMSE_VS_CPU, ("int8 prompt mode", i, nmse(par[i], cpu[i]))
        assert nmse(mfma[i], cpu[i]) < 4 * NMSE_VS_CPU, ("mfma", i, nmse(mfma[i], cpu[i]))


@pytest.mark.parametrize("name", ["dense", "iq"])
def test_split_mode_graph_two_logical_devices(name, models, tmp_path):
    """-sm graph: libllama places the weights in the split buffer type (per-device slices uploaded by the shim's set_tensor), builds per-device
    sub-graphs on the splits and GGML_OP_REDUCE nodes that the shim executes across its backends.  One MI355X per box here, so the two logical
    devices (GGML_CDNA4_FAKE_DEVICES=2) share the physical GPU: slicing, scheduling and the reduce are the multi-device code paths.
    Prompt batches > 32 tokens carry their partial sums as f16 (cparams.reduce_type, llama-build-context.cpp:1198): ADD / REDUCE / RMS_NORM on f16.
    (A LLAMA-arch MoE model is not a -sm graph case of the reference: build_llama.cpp:165-180 hands the un-split parents to llm_build_moe_ffn.)"""
    env = {"GGML_CDNA4_FAKE_DEVICES": "2"}
    gpu = logits(models[name], 99, 48, 3, sm="graph", env=env, tmp=str(tmp_path)); cpu = logits(models[name], 0, 48, 3, tmp=str(tmp_path))
    for i in range(gpu.shape[0]):
        assert nmse(gpu[i], cpu[i]) < NMSE_VS_CPU, (name, i, nmse(gpu[i], cpu[i]))


@pytest.mark.parametrize("name,sm,ndev", [("dense", "none", 1), ("iq", "none", 1), ("moe", "none", 1), ("qwen3", "none", 1), ("dense", "graph", 2), ("iqk", "graph", 2), ("dense", "layer", 2)])
def test_no_read_of_memory_nobody_wrote(name, sm, ndev, models, tmp_path):
    """A fresh process gets zero pages from the driver, which hides a kernel that reads device memory nobody wrote (a probability-0 key times a zero V is 0; times whatever a
    recycled allocation holds it may be a NaN).  GGML_CDNA4_POISON_MB fills 4 GB of device memory with NaN bits and releases them before the backend allocates anything, so that
    compute buffers, KV caches and workspaces of THIS process start as NaNs, and GGML_CDNA4_CHECK_NAN walks the graphs eagerly and aborts at the first node with a non-finite result.
    Round 6: found with -sm graph -- its per-device KV caches live in split allocations that nobody cleared (the reference's split_buffer_clear is a no-op too), and the
    prompt attention multiplied masked keys' zeros with the garbage behind the written rows."""
    env = {"GGML_CDNA4_POISON_MB": "4096", "GGML_CDNA4_CHECK_NAN": "1", "GGML_CDNA4_PARAMS": "graphs=0"}
    if ndev > 1:
        env["GGML_CDNA4_FAKE_DEVICES"] = str(ndev)
    gpu = logits(models[name], 99, 48, 3, sm=sm, env=env, tmp=str(tmp_path)); cpu = logits(models[name], 0, 48, 3, tmp=str(tmp_path))
    for i in range(gpu.shape[0]):
        bar = (4 * NMSE_VS_CPU if i == 0 else 2e-2) if name == "iqk" else (4 * NMSE_VS_CPU if name in ("iq", "moe") else NMSE_VS_CPU)      # (iqk: the bars of test_logits_more_type_families)
        assert nmse(gpu[i], cpu[i]) < bar, (name, sm, i, nmse(gpu[i], cpu[i]))
    # ... and with the HIP graphs on (what a user runs): same poison, finite logits
    env.pop("GGML_CDNA4_CHECK_NAN"); env.pop("GGML_CDNA4_PARAMS")
    g2 = logits(models[name], 99, 48, 3, sm=sm, env=env, tmp=str(tmp_path))
    assert np.all(np.isfinite(g2))


@pytest.mark.parametrize("name", ["dense", "qwen3"])
def test_quantized_kv_cache_stays_on_the_device(name, models, tmp_path):
    """-ctk q8_0 -ctv q8_0 (round 6; VERDICT r05 "missing" 4): the KV-cache writes (CPY f32 -> Q8_0) and FLASH_ATTN_EXT on Q8_0 K / V views run on the device -- the shim must not
    print its "attention falls to the CPU backend" warning, the graph must have no CPU split inside the layers -- and the logits match the reference CPU backend running the SAME
    quantized cache (its iqk flash attention on Q8_0 K / V), prompt and decode steps."""
    env = {"LLAMA_LOGITS_CACHE_TYPE": "q8_0"}
    out = os.path.join(str(tmp_path), "q8kv.bin")
    o, err = run([LOGITS, models[name], "99", "48", "8", "none", out, "3"], env=dict(env, LLAMA_LOGITS_KV_OFFLOAD="1"))
    assert "falls to the CPU backend" not in err, err[-600:]
    gpu = np.fromfile(out, np.float32).reshape(4, N_VOCAB); os.remove(out)
    cpu = _logits(models[name], 0, 48, 3, "none", env, str(tmp_path), False)
    assert np.all(np.isfinite(gpu))
    for i in range(4):
        assert nmse(gpu[i], cpu[i]) < 4 * NMSE_VS_CPU, (name, i, nmse(gpu[i], cpu[i]))
    f16 = logits(models[name], 99, 48, 3, tmp=str(tmp_path))          # and the quantized cache is a small perturbation of the f16 one, not something else
    for i in range(4):
        assert nmse(gpu[i], f16[i]) < 2e-2, (name, i, nmse(gpu[i], f16[i]))


def test_split_mode_graph_row_scaled_types(models, tmp_path):
    """-sm graph with K-split tensors of the ROW-SCALED types (ffn_down: IQ5_KS / IQ4_KSS, attn_k: IQ4_KS row-split): every split must carry a copy of the row's
    meta bytes in front of its block range (ggml-cuda.cu:1073-1086).  Same device kernels on both sides (one device vs two logical devices), so the logits
    differ only by the summation order across the K halves and the f16 partial sums of the prompt batch."""
    env = {"GGML_CDNA4_FAKE_DEVICES": "2"}
    two = logits(models["iqk"], 99, 48, 3, sm="graph", env=env, tmp=str(tmp_path)); one = logits(models["iqk"], 99, 48, 3, tmp=str(tmp_path))
    for i in range(two.shape[0]):
        assert nmse(two[i], one[i]) < NMSE_VS_CPU, (i, nmse(two[i], one[i]))


@pytest.mark.parametrize("name", ["dense", "moe"])
def test_split_mode_layer_two_logical_devices(name, models, tmp_path):
    """-sm layer: whole layers alternate between the two backends; the scheduler copies the residual stream across (cpy_tensor_async / events)"""
    env = {"GGML_CDNA4_FAKE_DEVICES": "2"}
    gpu = logits(models[name], 99, 48, 3, sm="layer", env=env, tmp=str(tmp_path)); cpu = logits(models[name], 0, 48, 3, tmp=str(tmp_path))
    for i in range(gpu.shape[0]):
        assert nmse(gpu[i], cpu[i]) < NMSE_VS_CPU, (name, i, nmse(gpu[i], cpu[i]))


@pytest.mark.parametrize("name", ["dense", "moe"])
def test_decode_steps_replayed_from_a_hip_graph(name, models, tmp_path):
    """8 decode steps: from the third one on the token graph is a replayed HIP graph whose KV-cache write positions come from the slot table
    (the reference captures CUDA graphs the same way, ggml-cuda.cu:4408-4760).  Same kernels either way: identical logits with graphs off."""
    on = logits(models[name], 99, 5, 8, tmp=str(tmp_path)); off = logits(models[name], 99, 5, 8, env={"GGML_CDNA4_PARAMS": "graphs=0"}, tmp=str(tmp_path))
    cpu = logits(models[name], 0, 5, 8, tmp=str(tmp_path))
    np.testing.assert_array_equal(on, off)
    for i in range(on.shape[0]):
        assert nmse(on[i], cpu[i]) < NMSE_VS_CPU, (name, i, nmse(on[i], cpu[i]))


def _free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0)); return sk.getsockname()[1]


def _serve_and_complete(server, model, ngl, tmp_path, tag, env=None):
    """start llama-server, wait for /health, POST one greedy /completion, return (body, server log)"""
    import time, urllib.request
    port = _free_port(); logp = str(tmp_path / ("server_%s.log" % tag)); log = open(logp, "wb")
    e = dict(os.environ); e.update(env or {})
    p = subprocess.Popen([server, "-m", model, "-ngl", str(ngl), "-fa", "1", "--host", "127.0.0.1", "--port", str(port), "-c", "512", "-t", "4", "--no-warmup"],
                         stdout=log, stderr=subprocess.STDOUT, env=e)
    try:
        ok = False
        for _ in range(120):
            if p.poll() is not None:
                break
            try:
                with urllib.request.urlopen("http://127.0.0.1:%d/health" % port, timeout=2) as r:
                    if r.status == 200:
                        ok = True; break
            except Exception:      # noqa: BLE001 -- not listening yet / still loading (503)
                time.sleep(0.5)
        assert ok, open(logp, "rb").read()[-3000:].decode(errors="replace")
        # greedy, special and byte tokens masked (a random model would emit invalid UTF-8 through them and the JSON reply would fail)
        body = {"prompt": "t1 t2 t3 t4 t5 t6 t7 t8 t9 t10 t11 t12", "n_predict": 8, "temperature": 0.0, "cache_prompt": False, "ignore_eos": True, "logit_bias": [[i, False] for i in range(0, 259)]}
        req = urllib.request.Request("http://127.0.0.1:%d/completion" % port, data=json.dumps(body).encode(), headers={"Content-Type": "application/json"})
        with urllib.request.urlopen(req, timeout=60) as r:
            out = json.loads(r.read().decode())
    finally:
        p.terminate()
        try:
            p.wait(timeout=20)
        except subprocess.TimeoutExpired:
            p.kill()
        log.close()
    return out, open(logp, "rb").read().decode(errors="replace")


def test_llama_server_serves_a_completion_through_the_shim(tmp_path):
    """north_star: "llama-bench and llama-server load unmodified GGUF files".  The reference's llama-server (unmodified sources, Makefile.llama) loads a synthetic GGUF that
    carries a small SentencePiece vocabulary with every layer on the device, answers /health, tokenizes a prompt and generates greedily; the continuation equals the one the
    same binary produces on the reference CPU backend (GPU hidden)."""
    server = os.path.join(BIN, "llama-server")
    if not os.path.exists(server) or ob.ref_path() is None:
        pytest.skip("llama-server / reference quantizer not built")
    import gguf_synth as gs
    model = gs.tiny_model(str(tmp_path / "dense_vocab.gguf"), ob.Ref(), n_vocab=N_VOCAB, vocab=True, seed=11)
    gpu, log = _serve_and_complete(server, model, 99, tmp_path, "gpu")
    assert gpu.get("tokens_predicted") == 8 and gpu.get("tokens_evaluated", 0) >= 12, gpu
    assert "gfx950" in log or "CUDA0" in log, log[-2000:]                 # the model was placed on the shim's device
    cpu, _ = _serve_and_complete(server, model, 0, tmp_path, "cpu", env={"HIP_VISIBLE_DEVICES": "-1"})
    assert cpu.get("tokens_predicted") == 8
    assert gpu["content"].split()[:4] == cpu["content"].split()[:4], (gpu["content"], cpu["content"])      # (greedy ties further out may flip on a random model: first tokens)


@pytest.mark.parametrize("ndev", [4, 8])
def test_split_mode_graph_four_and_eight_logical_devices(ndev, models, tmp_path):
    """-sm graph over 4 and 8 logical devices on the one physical GPU (GGML_CDNA4_FAKE_DEVICES): split buffers with 4 / 8 slices, per-device sub-graphs, GGML_OP_REDUCE over
    4 / 8 partials incl. its copy-only targets (op_params[4], reduce.cu:125-598), KV heads distributed over the devices (the `wide` model: 32 / 8 heads, 4096-wide rows).
    With 8 devices every device holds ONE KV head; the logits must match the single-device run of the same kernels up to the summation order across the slices."""
    env = {"GGML_CDNA4_FAKE_DEVICES": str(ndev)}
    many = logits(models["wide"], 99, 48, 3, sm="graph", env=env, tmp=str(tmp_path)); one = logits(models["wide"], 99, 48, 3, tmp=str(tmp_path))
    cpu = logits(models["wide"], 0, 48, 3, tmp=str(tmp_path))
    for i in range(many.shape[0]):
        assert nmse(many[i], one[i]) < NMSE_VS_CPU, (ndev, i, nmse(many[i], one[i]))
        assert nmse(many[i], cpu[i]) < NMSE_VS_CPU, (ndev, i, nmse(many[i], cpu[i]))


def test_split_mode_graph_unequal_tensor_split(models, tmp_path):
    """-sm graph with an uneven -ts 3,1,2,2 over four logical devices: slice boundaries that are not equal shares (llama-load-tensors.cpp:5452-5499 rounds them to the
    split granularity), a device with a single KV head"""
    env = {"GGML_CDNA4_FAKE_DEVICES": "4", "LLAMA_LOGITS_TENSOR_SPLIT": "3,1,2,2"}
    many = logits(models["wide"], 99, 48, 3, sm="graph", env=env, tmp=str(tmp_path)); one = logits(models["wide"], 99, 48, 3, tmp=str(tmp_path))
    for i in range(many.shape[0]):
        assert nmse(many[i], one[i]) < NMSE_VS_CPU, (i, nmse(many[i], one[i]))


def test_supports_op_agrees_with_the_entry_points_on_the_gpu():
    """tests/shim_fuzz_case.py on the REAL runtime (its CPU twin runs on the stand-in runtime, where kernels do nothing): randomized one-op graphs over 13 op families, every node
    supports_op accepts is launched on zeroed buffers and synchronized -- a launch geometry the runtime refuses, a kernel that faults on an edge shape or an entry point that refuses
    what supports_op accepted aborts the child (the last `computing ...` line names the node)"""
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "shim_fuzz_case.py"), "40", "7"], capture_output=True, text=True, timeout=900)
    tail = p.stdout[-1500:]
    assert p.returncode == 0, "child exit %d\n%s\n%s" % (p.returncode, tail, p.stderr[-3000:])
    assert "offered / accepted per op" in tail and p.stdout.count("computing ") > 200
