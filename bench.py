#!/usr/bin/env python3
"""bench.py -- the BASELINE.json metric on the hot path: llama-bench pp512 + tg128 for Llama-3-8B Q4_K_M, restricted to
the quantized mat-mul path this repo implements (every MUL_MAT / FUSED_UP_GATE / MUL_MAT_ID of the model graph, nothing else).

    python bench.py --gpus N --steps K --warmup W [--config c2|c3|c4shard|c5]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A "step" = one llama-bench repetition of the mat-mul path: one prompt pass (pp: every weight matrix applied to a 512-column
activation batch per ubatch, output.weight to the last token only, like llama-bench) followed by 128 single-token passes (tg128).
Weights: random-bit quant blocks with finite scales in the type mix src/llama-quantize.cpp produces for the config, all resident
in HBM before the timed region; activations: synthetic N(0,1) f32.  value = tokens / step time (whole job, all ranks).

Configs (BASELINE.json `configs`; the headline metric is quoted on c2, which is what the default run times):
    c2       Llama-3-8B Q4_K_M, pp512 + tg128                      (configs[1])
    c3       Llama-3-8B, IQ2_M-style mix: IQ2_S / IQ3_S / Q6_K      (configs[2], sub-4-bit LUT kernels)
    c4shard  Llama-3-70B Q4_K_M, the per-GPU shard of TP=8, pp2048 + tg128 (configs[3]; with --gpus 8 the real TP run)
    c5       Mixtral-8x7B Q4_K_M, MUL_MAT_ID + MOE_FUSED_UP_GATE, top-2 of 8 experts, seeded uniform ids (configs[4])
With N = 1 and the default config the JSON line also carries `configs`: a short run (1 step) of c3 / c4shard / c5 with the roofline
of each config's dominant kernel, so that the driver sees them (`--no-extra-configs` skips them).

N > 1: tensor parallel exactly like the reference's `-sm graph`: q/k/v/up/gate row-split, o/down K-split, one
all-reduce(sum) of the [n_embd x tokens] partials after o and after down, output.weight replicated.  The reduce goes through the
one-launch all-reduce over IPC-mapped windows (cdna4_window_*; f32 per token, bf16 on the wire per prompt ubatch, converted inside the
launch) when the windows validate against the collective library at start-up -- the decode pass is then captured in a HIP graph like the
single-GPU one -- and through RCCL over xGMI (C ABI communicator, eager launches) otherwise or with CDNA4_BENCH_REDUCE=rccl.
Total work is fixed => "scaling": "strong".

Extra objects on the JSON line: `roofline` (dominant kernel of the config = the fused up*gate decode GEMV, HIP-event timed live over
the layers' distinct weights; `traffic` = FETCH_SIZE of the same launches collected by a rocprofv3 --pmc child run of this script)
and `cpu_baseline` (the REAL reference CPU kernels from oracle/_ref driven by OpenMP on this host, bounded sample; rank 0, N=1 only)."""
import argparse
import csv
import glob
import json
import os
import re
import shutil
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
# (No OMP_PROC_BIND / OMP_PLACES here: libgomp then pins THIS process's main thread -- the one that launches every kernel -- to the first core of the affinity mask, and
#  every child process inherits that one-core mask: on a shared host that core is the busiest one, and a 64-thread llama-bench child ran on a single core, 16x slow.
#  Only the op-level CPU sample wants a pinned team; it runs in a child of its own, `--cpu-op-child`.)

import numpy as np   # noqa: E402
import torch         # noqa: E402

from __graft_entry__ import _load_package   # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E peak (MI355X_MICROARCH.md "Chip-level parameters"); the ONE place this constant lives
MFMA_F16_PEAK_TFLOPS = 2500.0  # dense f16/bf16 MFMA peak

Q4_K, Q5_K, Q6_K, IQ4_NL, IQ3_S, IQ2_S, Q8_2_X4 = 12, 13, 14, 20, 21, 22, 99           # enum ggml_type of this fork (ggml.h)
TYPE_SIZE = {Q4_K: 144, Q5_K: 176, Q6_K: 210, IQ2_S: 82, IQ3_S: 110, IQ4_NL: 18}
BLCK_SIZE = {Q4_K: 256, Q5_K: 256, Q6_K: 256, IQ2_S: 256, IQ3_S: 256, IQ4_NL: 32}
D_OFFS = {Q4_K: (0, 2), Q5_K: (0, 2), Q6_K: (208,), IQ2_S: (0,), IQ3_S: (0,), IQ4_NL: (0,)}
TYPE_NAME = {Q4_K: "Q4_K", Q5_K: "Q5_K", Q6_K: "Q6_K", IQ2_S: "IQ2_S", IQ3_S: "IQ3_S", IQ4_NL: "IQ4_NL"}
N_GEN = 128             # tokens generated per step unless the config says otherwise (cfg["n_gen"])


def use_more_bits(i, n):       # src/llama-quantize.cpp:312-314
    return i < n // 8 or i >= 7 * n // 8 or (i - n // 8) % 3 == 2


def q4_k_m_types(name, il, nl):
    """LLAMA_FTYPE_MOSTLY_Q4_K_M (src/llama-quantize.cpp:631-632,731-737): attn_v / ffn_down -> Q6_K on the use_more_bits layers, output -> Q6_K"""
    if name == "output":
        return Q6_K
    if name in ("wv", "down") and use_more_bits(il, nl):
        return Q6_K
    return Q4_K


def iq2_m_types(name, il, nl):
    """LLAMA_FTYPE_MOSTLY_IQ2_M restricted to this repo's types (src/llama-quantize.cpp:507-545): default IQ2_S; attn_output -> IQ3_S;
    ffn_down of the first n/8 layers -> IQ3_S; attn_v -> Q6_K (the reference bumps it to IQ4_K at gqa >= 4 -- an ik-quant outside
    SURVEY 8a -- Q6_K stands in for it); output.weight -> Q6_K (reference: Q5_K)."""
    if name in ("output", "wv"):
        return Q6_K
    if name == "wo" or (name == "down" and il < nl // 8):
        return IQ3_S
    return IQ2_S


def iq4_nl_types(name, il, nl):
    """LLAMA_FTYPE_MOSTLY_IQ4_NL: IQ4_NL everywhere, output.weight (= the tied token embedding) -> Q6_K"""
    return Q6_K if name == "output" else IQ4_NL


CONFIGS = {
    # Qwen3-0.6B (BASELINE configs[0]; hparams of the released model: n_embd 1024, n_ff 3072, 16 heads / 8 KV heads x 128, 28 layers, vocab 151936, tied embeddings)
    "c1": dict(name="Qwen3-0.6B IQ4_NL", n_embd=1024, n_ff=3072, n_head=16, n_head_kv=8, head_dim=128, n_layer=28, n_vocab=151936, n_expert=0, n_used=0,
               types=iq4_nl_types, n_prompt=128, n_gen=32, shard=1),
    # Llama-3-8B (n_embd 4096, n_ff 14336, 32 heads / 8 KV heads x 128, 32 layers, vocab 128256)
    "c2": dict(name="Llama-3-8B Q4_K_M", n_embd=4096, n_ff=14336, n_head_kv=8, head_dim=128, n_layer=32, n_vocab=128256, n_expert=0, n_used=0,
               types=q4_k_m_types, n_prompt=512, shard=1),
    "c3": dict(name="Llama-3-8B IQ2_M-style mix (IQ2_S / IQ3_S / Q6_K)", n_embd=4096, n_ff=14336, n_head_kv=8, head_dim=128, n_layer=32, n_vocab=128256,
               n_expert=0, n_used=0, types=iq2_m_types, n_prompt=512, shard=1),
    # Llama-3-70B: 8192 / 28672 / 64.8 x 128 / 80 layers; the per-GPU shard of TP = 8
    "c4shard": dict(name="Llama-3-70B Q4_K_M, per-GPU shard of TP=8", n_embd=8192, n_ff=28672, n_head_kv=8, head_dim=128, n_layer=80, n_vocab=128256,
                    n_expert=0, n_used=0, types=q4_k_m_types, n_prompt=2048, shard=8),
    # Mixtral-8x7B: 4096 / 14336, 8 experts top-2, 32 layers, vocab 32000
    "c5": dict(name="Mixtral-8x7B Q4_K_M", n_embd=4096, n_ff=14336, n_head_kv=8, head_dim=128, n_layer=32, n_vocab=32000, n_expert=8, n_used=2,
               types=q4_k_m_types, n_prompt=512, shard=1),
}
GEMM_KERNELS = ("gemm_mfma_kernel", "gemm_wlds_kernel", "gemm_pp_kernel", "gemm_ppf_kernel")      # the prompt GEMM instantiations a kernel trace may show
N_UBATCH = 512          # llama-bench default n_ubatch (common/common.h:296-297): a longer prompt runs as ubatches of 512


def synth_weights(t, m, k, gen, device):
    """random-bit blocks with finite fp16 super-block scales (any byte pattern is a valid block)."""
    ts = TYPE_SIZE[t]; nb = k // BLCK_SIZE[t]
    w = torch.randint(0, 256, (m, nb, ts), dtype=torch.uint8, device=device, generator=gen)
    for off in D_OFFS[t]:
        d = (torch.rand((m, nb), device=device, generator=gen) * 0.02 + 1e-3).to(torch.float16)
        if off != 2:
            d = d * (torch.randint(0, 2, (m, nb), device=device, generator=gen) * 2 - 1).to(torch.float16)
        w[:, :, off:off + 2] = d.view(torch.uint8).view(m, nb, 2)
    return w.view(m, nb * ts)


class Model:
    """The mat-mul path of one forward pass of a Llama / Mixtral graph, sharded for tensor-parallel rank `rank` of `world`
    (`cfg["shard"]` > 1 on one process: the shapes of one rank of that TP degree, no collectives)."""

    def __init__(self, be, cfg, rank, world, device, n_layer=None):
        self.be, self.cfg, self.rank, self.world, self.dev = be, cfg, rank, world, device
        s = self.shard = world if world > 1 else cfg["shard"]
        self.E, self.NF, self.NL, self.NV = cfg["n_embd"], cfg["n_ff"], (n_layer or cfg["n_layer"]), cfg["n_vocab"]
        self.KV = cfg["n_head_kv"] * cfg["head_dim"]; self.n_expert, self.n_used = cfg["n_expert"], cfg["n_used"]
        self.QD = cfg.get("n_head", cfg["n_embd"] // cfg["head_dim"]) * cfg["head_dim"]        # rows of wq = columns of wo (Qwen3: != n_embd)
        self.emit_q8 = os.environ.get("CDNA4_BENCH_EMIT_Q8", "0") == "1"   # measured neutral at N = 1, harmful on TP shards (profiles/r01_notes.md)
        gen = torch.Generator(device=device); gen.manual_seed(1234 + rank)
        assert cfg["n_head_kv"] % s == 0 and (self.NF // s) % 256 == 0 and (self.E // s) % 256 == 0
        ty = cfg["types"]; nl = cfg["n_layer"]; E, NF, KV, QD = self.E, self.NF, self.KV, self.QD
        self.layers = []
        for il in range(self.NL):
            L = dict(
                wq=(ty("wq", il, nl), synth_weights(ty("wq", il, nl), QD // s, E, gen, device)),
                wk=(ty("wk", il, nl), synth_weights(ty("wk", il, nl), KV // s, E, gen, device)),
                wv=(ty("wv", il, nl), synth_weights(ty("wv", il, nl), KV // s, E, gen, device)),
                wo=(ty("wo", il, nl), synth_weights(ty("wo", il, nl), E, QD // s, gen, device)),         # K-split
            )
            tu, td = ty("up", il, nl), ty("down", il, nl)
            if self.n_expert:       # experts are split the same way INSIDE each expert (llama-build-context.cpp:1726-1735)
                L["up"] = (tu, torch.stack([synth_weights(tu, NF // s, E, gen, device) for _ in range(self.n_expert)]))
                L["gate"] = (tu, torch.stack([synth_weights(tu, NF // s, E, gen, device) for _ in range(self.n_expert)]))
                L["down"] = (td, torch.stack([synth_weights(td, E, NF // s, gen, device) for _ in range(self.n_expert)]))
            else:
                L["up"] = (tu, synth_weights(tu, NF // s, E, gen, device))
                L["gate"] = (tu, synth_weights(tu, NF // s, E, gen, device))
                L["down"] = (td, synth_weights(td, E, NF // s, gen, device))                               # K-split
            self.layers.append(L)
        self.output = (ty("output", 0, nl), synth_weights(ty("output", 0, nl), self.NV, E, gen, device))   # replicated (llama-build-context.cpp:2499-2530)
        self.bufs = {}

    def weight_bytes(self):
        n = self.output[1].numel()
        for L in self.layers:
            n += sum(v[1].numel() for v in L.values())
        return n

    def token_weight_bytes(self):
        """weight bytes one generated token reads (MoE: n_used of n_expert experts)"""
        n = self.output[1].numel()
        for L in self.layers:
            for k, v in L.items():
                n += v[1].numel() * self.n_used // self.n_expert if (self.n_expert and k in ("up", "gate", "down")) else v[1].numel()
        return n

    def _buf(self, name, n, *shape):
        key = (name, n)
        if key not in self.bufs:
            self.bufs[key] = torch.empty(shape, dtype=torch.float32, device=self.dev)
        return self.bufs[key]

    def prepare(self, n):
        """activations for a batch of n columns (synthetic, fixed) + output buffers; nothing is allocated in the timed region."""
        g = torch.Generator(device=self.dev); g.manual_seed(99 + n)
        s = self.shard; E, NF, KV, QD = self.E, self.NF, self.KV, self.QD
        self.bufs[("x", n)] = torch.randn((n, E), device=self.dev, generator=g)           # layer input (after norm)
        self.bufs[("attn", n)] = torch.randn((n, QD // s), device=self.dev, generator=g)   # attention output slice (wo input)
        self.bufs[("x1", n)] = torch.randn((1, E), device=self.dev, generator=g)
        for name, m in (("q", QD // s), ("k", KV // s), ("v", KV // s), ("o", E)):
            self._buf(name, n, n, m)
        if self.n_expert:
            ids = torch.stack([torch.randperm(self.n_expert, device=self.dev, generator=g)[:self.n_used] for _ in range(n)]).to(torch.int32)
            self.bufs[("ids", n)] = ids.contiguous()                                       # seeded uniform top-k routing
            self.bufs[("x3", n)] = self.bufs[("x", n)].view(n, 1, E)
            self._buf("ffn", n, n, self.n_used, NF // s); self._buf("down", n, n, self.n_used, E)
        else:
            self._buf("ffn", n, n, NF // s); self._buf("down", n, n, E)
        self._buf("logits", 1, 1, self.NV)
        if n == 1 and not self.n_expert and (NF // s) % 128 == 0:          # decode: the fused up*gate launch can also emit ffn_down's int8 input (cdna4_fused_up_gate_q8)
            self.bufs[("ffn_q8", 1)] = torch.empty((1, (NF // s) // 128 * 144), dtype=torch.uint8, device=self.dev)
        if n > 32 and self.world > 1:                # prompt-size partial sums travel as bf16 (reduce_type, llama-build-context.cpp:1198-1200)
            self.bufs[("red16", n)] = torch.empty((n, E), dtype=torch.bfloat16, device=self.dev)
        self.be.reserve_workspace((n * max(1, self.n_used) + 512) * max(NF // s, E) * 2 + (16 << 20))

    def reduce(self, t, n):
        """GGML_OP_REDUCE of a [n, n_embd] partial sum: f32 for n <= 32, bf16 on the wire for prompt batches like the reference
        (src/llama.cpp:8147,8227-8242: reduce_type)."""
        r16 = self.bufs.get(("red16", n))
        if r16 is None or os.environ.get("CDNA4_BENCH_REDUCE_F32") == "1":
            self.be.reduce(t)
        elif getattr(self.be, "window", None) and r16.numel() * 2 <= self.be.window_bytes:    # bf16 on the wire, converted inside the one-shot launch
            self.be.reduce(t, wire=torch.bfloat16)
        else:
            r16.copy_(t); self.be.reduce(r16); t.copy_(r16)

    def ffn(self, L, n):
        be = self.be; x = self.bufs[("x", n)]
        if self.n_expert:       # MOE_FUSED_UP_GATE -> MUL_MAT_ID (the weighted sum over the used experts is not a mat-mul)
            ids = self.bufs[("ids", n)]
            f = be.moe_fused_up_gate(L["up"][0], L["up"][1], L["gate"][1], self.bufs[("x3", n)], ids, out=self.bufs[("ffn", n)])
            return be.mul_mat_id(L["down"][0], L["down"][1], f, ids, out=self.bufs[("down", n)])[:, 0]
        if n == 1 and self.emit_q8:
            f, fq = be.fused_up_gate_q8(L["up"][0], L["up"][1], L["gate"][1], x, out=self.bufs[("ffn", n)], q8_out=self.bufs[("ffn_q8", 1)])
            return be.mul_mat(L["down"][0], L["down"][1], fq, out=self.bufs[("down", n)], x_type=Q8_2_X4)
        f = be.fused_up_gate(L["up"][0], L["up"][1], L["gate"][1], x, out=self.bufs[("ffn", n)])
        return be.mul_mat(L["down"][0], L["down"][1], f, out=self.bufs[("down", n)])

    def forward(self, n, last_only_logits):
        """all mat-muls of one forward pass over n columns, in graph order (llm_build_llama: q,k,v -> o -> fused up*gate -> down)."""
        be = self.be; x = self.bufs[("x", n)]; attn = self.bufs[("attn", n)]
        for L in self.layers:
            # q,k,v share src1: one call; same-type matrices are served by one decode launch (ggml.c:17984-18000 fuses them too)
            be.mul_mat_multi([L["wq"][0], L["wk"][0], L["wv"][0]], [L["wq"][1], L["wk"][1], L["wv"][1]], x,
                             outs=[self.bufs[("q", n)], self.bufs[("k", n)], self.bufs[("v", n)]])
            o = be.mul_mat(L["wo"][0], L["wo"][1], attn, out=self.bufs[("o", n)])
            if self.world > 1:
                self.reduce(o, n)                                               # GGML_OP_REDUCE after attention-out
            d = self.ffn(L, n)
            if self.world > 1:
                self.reduce(d, n)                                               # GGML_OP_REDUCE after ffn-down
        if last_only_logits is not None:
            xl = self.bufs[("x1", n)] if last_only_logits or n == 1 else x
            be.mul_mat(self.output[0], self.output[1], xl, out=self.bufs[("logits", 1)])


# ---- reproducibility record (VERDICT r02 "make the measurement reproducible"): runtime versions and the GPU's clocks / power / temperature
_SYSFS_OF_DEVICE = {}


def _sysfs_gpu_dir(idx):
    """/sys/class/drm/cardN/device of HIP device `idx`, matched by PCI bus id (a node shows the cards of ALL its GPUs in sysfs, also the ones this container may not
    use: the n-th card is not the n-th HIP device)"""
    if idx in _SYSFS_OF_DEVICE:
        return _SYSFS_OF_DEVICE[idx]
    import ctypes
    found = None
    try:
        hip = ctypes.CDLL("libamdhip64.so"); buf = ctypes.create_string_buffer(64)
        if hip.hipDeviceGetPCIBusId(buf, 64, idx) == 0:
            bus = buf.value.decode().lower()
            for d in glob.glob("/sys/class/drm/card[0-9]*/device"):
                if os.path.realpath(d).lower().endswith(bus) and os.path.exists(os.path.join(d, "pp_dpm_sclk")):
                    found = d; break
    except OSError:
        pass
    _SYSFS_OF_DEVICE[idx] = found
    return found


def _read(path):
    try:
        with open(path) as f:
            return f.read().strip()
    except OSError:
        return None


def _cur_level_mhz(txt):
    """pp_dpm_* lists `N: 2400Mhz *`; the starred line is the current level"""
    if not txt:
        return None
    for line in txt.splitlines():
        if line.rstrip().endswith("*"):
            m = re.search(r"(\d+)\s*[Mm][Hh][Zz]", line)
            if m:
                return int(m.group(1))
    return None


def gpu_sample(idx=0):
    """one cheap sample straight from sysfs (no subprocess): sclk / mclk / fclk MHz, socket power W, junction temperature C"""
    d = _sysfs_gpu_dir(idx)
    if d is None:
        return None
    out = {"sysfs": d.split("/")[4]}
    for key, f in (("sclk_mhz", "pp_dpm_sclk"), ("mclk_mhz", "pp_dpm_mclk"), ("fclk_mhz", "pp_dpm_fclk")):
        out[key] = _cur_level_mhz(_read(os.path.join(d, f)))
    for hw in glob.glob(os.path.join(d, "hwmon", "hwmon*")):
        for key, f, scale in (("power_w", "power1_average", 1e-6), ("power_w", "power1_input", 1e-6), ("temp_c", "temp2_input", 1e-3), ("temp_c", "temp1_input", 1e-3),
                              ("power_cap_w", "power1_cap", 1e-6)):
            if out.get(key) is None:
                v = _read(os.path.join(hw, f))
                if v and v.lstrip("-").isdigit():
                    out[key] = round(int(v) * scale, 1)
    out["perf_level"] = _read(os.path.join(d, "power_dpm_force_performance_level"))
    return out


class GpuSampler:
    """samples gpu_sample() every `period` s on a thread while the timed region runs; reports min / max / mean per field"""
    def __init__(self, idx=0, period=0.25):
        import threading
        self.idx, self.period, self.samples, self._stop = idx, period, [], threading.Event()
        self._t = threading.Thread(target=self._run, daemon=True)

    def _run(self):
        while not self._stop.is_set():
            s = gpu_sample(self.idx)
            if s:
                self.samples.append(s)
            self._stop.wait(self.period)

    def __enter__(self):
        self._t.start(); return self

    def __exit__(self, *a):
        self._stop.set(); self._t.join(timeout=2)

    def summary(self):
        out = {"n_samples": len(self.samples)}
        for key in ("sclk_mhz", "mclk_mhz", "fclk_mhz", "power_w", "temp_c"):
            v = [s[key] for s in self.samples if s.get(key) is not None]
            if v:
                out[key] = {"min": min(v), "max": max(v), "mean": round(sum(v) / len(v), 1)}
        return out


def env_info():
    """runtime versions of THIS process (torch ships its own libamdhip64: whichever HIP runtime is loaded first serves the process) and of /opt/rocm"""
    import ctypes
    info = {"torch": torch.__version__, "torch_hip": getattr(torch.version, "hip", None), "rocm_dir_version": _read("/opt/rocm/.info/version"),
            "kernel": _read("/proc/sys/kernel/osrelease"), "amdgpu_driver": _read("/sys/module/amdgpu/version")}
    try:
        hip = ctypes.CDLL("libamdhip64.so")
        v = ctypes.c_int(0)
        if hip.hipRuntimeGetVersion(ctypes.byref(v)) == 0:
            info["hip_runtime_version"] = v.value
        if hip.hipDriverGetVersion(ctypes.byref(v)) == 0:
            info["hip_driver_version"] = v.value
    except OSError:
        pass
    try:
        for line in open("/proc/self/maps"):
            if "libamdhip64" in line:
                info["libamdhip64"] = line.split()[-1]; break
    except OSError:
        pass
    cpu = None
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                cpu = line.split(":", 1)[1].strip(); break
    except OSError:
        pass
    info["host_cpu"] = cpu; info["host_logical_cpus"] = os.cpu_count()
    try:
        info["host_affinity_cpus"] = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        pass
    info["cgroup_cpu_max"] = _read("/sys/fs/cgroup/cpu.max")          # "quota period" or "max period": a CPU quota makes every multi-threaded host leg (and a busy host) slow
    info["loadavg"] = _read("/proc/loadavg")
    return info


def _usable_cpus():
    """logical CPUs this process may really use: the affinity mask, capped by a cgroup CPU quota"""
    try:
        n = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        n = os.cpu_count() or 1
    q = (_read("/sys/fs/cgroup/cpu.max") or "max").split()
    if len(q) == 2 and q[0] != "max":
        try:
            n = max(1, min(n, int(float(q[0]) / float(q[1]) + 0.999)))
        except ValueError:
            pass
    return n


def _host_threads():
    n = _usable_cpus()
    return max(1, min(64, n // 2 if n > 16 else n))           # one thread per physical core (SMT siblings only fight over the same AVX-512 units), at most 64


_GGUF_DIR = None


def synth_gguf(kind, log):
    """the synthetic GGUF of `kind` ("llama3-8b-q4km" | "qwen3-0.6b-iq4nl"), written once per bench run under /tmp (tests/gguf_synth.py; removed at exit)"""
    global _GGUF_DIR
    import atexit
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import gguf_synth
    if _GGUF_DIR is None:
        _GGUF_DIR = tempfile.mkdtemp(prefix="cdna4_gguf_", dir="/tmp")
        atexit.register(shutil.rmtree, _GGUF_DIR, True)
    path = os.path.join(_GGUF_DIR, kind + ".gguf")
    if kind.endswith("-L32") and not os.path.exists(path):      # the 26 GB Mixtral-shaped file: memory-backed when the host has room (written outside every timed region, removed at exit)
        try:
            st = os.statvfs("/dev/shm")
            if st.f_bavail * st.f_frsize > (96 << 30):
                d = tempfile.mkdtemp(prefix="cdna4_gguf_", dir="/dev/shm"); atexit.register(shutil.rmtree, d, True)
                path = os.path.join(d, kind + ".gguf"); _BIG_GGUF[kind] = path
        except OSError:
            pass
    path = _BIG_GGUF.get(kind, path)
    if not os.path.exists(path):
        t0 = time.time()
        if kind == "llama3-8b-q4km":
            gguf_synth.bench_model(path)
        elif kind == "qwen3-0.6b-iq4nl":
            gguf_synth.qwen3_06b_model(path)
        elif kind == "llama3-8b-iq2m":          # BASELINE configs[2]: the type mix of CONFIGS["c3"] (iq2_m_types) under the tensor names of the file
            names = {"attn_q": "wq", "attn_k": "wk", "attn_v": "wv", "attn_output": "wo", "ffn_up": "up", "ffn_gate": "gate", "ffn_down": "down", "output": "output", "token_embd": "output"}
            gguf_synth.bench_model(path, types=lambda name, il, nl: iq2_m_types(names[name], il, nl), seed=3, name="Llama-3-8B-IQ2_M-mix-synth")
        elif kind.startswith("mixtral-8x7b-q4km-L"):      # BASELINE configs[4] shapes with the first L layers (the full 32-layer file is 26 GB: see llama_bench_layers)
            gguf_synth.bench_model(path, n_embd=4096, n_ff=14336, n_head=32, n_head_kv=8, n_layer=int(kind.rsplit("L", 1)[1]), n_vocab=32000, n_expert=8, n_used=2, seed=5,
                                   name="Mixtral-8x7B-synth")
        else:
            raise ValueError(kind)
        log("synthetic GGUF %s written in %.1f s" % (kind, time.time() - t0))
    return path


_CPU_THREADS = None
_BIG_GGUF = {}


def cpu_llama_threads(log):
    """thread count for the CPU legs: the best of {n, n/2, n/4} (n = _host_threads()) on a 2-second probe -- llama-bench tg16 of the small Qwen3-shaped GGUF.  The hosts of this
    pool differ (CPU quotas, busy neighbours); an over-subscribed OpenMP team is 10-20x slower than a fitting one, and a wrong guess would make the baseline meaningless."""
    global _CPU_THREADS
    if _CPU_THREADS is not None:
        return _CPU_THREADS
    n = _host_threads(); best, best_ts = n, 0.0
    try:
        model = synth_gguf("qwen3-0.6b-iq4nl", log)
        for t in sorted({n, max(1, n // 2), max(1, n // 4)}, reverse=True):
            r = run_llama_bench(lambda *a: None, model, 0, 16, 1, gpu=False, threads=t, timeout=25)
            ts = r["tg16_tok_s"] if r else 0.0
            log("cpu thread probe: -t %d -> tg16 %.1f tok/s" % (t, ts))
            if ts > best_ts * 1.05:
                best, best_ts = t, ts
    except Exception as e:
        log("cpu thread probe failed: %r" % (e,))
    _CPU_THREADS = best
    return best


def run_llama_bench(log, model, n_prompt, n_gen, reps, gpu, threads=8, timeout=600, extra_env=None, extra_args=()):
    """the reference's OWN llama-bench binary (unmodified sources, built by ik_llama.cpp_amd/backend/Makefile.llama).  gpu = True: -ngl 99 -fa 1 through the
    backend shim (KV cache in HBM, every node on the device); gpu = False: -ngl 0 with the GPU hidden (HIP_VISIBLE_DEVICES=-1: the shim reports 0 devices), i.e.
    the reference's CPU backend (iqk_mul_mat, iqk flash attention) on this host.  Returns None when the binary is absent or the run fails."""
    exe = os.path.join(ROOT, "oracle", "_ref", "llama", "bin", "llama-bench")
    if not os.path.exists(exe):
        log("llama-bench leg skipped: %s not built" % exe); return None
    env = dict(os.environ)
    if gpu:
        env["GGML_CDNA4_STATS"] = "1"
    else:
        env["HIP_VISIBLE_DEVICES"] = "-1"; env.pop("ROCR_VISIBLE_DEVICES", None)
        env.pop("OMP_PLACES", None); env.pop("OMP_PROC_BIND", None)        # (this script pins its own OpenMP team; the child places its threads itself)
    env.update(extra_env or {})
    cmd = [exe, "-m", model, "-p", str(n_prompt), "-n", str(n_gen), "-ngl", "99" if gpu else "0", "-fa", "1", "-t", str(threads), "-r", str(reps), "-o", "json"] + list(extra_args)
    try:
        t0 = time.time()
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout, env=env)
        wall = time.time() - t0
        err = r.stderr.decode(errors="replace")
        if r.returncode != 0:
            log("llama-bench failed rc=%d: %s" % (r.returncode, err[-400:])); return None
        txt = r.stdout.decode(errors="replace"); res = json.loads(txt[txt.index("["):])
        pps = [x for x in res if x["n_prompt"] > 0]; tgs = [x for x in res if x["n_gen"] > 0]
        pp = pps[0] if pps else None; tg = tgs[0] if tgs else None; any_ = pp or tg
        t_total = (n_prompt / pp["avg_ts"] if pp else 0.0) + (n_gen / tg["avg_ts"] if tg else 0.0)
        out = {"cmd": " ".join(["llama-bench"] + cmd[3:]) + ("" if gpu else "  (HIP_VISIBLE_DEVICES=-1)"),
               "model": "%s, %.2f GiB, %d params" % (any_.get("model_type", "?"), any_["model_size"] / 2 ** 30, any_["model_n_params"]),
               "value": round((n_prompt + n_gen) / t_total, 1), "unit": "tok/s", "wall_s": round(wall, 1),
               "cpu_info": any_.get("cpu_info", "").strip(), "gpu_info": any_.get("gpu_info", "").strip()}
        if pp:
            out["pp%d_tok_s" % n_prompt] = round(pp["avg_ts"], 1); out["pp_stddev"] = round(pp["stddev_ts"], 1)
        if tg:
            out["tg%d_tok_s" % n_gen] = round(tg["avg_ts"], 1); out["tg_stddev"] = round(tg["stddev_ts"], 2)
        if gpu:         # GGML_CDNA4_STATS lines of the shim: graph replay counts and host-side time (diagnoses a tg gap between boxes)
            stats = [ln.strip() for ln in err.splitlines() if ln.startswith("cdna4[")]
            out["shim_stats"] = stats[-6:]
            fl = [ln for ln in stats if "fused launches issued or captured: ADD" in ln]
            if fl:      # fusions taken by the last context (the tg leg): name -> launches issued or captured
                out["fusions"] = {k.strip(): int(v) for k, v in re.findall(r"([A-Za-z_+,. \-]+?) (\d+)(?:,|$)", fl[-1].split("captured:", 1)[1])}
            ms = re.findall(r"(\d+) eager, (\d+) captured, (\d+) replayed, (\d+) capture failures, (\d+) too small", err)
            if ms:      # one line per context: the prompt test's (prompt-size graphs run eagerly: counted as "not captured"), then the generation test's
                m = max(ms, key=lambda t: int(t[2]))
                out["graphs"] = {"eager": int(m[0]), "captured": int(m[1]), "replayed": int(m[2]), "capture_failures": int(m[3]), "not_captured": int(m[4])}
        return out
    except Exception as e:
        log("llama-bench leg failed: %r" % (e,)); return None


def cpu_baseline(log, cfg, gguf_kind="llama3-8b-q4km", n_prompt=512, n_gen=128, op_level=True, reps=3, timeout=150):
    """The reference CPU path timed on this host.  Primary number: the reference's own `llama-bench -ngl 0` on the same synthetic GGUF the GPU
    leg runs (whole model, 3 repetitions).  Secondary (`op_level`): the real iqk_mul_mat kernels of oracle/_ref on the mat-mul sequence alone (what `value`
    times on the GPU), sampled over a few layers and extrapolated."""
    nth = cpu_llama_threads(log)
    lb = None
    try:       # bounded: a healthy host needs 10-25 s for this leg; a host that cannot do it in 150 s reports no whole-model number rather than stalling the run
        lb = run_llama_bench(log, synth_gguf(gguf_kind, log), n_prompt, n_gen, reps, gpu=False, threads=nth, timeout=timeout)
    except Exception as e:
        log("cpu_baseline llama-bench leg failed: %r" % (e,))
    opl = None
    if op_level:        # in a child process with a pinned OpenMP team (see the note at the top of this file)
        try:
            env = dict(os.environ, OMP_PROC_BIND="close", OMP_PLACES="cores", CDNA4_CPU_THREADS=str(nth))
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-op-child"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=240, env=env)
            txt = r.stdout.decode(errors="replace").strip().splitlines()
            opl = json.loads(txt[-1]) if r.returncode == 0 and txt else None
            if opl is None:
                log("cpu op-level child failed rc=%d: %s" % (r.returncode, r.stderr.decode(errors="replace")[-300:]))
        except Exception as e:
            log("cpu op-level child failed: %r" % (e,))
    if lb is None and opl is None:
        return None
    out = {"value": lb["value"] if lb else opl["value"], "unit": "tok/s", "cores": nth, "kind": "reference",
           "sample": ("reference llama-bench (unmodified sources, CPU backend: iqk_mul_mat + iqk flash attention), whole model, -p %d -n %d -r %d -t %d, GPU hidden"
                      % (n_prompt, n_gen, reps, nth)) if lb else opl["sample"]}
    if lb:
        out["pp%d_tok_s" % n_prompt] = lb["pp%d_tok_s" % n_prompt]; out["tg%d_tok_s" % n_gen] = lb["tg%d_tok_s" % n_gen]
        out["llama_bench"] = lb
    if opl:
        out["op_level"] = opl
    return out


def cpu_op_level(log, cfg):
    """The reference CPU kernels (oracle/_ref, real iqk_mul_mat incl. its N>=32 repack path) on the mat-mul sequence alone: bounded sample =
    tg over 8 distinct layers + pp512 over 4 layers (median of 5), extrapolated to the 32-layer model (+ output.weight)."""
    try:
        from oracle import bindings as ob
        if ob.ref_path() is None:
            return None
        ref = ob.Ref(); orc = ob.Oracle()
    except Exception as e:      # no reference library on this host
        log("cpu_baseline unavailable: %r" % (e,)); return None
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from common import random_block_bytes
    E, NF, NL, NV, NP = cfg["n_embd"], cfg["n_ff"], cfg["n_layer"], cfg["n_vocab"], cfg["n_prompt"]
    nth = int(os.environ.get("CDNA4_CPU_THREADS", "0")) or _host_threads()
    shapes = [("wq", Q4_K, E, E), ("wk", Q4_K, 1024, E), ("wv", Q6_K, 1024, E), ("wo", Q4_K, E, E),
              ("up", Q4_K, NF, E), ("gate", Q4_K, NF, E), ("down", Q6_K, E, NF)]
    n_tg_layers = 8                        # 1.16 GB of distinct weights per sweep: well beyond the host's L3, like a real token
    layers = [[(t, random_block_bytes(t, m, k, 7 * li + i), m, k) for i, (_, t, m, k) in enumerate(shapes)] for li in range(n_tg_layers)]
    rng = np.random.default_rng(0)

    def run(n, layer_list, reps, stat):
        times = []
        acts = {}
        for k in (E, NF):
            x = rng.standard_normal((n, k)).astype(np.float32)
            acts[k] = {vdt: ref.quantize_activations(vdt, x) for vdt in (ob.Q8_2_X4,)}
        outs = {m: np.zeros((n, m), np.float32) for m in (E, 1024, NF)}
        for _ in range(reps):
            t0 = time.perf_counter()
            for L in layer_list:
                for (t, w, m, k) in L:
                    vdt = ob.vec_dot_type(t)
                    ref.mul_mat_omp(orc, t, w, acts[k][vdt], vdt, n, k, outs[m], nth)
            times.append(time.perf_counter() - t0)
        return stat(times) / len(layer_list)     # seconds per layer
    med = lambda v: float(np.median(v))
    run(1, layers, 2, min)                                        # warm the thread team
    t_tg_layer = run(1, layers, 15, med)
    run(NP, layers[:4], 2, min)                                   # first touch of the work buffers, thread placement settled
    t_pp_layer = run(NP, layers[:4], 5, med)
    wout = random_block_bytes(Q6_K, NV, E, 99)
    xq = ref.quantize_activations(ob.Q8_2_X4, rng.standard_normal((1, E)).astype(np.float32)); lo = np.zeros((1, NV), np.float32)
    t_out = 1e30
    for _ in range(3):
        t0 = time.perf_counter(); ref.mul_mat_omp(orc, Q6_K, wout, xq, ob.Q8_2_X4, 1, E, lo, nth); t_out = min(t_out, time.perf_counter() - t0)
    t_tg = NL * t_tg_layer + t_out
    t_pp = NL * t_pp_layer + t_out
    total = t_pp + N_GEN * t_tg
    return {"value": round((NP + N_GEN) / total, 2), "unit": "tok/s", "cores": nth,
            "pp512_tok_s": round(NP / t_pp, 1), "tg128_tok_s": round(1.0 / t_tg, 2),
            "sample": "reference iqk_mul_mat (oracle/_ref %s build) on the mat-mul sequence alone: tg timed over %d distinct layers "
                      "(median of 15 sweeps) and pp512 over 4 layers (median of 5 after 2 warm-ups) + output.weight, extrapolated x%d layers; %d OpenMP threads"
                      % (ref.variant, n_tg_layers, NL, nth)}


def _abort_capture(stream):
    """A capture that failed half way (e.g. a collective that cannot be captured) leaves the stream -- and, in thread-local mode, this
    thread -- in capture state, and every later allocation fails with hipErrorStreamCaptureUnsupported.  End it explicitly."""
    import ctypes
    try:
        hip = ctypes.CDLL("libamdhip64.so")
        status = ctypes.c_int(0)
        hip.hipStreamIsCapturing(ctypes.c_void_p(stream.cuda_stream), ctypes.byref(status))
        if status.value != 0:
            gph = ctypes.c_void_p(0)
            hip.hipStreamEndCapture(ctypes.c_void_p(stream.cuda_stream), ctypes.byref(gph))
            if gph.value:
                hip.hipGraphDestroy(gph)
        hip.hipGetLastError()
    except Exception:
        pass


# ---- the dominant decode kernel of a config: the fused up*gate launch (dense: FUSED_UP_GATE; MoE: MOE_FUSED_UP_GATE over the used experts)
def norm_variant_available(model):
    """does the library serve FUSED_RMS_NORM + FUSED_UP_GATE of one token as ONE launch for this model's row length?  (rows of up to 4096 values: the whole row sits in the
    mat-vec's pre-loaded chunks; the 8192-wide rows of the 70B shard keep the stand-alone norm -- the shim issues two launches there, and so does this harness)"""
    if getattr(model, "norm_ok", None) is None:
        x1 = model.bufs[("x", 1)]; ffn = model.bufs[("ffn", 1)]; L = model.layers[0]
        g = torch.Generator(device=x1.device); g.manual_seed(11)
        model.norm_w = (torch.rand((model.E,), device=x1.device, generator=g) + 0.5).contiguous()
        try:
            model.be.fused_up_gate_norm(L["up"][0], L["up"][1], L["gate"][1], x1, model.norm_w, out=ffn); model.norm_ok = True
        except Exception:      # noqa: BLE001 (CDNA4_E_UNSUPPORTED)
            model.norm_ok = False
    return model.norm_ok


def dominant_sweep(model, n_layers=None, norm=False):
    """norm = True (dense models): the launch the timed llama-bench run issues for this op -- ffn_norm rides in the mat-vec's prologue (cdna4_fused_up_gate_fused with
    cdna4_fusion.norm_w: the FX = 1 instantiation of gemv_kernel); False: the plain FUSED_UP_GATE launch of the mat-mul harness"""
    be = model.be; x1 = model.bufs[("x", 1)]; ffn = model.bufs[("ffn", 1)]
    layers = model.layers[:n_layers] if n_layers else model.layers
    if model.n_expert:
        ids = model.bufs[("ids", 1)]; x3 = model.bufs[("x3", 1)]
        def sweep():
            for L in layers:
                be.moe_fused_up_gate(L["up"][0], L["up"][1], L["gate"][1], x3, ids, out=ffn)
    elif norm and norm_variant_available(model):
        nw = model.norm_w
        def sweep():
            for L in layers:
                be.fused_up_gate_norm(L["up"][0], L["up"][1], L["gate"][1], x1, nw, out=ffn)
    else:
        def sweep():
            for L in layers:
                be.fused_up_gate(L["up"][0], L["up"][1], L["gate"][1], x1, out=ffn)
    return sweep, len(layers)


def dominant_bytes(model):
    """M*K*bpw/8 per matrix read + 4*K*N + 4*M*N (N = 1; MoE: n_used experts x (up + gate))"""
    t = model.layers[-1]["up"][0]; m_loc = model.NF // model.shard; nmat = 2 * (model.n_used or 1)
    return nmat * m_loc * (model.E // BLCK_SIZE[t]) * TYPE_SIZE[t] + 4 * model.E + 4 * m_loc * (model.n_used or 1), t, m_loc


def measure_traffic(config, log):
    """HBM bytes per launch of the dominant kernel from the PMC counters: a child run of this script under
    `rocprofv3 --pmc FETCH_SIZE --kernel-trace` (counters cannot be read in-process; --pmc with kernel-trace only, as the pool requires).
    gfx950 correction: FETCH_SIZE reports exactly 1/2 of the bytes of a wide coalesced stream (MI355X_MICROARCH.md, HBM section)."""
    prof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(prof):
        return None, "rocprofv3 not found"
    tmp = tempfile.mkdtemp(prefix="cdna4_pmc_")
    try:
        env = dict(os.environ); env["TMPDIR"] = tmp; env["CDNA4_HANDOFF_SELFTEST"] = "0"      # (the library's start-up self-test launches a small split-K GEMM: keep it out of the child's trace)
        cmd = ["timeout", "150", prof, "--pmc", "FETCH_SIZE", "--kernel-trace", "--output-format", "csv", "-d", tmp, "-o", "pmc", "--",
               sys.executable, os.path.abspath(__file__), "--pmc-child", "--config", config]
        r = subprocess.run(cmd, cwd=tmp, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=200)
        files = glob.glob(os.path.join(tmp, "**", "*counter_collection.csv"), recursive=True)
        if r.returncode != 0 or not files:
            return None, "rocprofv3 child rc=%d: %s" % (r.returncode, r.stderr.decode(errors="replace")[-300:])
        acc = {}
        for row in csv.DictReader(open(files[0])):
            if row.get("Counter_Name") == "FETCH_SIZE":
                acc.setdefault(row["Kernel_Name"], []).append(float(row["Counter_Value"]))
        best = None
        for k, v in acc.items():
            if "gemv_kernel" in k and (best is None or sum(v) > sum(acc[best])):
                best = k
        if best is None:
            return None, "no gemv dispatch in the counter file"
        by = int(round(2 * 1024 * sum(acc[best]) / len(acc[best])))
        src = {"kernel": best, "dispatches": len(acc[best]), "FETCH_SIZE_KB_avg": round(sum(acc[best]) / len(acc[best]), 2),
               "method": "live: rocprofv3 --pmc FETCH_SIZE --kernel-trace child of this run; bytes = 2 * 1024 * FETCH_SIZE (gfx950 wide-stream correction)"}
        # the same trace holds the prompt GEMM launches of the child (one ubatch each): the KERNEL's own duration, without the f32 -> f16 activation image the op runs first
        try:
            kt = glob.glob(os.path.join(tmp, "**", "*kernel_trace.csv"), recursive=True)
            dur = {}
            for row in csv.DictReader(open(kt[0])):
                if any(k in row.get("Kernel_Name", "") for k in GEMM_KERNELS):
                    dur.setdefault(row["Kernel_Name"], []).append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) * 1e-3)
            ks = []
            for name, d in dur.items():
                v = sorted(d)[:-1] if len(d) > 2 else d      # (drop the slowest: the first, cold launch)
                ks.append({"kernel": name, "dispatches": len(d), "avg_us": round(sum(v) / len(v), 2)})
            ks.sort(key=lambda e: e["avg_us"])
            if ks:      # the child launches the op at one ubatch and (headline config) at 4096 tokens: different instances of the kernel, the shorter one is the ubatch
                src["prefill_kernel"] = ks[0]
                if len(ks) > 1:
                    src["prefill_kernel_4096"] = ks[-1]
        except Exception as e:      # noqa: BLE001
            src["prefill_kernel"] = {"error": repr(e)[:120]}
        return by, src
    except Exception as e:
        return None, repr(e)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def pmc_child(args):
    """the profiled child: a few launches of the dominant kernel of `--config` on distinct weights, nothing else"""
    cfg = CONFIGS[args.config]
    device = torch.device("cuda", 0); torch.cuda.set_device(0)
    pkg = _load_package(); be = pkg.Cdna4Backend(0)
    model = Model(be, cfg, 0, 1, device, n_layer=16 if not cfg["n_expert"] else 4)
    model.prepare(1)
    dense = (not cfg["n_expert"]) and norm_variant_available(model)
    # the launch the timed run issues (dense: norm-carrying), as ONE dependent chain on the stream: 16 sweeps x 16 layers = 256 dispatches over 1 GB of distinct weights
    # (beyond the 256 MB Infinity Cache, as in a real token); the readers drop the first 32 (clock / cache warm-up).  Then the harness variant, 64 dispatches.
    sweep, nl = dominant_sweep(model, norm=dense)
    for _ in range(max(1, 256 // nl)):
        sweep()
    torch.cuda.synchronize()
    if dense:
        sweep_plain, _ = dominant_sweep(model, norm=False)
        for _ in range(4):
            sweep_plain()
        torch.cuda.synchronize()
    # + the prompt form of the same op (one ubatch) on the 4 layers' weights: its GEMM kernel's duration is read from the kernel trace
    nub = min(cfg["n_prompt"], N_UBATCH); model.prepare(nub)
    L0 = model.layers[0]
    for L in (model.layers[:4] * 2 if not model.n_expert else model.layers * 2):
        if model.n_expert:
            be.moe_fused_up_gate(L["up"][0], L["up"][1], L["gate"][1], model.bufs[("x3", nub)], model.bufs[("ids", nub)], out=model.bufs[("ffn", nub)])
        else:
            be.fused_up_gate(L["up"][0], L["up"][1], L["gate"][1], model.bufs[("x", nub)], out=model.bufs[("ffn", nub)])
    torch.cuda.synchronize()
    if args.config == "c2" and not model.n_expert:      # + the 4096-token prompt of BASELINE.json (one ubatch of pp4096)
        n4k = 4096; g4 = torch.Generator(device=device); g4.manual_seed(7)
        x4 = torch.randn((n4k, model.E), device=device, generator=g4); f4 = torch.empty((n4k, model.layers[0]["up"][1].shape[0]), device=device)
        be.reserve_workspace(n4k * model.E * 2 + (8 << 20))
        for L in model.layers[:4] * 4:       # 16 launches: the first ones run while the clocks still ramp
            be.fused_up_gate(L["up"][0], L["up"][1], L["gate"][1], x4, out=f4)
        torch.cuda.synchronize()
    be.close()


def run_config(args, key, be, rank, world, device, log, steps, warmup, full):
    """time one config; returns its result dict (rank 0) -- `full`: cpu baseline / pmc traffic / 4k prefill extras (the headline config)"""
    import torch.distributed as dist
    cfg = CONFIGS[key]
    NP = cfg["n_prompt"]; nub = min(NP, N_UBATCH); n_ubatches = NP // nub
    NG = cfg.get("n_gen", N_GEN)
    model = Model(be, cfg, rank, world, device)
    model.prepare(nub); model.prepare(1)
    log("[%s] weights resident: %.3f GB on rank 0 (%s)" % (key, model.weight_bytes() / 1e9, be.description()))

    # ---- decode pass captured in a HIP graph (~130 launches / token)
    graph = None
    dbg_dev = os.environ.get("CDNA4_BENCH_DEBUG_ONE_DEVICE")
    model.forward(1, False); torch.cuda.synchronize()
    # N > 1: eager launches by default.  With 64 all-reduces per token the host (193 calls, ~8.5 us each) is about as fast as the
    # devices, and a capture that fails inside a collective can leave the process unusable.  CDNA4_BENCH_TP_GRAPH=1 captures the
    # all-reduces with the kernels (RCCL supports stream capture; thread-local capture mode so that RCCL's helper threads cannot
    # invalidate it).  (gloo's CUDA path joins its own streams into a capture and cannot be captured: the debug mode runs eagerly)
    # With the IPC-window reduce (one capturable launch, no collective library inside the capture) the TP decode pass is captured like the single-GPU one.
    windows = bool(getattr(be, "window", None))
    if not args.no_graph and (world == 1 or ((os.environ.get("CDNA4_BENCH_TP_GRAPH", "1" if windows else "0") == "1") and (dbg_dev is None or windows))):
        cap_stream = torch.cuda.Stream(device=device)
        try:
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=cap_stream, capture_error_mode="thread_local" if world > 1 else "global"):
                model.forward(1, False)
            graph = g                # (first replay only after all ranks agreed below: a replay runs the captured collectives)
        except Exception as e:
            log("HIP graph capture of the decode pass failed (%r): running eagerly" % (str(e)[:200],))
            graph = None
            _abort_capture(cap_stream)
            try:
                torch.cuda.synchronize()
            except Exception:
                pass
    if world > 1:           # every rank must take the same path
        flag = torch.tensor([1 if graph is not None else 0], device=device)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) == 0:
            graph = None
    if graph is not None:
        graph.replay(); torch.cuda.synchronize()

    # eager decode: replay a recorded call plan (arguments marshalled once) instead of going through the Python wrappers every token
    plan = None
    if graph is None and getattr(be.reduce, "__self__", None) is be:
        with be.record() as plan:
            model.forward(1, False)
        torch.cuda.synchronize()

    # prompt ubatches: eager launches replayed from a recorded call plan as well (arguments marshalled once: ~2 us of host time per call instead of ~8.5 us through
    # the Python wrappers -- 225 calls per ubatch; keeps the prompt pass GPU-bound on a busy host)
    pp_plans = {}
    if getattr(be.reduce, "__self__", None) is be and os.environ.get("CDNA4_BENCH_NO_PP_PLAN") is None:
        for flag in ([None, True] if n_ubatches > 1 else [True]):
            with be.record() as pl:
                model.forward(nub, flag)
            pp_plans[flag] = pl
        torch.cuda.synchronize()

    def decode_token():
        if graph is not None:
            graph.replay()
        elif plan is not None:
            plan.replay(be._check)
        else:
            model.forward(1, False)

    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]

    host_ms = {"pp": [], "tg": []}             # host time spent SUBMITTING each phase (no sync inside): a phase whose submit time approaches its GPU time is host-bound

    def step():
        ev[0].record()
        h0 = time.perf_counter()
        for ub in range(n_ubatches):           # pp: ubatches of 512; output.weight only for the last token of the prompt (llama-bench)
            flag = True if ub == n_ubatches - 1 else None
            if flag in pp_plans:
                pp_plans[flag].replay(be._check)
            else:
                model.forward(nub, flag)
        h1 = time.perf_counter()
        ev[1].record()
        for _ in range(NG):                    # tg128
            decode_token()
        h2 = time.perf_counter()
        ev[2].record()
        host_ms["pp"].append((h1 - h0) * 1e3); host_ms["tg"].append((h2 - h1) * 1e3)

    def sync_all():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(warmup):
        step()
    sync_all()
    host_ms["pp"].clear(); host_ms["tg"].clear()
    sampler = GpuSampler(device.index or 0) if (full and rank == 0) else None      # clocks / power / temperature while the timed region runs (sysfs reads on a thread)
    if sampler:
        sampler.__enter__()
    t0 = time.perf_counter()
    pp_ms = tg_ms = 1e-9
    pp_list, tg_list = [], []
    for _ in range(steps):
        step()
        torch.cuda.synchronize()
        a, b = ev[0].elapsed_time(ev[1]), ev[1].elapsed_time(ev[2])
        pp_ms += a; tg_ms += b; pp_list.append(a); tg_list.append(b)
    sync_all()
    elapsed = time.perf_counter() - t0
    if sampler:
        sampler.__exit__()
    tt = torch.tensor([elapsed, pp_ms, tg_ms], dtype=torch.float64, device=device)
    if world > 1:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    elapsed, pp_ms, tg_ms = [float(v) for v in tt.cpu()]
    steps_ = max(steps, 1)
    ms_per_step = elapsed * 1e3 / steps_
    tokens = NP + NG

    # ---- N > 1: the exchange step on its own ("reduce time broken out")
    reduce_info = None
    if world > 1:
        o1 = model.bufs[("o", 1)]; opp = model.bufs[("o", nub)]
        o1.zero_(); opp.zero_()                      # (repeated in-place sums of zeros stay finite)
        er0, er1, er2 = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        for _ in range(4):
            model.reduce(o1, 1); model.reduce(opp, nub)
        sync_all()
        er0.record()
        for _ in range(2 * model.NL):
            model.reduce(o1, 1)
        er1.record()
        for _ in range(2 * model.NL):
            model.reduce(opp, nub)
        er2.record(); sync_all()
        rt = torch.tensor([er0.elapsed_time(er1), er1.elapsed_time(er2)], dtype=torch.float64, device=device)
        dist.all_reduce(rt, op=dist.ReduceOp.MAX)
        r_tok, r_pp = [float(v) for v in rt.cpu()]
        # which path carried each message class, and every available path timed on the two classes (SURVEY 8e "reduce time broken out"; the first run on real multi-GPU
        # hardware must say by itself whether the IPC windows, the C-ABI communicator or torch.distributed did the work): eager calls, max over ranks
        two_shot_min = int(os.environ.get("CDNA4_WINDOW_TWO_SHOT_MIN", 256 * 1024))
        def carried(nbytes):
            if windows and nbytes <= be.window_bytes:
                return "ipc-window " + ("two-shot (reduce-scatter + all-gather)" if (world > 2 and nbytes >= two_shot_min) else "one-shot")
            return "C-ABI communicator (RCCL)" if getattr(be, "comm", None) else "torch.distributed (RCCL)"
        classes = {"token_f32": (o1, None, o1.numel() * 4), "prompt_ubatch_bf16_wire": (opp, torch.bfloat16, opp.numel() * 2)}
        paths = {}
        if windows:
            paths["ipc_window"] = lambda b, w: be.window_reduce(b, wire=w) if b.numel() * (2 if w is not None else b.element_size()) <= be.window_bytes else None
        if getattr(be, "comm", None):
            def _rccl(b, w):
                if w is not None:
                    t16 = b.to(w); be._check(be.lib.cdna4_all_reduce_sum(be.comm, t16.data_ptr(), t16.numel(), 30, be._stream())); b.copy_(t16)      # 30 = GGML_TYPE_BF16
                else:
                    be._check(be.lib.cdna4_all_reduce_sum(be.comm, b.data_ptr(), b.numel(), 0, be._stream()))
                return b
            paths["c_abi_rccl"] = _rccl
        def _torch(b, w):
            if w is not None:
                t16 = b.to(w); dist.all_reduce(t16); b.copy_(t16)
            else:
                dist.all_reduce(b)
            return b
        if dbg_dev is None:
            paths["torch_distributed"] = _torch
        table = {}
        for cname, (buf, wire, nbytes) in classes.items():
            row = {"bytes_on_wire": nbytes, "carried_by": carried(nbytes)}
            for pname, fn in paths.items():
                try:
                    buf.zero_()
                    if fn(buf, wire) is None:
                        row[pname + "_us"] = None; continue
                    sync_all(); er0.record()
                    for _ in range(20):
                        fn(buf, wire)
                    er1.record(); sync_all()
                    tt_ = torch.tensor([er0.elapsed_time(er1) * 1e3 / 20], dtype=torch.float64, device=device); dist.all_reduce(tt_, op=dist.ReduceOp.MAX)
                    row[pname + "_us"] = round(float(tt_.item()), 2)
                except Exception as e:      # noqa: BLE001 -- a path that fails is reported, the run goes on (every rank takes the same branch: the paths exist on all or none)
                    row[pname + "_us"] = "failed: %r" % (str(e)[:120],)
            table[cname] = row
        reduce_info = {"per_token_ms": round(r_tok, 4), "per_ubatch_ms": round(r_pp, 4), "reduces_per_pass": 2 * model.NL, "paths": table,
                       "share_of_tg_time": round(r_tok / max(tg_ms / (steps_ * NG), 1e-9), 4),
                       "share_of_pp_time": round(r_pp * n_ubatches / max(pp_ms / steps_, 1e-9), 4),
                       "wire": "f32 [1, %d] per token; bf16 [%d, %d] per prompt ubatch" % (model.E, nub, model.E)}

    # ---- roofline of the dominant kernel: the fused up*gate decode launch, timed with HIP events on the launch stream over the layers'
    # DISTINCT weights (cold L2 / Infinity Cache: >= 2 GB per sweep).
    # `frac` is quoted on the launch the TIMED run (llama-bench through the shim) issues for this op: dense models carry ffn_norm in the mat-vec's prologue (FX = 1 instantiation,
    # + 4 K bytes of norm weights); the plain launch of the mat-mul harness is reported beside it as `harness_variant` ("do this" 2a).  Both: 5 sweeps over the
    # layers' distinct weights, interleaved per sweep.
    dense = (not model.n_expert) and norm_variant_available(model)
    sweep, nlay = dominant_sweep(model, norm=dense)
    sweep_h, _ = dominant_sweep(model, norm=False)
    sweep(); sweep_h(); torch.cuda.synchronize()
    # recorded once, replayed for the timing: a replayed call costs the host ~2 us, the Python wrappers 10-20 us -- more than the 15 us launch they time, and the stream would
    # run dry between launches (the first version of this leg read 19.4 us from the events where the kernel trace of the same launches said 16.5)
    with be.record() as plan_n:
        sweep()
    with be.record() as plan_h:
        sweep_h()
    e0, e1, e2 = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    nsweep = 5 if full else 2
    k_ms = kh_ms = 0.0
    for _ in range(nsweep):
        e0.record(); plan_n.replay(be._check); e1.record()
        if dense:
            plan_h.replay(be._check)
        e2.record(); torch.cuda.synchronize()
        k_ms += e0.elapsed_time(e1) / (nsweep * nlay); kh_ms += e1.elapsed_time(e2) / (nsweep * nlay)
    alg_bytes_h, t_dom, m_loc = dominant_bytes(model)
    alg_bytes = alg_bytes_h + (4 * model.E if dense else 0)
    ach = alg_bytes / (k_ms * 1e-3) / 1e9
    traffic, traffic_src = None, None
    if rank == 0 and world == 1 and not args.no_pmc and (full or not args.no_pmc_extra):
        traffic, traffic_src = measure_traffic(key, log)
        if traffic is None:
            log("[%s] live PMC traffic unavailable (%s)" % (key, traffic_src))
            traffic_src = {"method": "unavailable in this run", "reason": str(traffic_src)[:200]}
    kname = ("moe fused up*gate id-GEMV <%s> %d experts x 2 x %dx%d" % (TYPE_NAME[t_dom], model.n_used, m_loc, model.E)) if model.n_expert else \
            (("gemv_kernel<%s,1,fused up*gate, FX=1: RMS norm in the prologue> %dx%d x2 -- the launch inside the timed llama-bench run" if dense else "gemv_kernel<%s,1,fused up*gate> %dx%d x2") % (TYPE_NAME[t_dom], m_loc, model.E))
    roofline = {"bound": "hbm", "kernel": kname, "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4),
                "traffic": traffic, "traffic_source": traffic_src, "bytes_per_launch": alg_bytes, "avg_launch_us": round(k_ms * 1e3, 2)}
    if dense:
        ach_h = alg_bytes_h / (kh_ms * 1e-3) / 1e9
        roofline["harness_variant"] = {"kernel": "gemv_kernel<%s,1,fused up*gate> %dx%d x2 (plain FUSED_UP_GATE: what the mat-mul harness `matmul_only` launches)" % (TYPE_NAME[t_dom], m_loc, model.E),
                                       "bytes_per_launch": alg_bytes_h, "avg_launch_us": round(kh_ms * 1e3, 2), "achieved": round(ach_h, 1), "frac": round(ach_h / HBM_PEAK_GBS, 4)}
    # whole decode token against HBM: the bytes one token must read / the measured time per token
    tok_bytes = model.token_weight_bytes()
    tg_tok_ms = tg_ms / (steps_ * NG)
    roofline["decode_token"] = {"weight_bytes": tok_bytes, "ms": round(tg_tok_ms, 4), "achieved": round(tok_bytes / (tg_tok_ms * 1e-3) / 1e9, 1),
                                "frac": round(tok_bytes / (tg_tok_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}

    # secondary: the prefill MFMA kernel on the same weights (one ubatch)
    xp = model.bufs[("x", nub)]; ffp = model.bufs[("ffn", nub)]
    if model.n_expert:
        idp = model.bufs[("ids", nub)]; x3p = model.bufs[("x3", nub)]
        def pf(L):
            be.moe_fused_up_gate(L["up"][0], L["up"][1], L["gate"][1], x3p, idp, out=ffp)
        fl = 2.0 * 2 * m_loc * model.E * nub * model.n_used
    else:
        def pf(L):
            be.fused_up_gate(L["up"][0], L["up"][1], L["gate"][1], xp, out=ffp)
        fl = 2.0 * 2 * m_loc * model.E * nub
    pf(model.layers[0]); torch.cuda.synchronize()
    npf = min(len(model.layers), 32 if full else 8)
    # the launches of `npf` layers back to back as ONE captured graph (device time of the op = activation image + GEMM, without the host's launch gaps: eager, the two launches of
    # an op leave ~20 us of idle device per op on a busy host); eager launches if the capture is refused
    pf_graph = None
    try:
        st_pf = torch.cuda.Stream(device=device); pf_graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(pf_graph, stream=st_pf, capture_error_mode="thread_local"):
            for L in model.layers[:npf]:
                pf(L)
        pf_graph.replay(); torch.cuda.synchronize()
    except Exception as e:      # noqa: BLE001
        log("prefill roofline: graph capture refused (%r), eager launches" % (e,)); pf_graph = None; torch.cuda.synchronize()
    pf_sampler = GpuSampler(device.index or 0, period=0.02) if (full and rank == 0) else None      # clocks / power while the prompt GEMMs run (a clock-limited box shows here)
    if pf_sampler:
        pf_sampler.__enter__()
    try:      # (an exception in the timed launches or in the trace children must not leave the sampler thread running)
        e0.record()
        for _ in range(3 if full else 1):
            if pf_graph is not None:
                pf_graph.replay()
            else:
                for L in model.layers[:npf]:
                    pf(L)
        e1.record(); torch.cuda.synchronize()
        g_ms = e0.elapsed_time(e1) / npf / (3 if full else 1)
        roofline_prefill = {"bound": "mfma", "kernel": "%sgemm_mfma_kernel<%s,fused up*gate> N=%d" % ("grouped " if model.n_expert else "", TYPE_NAME[t_dom], nub),
                            "achieved": round(fl / (g_ms * 1e-3) / 1e12, 1), "peak": MFMA_F16_PEAK_TFLOPS, "unit": "TFLOP/s",
                            "frac": round(fl / (g_ms * 1e-3) / 1e12 / MFMA_F16_PEAK_TFLOPS, 4), "avg_launch_us": round(g_ms * 1e3, 1),
                            "timed": "HIP events around %d ops (f32 -> f16 activation image + GEMM each), %s" % (npf, "one captured graph" if pf_graph is not None else "eager launches")}
        # the whole prompt pass of the mat-mul harness against the MFMA roof: every weight matrix x the ubatch (MoE: the experts used), over the measured time per ubatch
        pass_flops = 0.0
        for L in model.layers:
            for kname, vv in L.items():
                rows = vv[1].shape[-2] * ((model.n_used) if (model.n_expert and kname in ("up", "gate", "down")) else 1)
                kcols = {"wq": model.E, "wk": model.E, "wv": model.E, "wo": model.QD // model.shard, "up": model.E, "gate": model.E, "down": model.NF // model.shard}[kname]
                pass_flops += 2.0 * rows * kcols * nub
        pp_ub_ms = pp_ms / (steps_ * n_ubatches)
        roofline_prefill["pp_pass_frac"] = round(pass_flops / (pp_ub_ms * 1e-3) / 1e12 / MFMA_F16_PEAK_TFLOPS, 4)
        roofline_prefill["pp_pass"] = {"gflop_per_ubatch": round(pass_flops / 1e9, 1), "ms_per_ubatch": round(pp_ub_ms, 3), "what": "all mat-muls of one %d-token ubatch (activation images included), mat-mul harness" % nub}
        pk = (traffic_src or {}).pop("prefill_kernel", None) if isinstance(traffic_src, dict) else None
        pk4 = (traffic_src or {}).pop("prefill_kernel_4096", None) if isinstance(traffic_src, dict) else None
        pk_method = "rocprofv3 --pmc FETCH_SIZE --kernel-trace child of this run (same trace as roofline.traffic; the counter run inflates long kernels by ~5 %)"
        if rank == 0 and world == 1 and not args.no_pmc and (full or not args.no_pmc_extra):
            kt = gemm_kernel_trace(key, log)       # the same child under a PLAIN kernel trace: the durations the MFMA fraction is quoted on
            if kt and kt.get("prefill_kernel"):
                pk = kt["prefill_kernel"]; pk4 = kt.get("prefill_kernel_4096") or pk4
                pk_method = "rocprofv3 --kernel-trace child of this run (no counters)"
            if kt and kt.get("decode_kernel"):
                dk = kt["decode_kernel"]
                vs_ev = dk["avg_us"] / (k_ms * 1e3)
                roofline["kernel_trace"] = dict(dk, frac=round(alg_bytes / (dk["avg_us"] * 1e-6) / 1e9 / HBM_PEAK_GBS, 4), vs_hip_events=round(vs_ev, 4),
                                                agreement=("within 3 %" if abs(vs_ev - 1.0) <= 0.03 else
                                                           "outside 3 %: the trace child is another PROCESS under the profiler (its own clock / power settling: config.gpu_during_timed_region), "
                                                           "timed minutes after the sweep; the same kernel's average inside the llama-bench trace of this run is in decode_token.llama_bench_trace.top5"),
                                                method="rocprofv3 --kernel-trace child: >= 200 launches of the timed variant as one dependent chain over 16 layers' distinct weights, "
                                                       "the first 32 dropped (cross-check of avg_launch_us, HIP events: vs_hip_events = trace / events)")
                if kt.get("decode_kernel_harness") and "harness_variant" in roofline:
                    roofline["harness_variant"]["kernel_trace_avg_us"] = kt["decode_kernel_harness"]["avg_us"]
        if pk and "avg_us" in pk:      # the GEMM kernel alone (rocprofv3 kernel trace of the PMC child): what the MFMA roof applies to; `frac` above is the whole op (activation image + GEMM), HIP events
            roofline_prefill["kernel_only"] = {"kernel": pk["kernel"], "avg_us": pk["avg_us"], "dispatches": pk["dispatches"], "achieved": round(fl / (pk["avg_us"] * 1e-6) / 1e12, 1),
                                               "frac": round(fl / (pk["avg_us"] * 1e-6) / 1e12 / MFMA_F16_PEAK_TFLOPS, 4), "method": pk_method}
        if full and not model.n_expert:
            # BASELINE.json states the MFMA target on a 4k-token prefill: same fused launch at N = 4096 (one ubatch of pp4096), 4 layers' weights
            try:
                n4k = 4096
                g4 = torch.Generator(device=device); g4.manual_seed(7)
                x4 = torch.randn((n4k, model.E), device=device, generator=g4); f4 = torch.empty((n4k, m_loc), device=device)
                be.reserve_workspace(n4k * model.E * 2 + 2 * (m_loc + 256) * model.E * 2 + (16 << 20))      # activation image + the f16 weight image of the large-batch route
                L0 = model.layers[0]
                fl4 = 2.0 * 2 * m_loc * model.E * n4k

                def time_form(form):
                    be.set_gemm_form(form)
                    be.fused_up_gate(L0["up"][0], L0["up"][1], L0["gate"][1], x4, out=f4); torch.cuda.synchronize()
                    e0.record()
                    for L in model.layers[:12]:
                        be.fused_up_gate(L["up"][0], L["up"][1], L["gate"][1], x4, out=f4)
                    e1.record(); torch.cuda.synchronize()
                    return e0.elapsed_time(e1) / 12, be.last_launch_info().get("kernel")
                # both prompt-GEMM forms, interleaved (default, per-wave, default, per-wave): boxes differ by 10-25 % on this kernel and the clocks ramp during the first launches, so only
                # a same-run interleaved pair says which form is faster here; `frac` is the DEFAULT form's best pass
                try:
                    runs = [time_form(f) for f in (1, 0, 2, 1, 0, 2)]
                finally:
                    be.set_gemm_form(1)
                g4_ms = min(runs[0][0], runs[3][0]); g4b = min(runs[1][0], runs[4][0]); g4c = min(runs[2][0], runs[5][0])
                roofline_prefill["n4096"] = {"achieved": round(fl4 / (g4_ms * 1e-3) / 1e12, 1), "frac": round(fl4 / (g4_ms * 1e-3) / 1e12 / MFMA_F16_PEAK_TFLOPS, 4),
                                             "avg_launch_us": round(g4_ms * 1e3, 1), "kernel_form": runs[0][1], "passes_us": [round(r[0] * 1e3, 1) for r in runs],
                                             "op_is": "f32 -> f16 activation image + (large-batch route: weights -> f16 image once +) GEMM, HIP events over 12 layers' weights, best of two interleaved passes per form",
                                             "per_wave_form": {"kernel_form": runs[1][1], "avg_launch_us": round(g4b * 1e3, 1), "frac": round(fl4 / (g4b * 1e-3) / 1e12 / MFMA_F16_PEAK_TFLOPS, 4)},
                                             "shared_tile_form": {"kernel_form": runs[2][1], "avg_launch_us": round(g4c * 1e3, 1), "frac": round(fl4 / (g4c * 1e-3) / 1e12 / MFMA_F16_PEAK_TFLOPS, 4)}}
                if pk4 and "avg_us" in pk4:
                    roofline_prefill["n4096"]["kernel_only"] = {"kernel": pk4["kernel"], "avg_us": pk4["avg_us"], "dispatches": pk4["dispatches"],
                                                                "achieved": round(fl4 / (pk4["avg_us"] * 1e-6) / 1e12, 1), "frac": round(fl4 / (pk4["avg_us"] * 1e-6) / 1e12 / MFMA_F16_PEAK_TFLOPS, 4)}
                del x4, f4
            except Exception as e:      # (memory-constrained shard configurations): the N = 512 figure above stands alone
                log("4k-token prefill roofline skipped: %r" % (e,))

    finally:
        if pf_sampler:
            pf_sampler.__exit__()
    if pf_sampler:
        roofline_prefill["gpu_during_timing"] = pf_sampler.summary()
    cpu = None
    if full and rank == 0 and world == 1 and not args.no_cpu_baseline and key == "c2":
        cpu = cpu_baseline(log, cfg)

    nmat = 7 * model.NL + 1
    res = {
        "value": round(tokens * steps_ / elapsed, 2), "unit": "tok/s", "ms_per_step": round(ms_per_step, 3),
        "config": {"workload": "%s on %dxMI355X: pp%d + tg%d, every MUL_MAT/FUSED_UP_GATE%s of the graph (%d weight tensors, %.3f GB per rank), "
                               "no attention/norm/rope ops" % (cfg["name"], world, NP, NG, "/MUL_MAT_ID/MOE_FUSED_UP_GATE" if model.n_expert else "", nmat, model.weight_bytes() / 1e9),
                   "parallelism": ("tp%d (row-split q/k/v/up/gate, K-split o/down, %s all-reduce x2 per layer)" % (world, "one-shot IPC-window" if getattr(be, "window", None) else "RCCL")) if world > 1 else
                                  ("single GPU, shapes of one rank of tp%d, no collectives" % model.shard if model.shard > 1 else "single GPU"),
                   "type_mix": cfg["types"].__doc__.split("\n")[0].strip(),
                   "pp%d_tok_s" % NP: round(NP * steps_ / (pp_ms * 1e-3), 1), "tg%d_tok_s" % NG: round(NG * steps_ / (tg_ms * 1e-3), 1),
                   "pp_ms_min_median": [round(min(pp_list), 3), round(float(np.median(pp_list)), 3)] if pp_list else None,
                   "tg_ms_min_median": [round(min(tg_list), 3), round(float(np.median(tg_list)), 3)] if tg_list else None,
                   "host_submit_ms_median": {"pp": round(float(np.median(host_ms["pp"])), 3), "tg": round(float(np.median(host_ms["tg"])), 3)} if host_ms["tg"] else None,
                   "decode_hip_graph": graph is not None, "weight_bytes_per_rank": model.weight_bytes(), "reduce": reduce_info,
                   "gpu_during_timed_region": sampler.summary() if sampler else None},
        "roofline": roofline, "roofline_prefill": roofline_prefill, "cpu_baseline": cpu,
    }
    del model, graph, plan, pp_plans
    torch.cuda.empty_cache()
    return res


def ab_compare(args, pkg, be_new, device, log):
    """--ab-lib: the headline kernels timed with TWO builds of the library in ONE process (same box, same clocks, same weights, interleaved ABAB):
    settles whether a difference between two driver runs is the code or the box.  A = --ab-lib (e.g. an older git revision), B = the in-tree build."""
    be_old = pkg.Cdna4Backend(device.index or 0, lib_path=os.path.abspath(args.ab_lib))
    cfg = CONFIGS["c2"]
    model = Model(be_new, cfg, 0, 1, device, n_layer=8)
    model.prepare(512); model.prepare(1)
    be_old.reserve_workspace((512 + 512) * cfg["n_ff"] * 2 + (16 << 20))
    x1 = model.bufs[("x", 1)]; xp = model.bufs[("x", 512)]; f1 = model.bufs[("ffn", 1)]; fp = model.bufs[("ffn", 512)]
    d1 = model.bufs[("down", 1)]; dp = model.bufs[("down", 512)]

    def timed(fn, reps):
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3 / (reps * len(model.layers))       # us per launch

    cases = {
        "decode fused up*gate Q4_K 2x14336x4096": (lambda b: [b.fused_up_gate(L["up"][0], L["up"][1], L["gate"][1], x1, out=f1) for L in model.layers], 10),
        "decode ffn_down 4096x14336": (lambda b: [b.mul_mat(L["down"][0], L["down"][1], f1, out=d1) for L in model.layers], 10),
        "prefill fused up*gate N=512": (lambda b: [b.fused_up_gate(L["up"][0], L["up"][1], L["gate"][1], xp, out=fp) for L in model.layers], 3),
        "prefill ffn_down N=512": (lambda b: [b.mul_mat(L["down"][0], L["down"][1], fp, out=dp) for L in model.layers], 3),
    }
    out = {"A": os.path.abspath(args.ab_lib), "B": "in-tree build", "unit": "us per launch (HIP events, 8 layers' distinct weights, best of 3 interleaved rounds)", "cases": {}}
    for name, (fn, reps) in cases.items():
        ta, tb = [], []
        for _ in range(3):
            ta.append(timed(lambda: fn(be_old), reps)); tb.append(timed(lambda: fn(be_new), reps))
        out["cases"][name] = {"A": round(min(ta), 2), "B": round(min(tb), 2), "B_over_A": round(min(tb) / min(ta), 4)}
    be_old.close()
    del model
    torch.cuda.empty_cache()
    return out


def llama_bench_end_to_end(log, n_prompt=512, n_gen=128, reps=5, gguf_kind="llama3-8b-q4km", extra_args=()):
    """End to end through the boundary: the reference's own llama-bench on a full-size synthetic GGUF, -ngl 99 -fa 1 (run_llama_bench).  Reported beside `value`
    (which times the mat-mul path alone)."""
    try:
        r = run_llama_bench(log, synth_gguf(gguf_kind, log), n_prompt, n_gen, reps, gpu=True, extra_args=extra_args)
    except Exception as e:
        log("llama-bench end-to-end leg failed: %r" % (e,)); return None
    if r:
        r["harness"] = "reference llama-bench (unmodified sources, linked against libggml-cuda-cdna4.so), -ngl 99 -fa 1, %d repetitions" % reps
    return r


def _short_kernel(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n); m = re.match(r"(?:void )?([A-Za-z0-9_]+(?:<[^(]*>)?)", n)
    return (m.group(1) if m else n)[:64]


def llama_bench_trace(log, n_prompt, n_gen, gguf_kind="llama3-8b-q4km", timeout=240):
    """`rocprofv3 --kernel-trace` around ONE repetition of the reference's llama-bench through the shim (decode steps replayed from HIP graphs as in the timed run; no counters):
    per-kernel durations and the idle gaps between consecutive kernels, folded into a small record -- a slow box can then be diagnosed from the bench line alone (which kernels
    stretched: the clock-sensitive mat-vec / GEMM kernels or the gaps, i.e. the host).  The trace run itself is not timed."""
    prof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    exe = os.path.join(ROOT, "oracle", "_ref", "llama", "bin", "llama-bench")
    if not os.path.exists(prof) or not os.path.exists(exe):
        return None
    tmp = tempfile.mkdtemp(prefix="cdna4_trace_")
    try:
        env = dict(os.environ); env["TMPDIR"] = tmp
        cmd = ["timeout", str(timeout), prof, "--kernel-trace", "--output-format", "csv", "-d", tmp, "-o", "t", "--",
               exe, "-m", synth_gguf(gguf_kind, log), "-p", str(n_prompt), "-n", str(n_gen), "-ngl", "99", "-fa", "1", "-t", "8", "-r", "1", "-o", "json"]
        r = subprocess.run(cmd, cwd=tmp, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout + 30)
        files = glob.glob(os.path.join(tmp, "**", "*kernel_trace.csv"), recursive=True)
        if r.returncode != 0 or not files:
            log("kernel trace of llama-bench failed rc=%d: %s" % (r.returncode, r.stderr.decode(errors="replace")[-300:])); return None
        rows = [(int(x["Start_Timestamp"]), int(x["End_Timestamp"]), _short_kernel(x["Kernel_Name"])) for x in csv.DictReader(open(files[0]))]
        rows.sort()
        if n_gen > 0 and n_prompt == 0:
            # llama-bench runs a warm-up repetition first (one token; with -r 1 then n_gen tokens): keep the last n_gen tokens = the kernels behind the last warm-up launch
            # (token boundary = the lm-head mat-vec, the longest launch of a token)
            per_tok = None
            names = [k for _, _, k in rows]
            head = max(set(names), key=lambda k: max(e - s0 for s0, e, kk in rows if kk == k))          # the kernel with the longest single launch: output.weight
            idx = [i for i, k in enumerate(names) if k == head]
            if len(idx) >= n_gen + 1:
                first = idx[-n_gen - 1] + 1; rows_t = rows[first:idx[-1] + 1]; per_tok = n_gen
            else:
                rows_t = rows
        else:
            rows_t = rows; per_tok = None
        dur = {}; gap_sum = 0; prev_end = None
        for s0, e, k in rows_t:
            d = dur.setdefault(k, [0, 0]); d[0] += 1; d[1] += e - s0
            if prev_end is not None and 0 <= s0 - prev_end < 500000:
                gap_sum += s0 - prev_end
            prev_end = max(e, prev_end or 0)
        ksum = sum(v[1] for v in dur.values()); div = float(per_tok or 1)
        top = sorted(dur.items(), key=lambda kv: -kv[1][1])[:6]
        out = {"what": ("one decoded token (mean of the last %d of a llama-bench -n %d run, HIP-graph replays)" % (per_tok, n_gen)) if per_tok else ("llama-bench -p %d -n %d, whole run incl. its warm-up pass" % (n_prompt, n_gen)),
               "kernels": int(round(len(rows_t) / div)), "kernel_sum_us": round(ksum / div / 1e3, 1), "gap_sum_us": round(gap_sum / div / 1e3, 1),
               "span_us": round((rows_t[-1][1] - rows_t[0][0]) / div / 1e3, 1) if rows_t else None,
               "top": [{"kernel": k, "calls": int(round(v[0] / div)), "avg_us": round(v[1] / v[0] / 1e3, 2), "share": round(v[1] / max(ksum, 1), 3)} for k, v in top],
               "method": "rocprofv3 --kernel-trace child (no counters), gaps = start - previous end on the device timeline (< 0.5 ms)"}
        return out
    except Exception as e:      # noqa: BLE001
        log("kernel trace of llama-bench failed: %r" % (e,)); return None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def gemm_kernel_trace(config, log):
    """durations of the prompt GEMM kernels of the profiled child (`--pmc-child`) under a PLAIN kernel trace: the counter run of measure_traffic inflates the long kernels by ~5 %
    (profiles/r03_notes.md); this is the figure the MFMA fraction is quoted on.  Returns {"prefill_kernel": ..., "prefill_kernel_4096": ..., "decode_kernel": ...} or None."""
    prof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(prof):
        return None
    tmp = tempfile.mkdtemp(prefix="cdna4_ktrace_")
    try:
        env = dict(os.environ); env["TMPDIR"] = tmp; env["CDNA4_HANDOFF_SELFTEST"] = "0"
        cmd = ["timeout", "150", prof, "--kernel-trace", "--output-format", "csv", "-d", tmp, "-o", "kt", "--", sys.executable, os.path.abspath(__file__), "--pmc-child", "--config", config]
        r = subprocess.run(cmd, cwd=tmp, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=200)
        kt = glob.glob(os.path.join(tmp, "**", "*kernel_trace.csv"), recursive=True)
        if r.returncode != 0 or not kt:
            return None
        dur = {}
        rows = sorted(csv.DictReader(open(kt[0])), key=lambda r: int(r["Start_Timestamp"]))
        for row in rows:
            n = row.get("Kernel_Name", "")
            if any(k in n for k in GEMM_KERNELS) or "gemv_kernel" in n or "dequant_slab_kernel" in n:
                dur.setdefault(n, []).append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) * 1e-3)
        gem, gv, dq = [], [], []
        for name, d in dur.items():
            if "gemv_kernel" in name:      # a dependent chain of >= 200 launches of the timed variant (pmc_child): the first 32 (clock / cache warm-up) dropped
                v = d[32:] if len(d) >= 64 else (sorted(d)[:-1] if len(d) > 2 else d)
                gv.append({"kernel": name, "dispatches": len(d), "averaged_over": len(v), "avg_us": round(sum(v) / len(v), 2)})
            else:
                v = sorted(d)[:-1] if len(d) > 2 else d
                (dq if "dequant_slab_kernel" in name else gem).append({"kernel": name, "dispatches": len(d), "avg_us": round(sum(v) / len(v), 2)})
        gem.sort(key=lambda e: e["avg_us"]); gv.sort(key=lambda e: -e["dispatches"])
        out = {}
        if gem:
            out["prefill_kernel"] = gem[0]
            if len(gem) > 1:
                out["prefill_kernel_4096"] = dict(gem[-1])
                if dq and "gemm_ppf" in gem[-1]["kernel"]:      # the large-batch route: the weight-image pass belongs to the op (one launch each per mat-mul)
                    out["prefill_kernel_4096"]["weight_image_us"] = dq[-1]["avg_us"]
                    out["prefill_kernel_4096"]["avg_us"] = round(gem[-1]["avg_us"] + dq[-1]["avg_us"], 2)
                    out["prefill_kernel_4096"]["kernel"] = gem[-1]["kernel"] + " + " + dq[-1]["kernel"]
        if gv:
            out["decode_kernel"] = gv[0]
            if len(gv) > 1:
                out["decode_kernel_harness"] = gv[1]
        return out
    except Exception as e:      # noqa: BLE001
        log("plain kernel trace of the GEMM child failed: %r" % (e,)); return None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def llama_bench_mixtral(log, reps=3, cpu=True):
    """BASELINE configs[4] end to end, MEASURED (the round-5 record extrapolated from 4- and 8-layer files): the reference's llama-bench through the shim on a
    32-layer Mixtral-8x7B-shaped synthetic GGUF (26 GB, 8 experts top-2, Q4_K_M mix; 288 GB of HBM hold it whole), and the reference CPU backend on the same file."""
    t0 = time.time()
    model = synth_gguf("mixtral-8x7b-q4km-L32", log)
    r = run_llama_bench(log, model, 512, 128, reps, gpu=True, timeout=400)
    if not r:
        return None, None
    r["harness"] = "reference llama-bench through the shim, 32-layer Mixtral-8x7B-shaped synthetic GGUF (%.1f GB written in this run, outside every timed region), -p 512 -n 128 -ngl 99 -fa 1 -r %d" % (os.path.getsize(model) / 1e9, reps)
    c = None
    if cpu:
        c = cpu_baseline(log, CONFIGS["c5"], "mixtral-8x7b-q4km-L32", 512, 128, op_level=False, reps=1, timeout=300)
    log("c5 end-to-end leg (file + GPU + CPU runs): %.0f s" % (time.time() - t0))
    return r, c


def llama_bench_layers(log, reps=5):
    """BASELINE configs[4] end to end: the reference's llama-bench through the shim on Mixtral-8x7B-shaped synthetic GGUFs.  The full model is a 26 GB file; two files with the
    first 4 and 8 layers are written and timed instead, and the time per token is extrapolated linearly to 32 layers (t(L) = a + b L: every layer is identical, `a` carries the
    embedding, the lm head and the per-graph host work).  Both measured points and the extrapolation are reported, labelled as such."""
    pts = {}
    for nl in (4, 8):
        r = run_llama_bench(log, synth_gguf("mixtral-8x7b-q4km-L%d" % nl, log), 512, 128, reps, gpu=True, timeout=300)
        if not r:
            return None
        pts[nl] = r
    out = {"harness": "reference llama-bench through the shim, Mixtral-8x7B shapes (8 experts, top-2), Q4_K_M mix, -p 512 -n 128 -ngl 99 -fa 1 -r %d" % reps,
           "measured": {"L%d" % nl: {k: v for k, v in r.items() if k.endswith("_tok_s") or k in ("value", "model", "pp_stddev", "tg_stddev")} for nl, r in pts.items()}}
    ext = {}
    for key, n_tok in (("pp512_tok_s", 512), ("tg128_tok_s", 128)):
        t4, t8 = n_tok / pts[4][key], n_tok / pts[8][key]
        b = (t8 - t4) / 4.0; a0 = t4 - 4 * b
        ext[key] = round(n_tok / (a0 + 32 * b), 1)
    ext["value"] = round(640.0 / (512.0 / ext["pp512_tok_s"] + 128.0 / ext["tg128_tok_s"]), 1)
    out["extrapolated_32_layers"] = dict(ext, note="linear in the layer count from the 4- and 8-layer files; NOT a run of the 26 GB model")
    out["pp512_tok_s"] = ext["pp512_tok_s"]; out["tg128_tok_s"] = ext["tg128_tok_s"]; out["value"] = ext["value"]
    return out


def compact_line(out, log):
    """The driver keeps a tail of the line: the full record goes to gpurun_out/bench_details.json (and, as one line, to stderr), the printed line keeps every field the contract
    names plus the numbers the rooflines are judged on, within a few KB."""
    full = json.dumps(out)
    try:
        d = os.path.join(ROOT, "gpurun_out"); os.makedirs(d, exist_ok=True)
        open(os.path.join(d, "bench_details.json"), "w").write(full)
    except OSError:
        pass
    log("[bench details] " + full)
    if len(full) <= 6000:
        return out
    o = {k: out[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data") if k in out}
    cfg = dict(out.get("config") or {})
    for k in ("gpu_during_timed_region", "pp_ms_min_median", "tg_ms_min_median", "host_submit_ms_median", "weight_bytes_per_rank", "type_mix"):
        cfg.pop(k, None)
    o["config"] = cfg
    rf = dict(out.get("roofline") or {}); ts = rf.pop("traffic_source", None)
    if isinstance(ts, dict):
        rf["traffic_source"] = {k: ts[k] for k in ("method", "counter", "kernel") if k in ts}
    rp = out.get("roofline_prefill") or {}
    ko = rp.get("kernel_only") or {}; n4 = rp.get("n4096") or {}; k4 = n4.get("kernel_only") or {}
    rf["prefill"] = {"bound": "mfma", "peak": rp.get("peak"), "unit": rp.get("unit"), "kernel": rp.get("kernel"), "n512_op_frac": rp.get("frac"), "n512_kernel_frac": ko.get("frac"),
                     "n512_kernel_us": ko.get("avg_us"), "n4096_op_frac": n4.get("frac"), "n4096_kernel_frac": k4.get("frac"), "n4096_kernel_us": k4.get("avg_us"), "n4096_form": n4.get("kernel_form"),
                     "n4096_op_frac_per_wave_form": (n4.get("per_wave_form") or {}).get("frac"), "n4096_op_frac_shared_tile_form": (n4.get("shared_tile_form") or {}).get("frac"),
                     "pp512_pass_frac": rp.get("pp_pass_frac")}
    p4 = rp.get("pp4096_llama_bench")
    if p4:      # north_star's 4k-token prefill, end to end through llama-bench: default ubatch and one 4096-token ubatch
        rf["prefill"]["pp4096_pass_frac"] = {k: {"tok_s": v.get("pp4096_tok_s"), "frac": v.get("pass_frac")} for k, v in p4.items() if isinstance(v, dict)}
    hv = rf.get("harness_variant")
    if hv:
        rf["harness_variant"] = {k: hv[k] for k in ("avg_launch_us", "frac", "kernel_trace_avg_us") if k in hv}
    ktr = rf.get("kernel_trace")
    if isinstance(ktr, dict):
        rf["kernel_trace"] = {k: ktr[k] for k in ("avg_us", "frac", "dispatches", "averaged_over", "vs_hip_events", "agreement") if k in ktr}
    dt = (rf.get("decode_token") or {}).get("llama_bench_trace")
    if dt:      # keep the diagnosis short: sums + the five heaviest kernels
        rf["decode_token"] = dict(rf["decode_token"], llama_bench_trace={"kernel_sum_us": dt.get("kernel_sum_us"), "gap_sum_us": dt.get("gap_sum_us"), "kernels": dt.get("kernels"),
                                                                          "top5": [[t["kernel"][:44], t["calls"], t["avg_us"]] for t in dt.get("top", [])[:5]]})
    pt = rp.get("pp512_llama_bench_trace")
    if pt:
        rf["prefill"]["pp512_trace"] = {"kernel_sum_us": pt.get("kernel_sum_us"), "gap_sum_us": pt.get("gap_sum_us"), "top3": [[t["kernel"][:44], t["calls"], t["avg_us"]] for t in pt.get("top", [])[:3]]}
    gd = rp.get("gpu_during_timing")
    if gd:
        rf["prefill"]["gpu_during_timing"] = {k: gd[k] for k in list(gd)[:6]}
    o["roofline"] = rf
    cb = out.get("cpu_baseline")
    if cb:
        o["cpu_baseline"] = {k: cb[k] for k in ("value", "unit", "cores", "kind", "sample", "pp512_tok_s", "tg128_tok_s") if k in cb}
    mo = out.get("matmul_only")
    if mo:
        o["matmul_only"] = {k: v for k, v in mo.items() if k != "config"}
    lb = out.get("llama_bench")
    if lb:
        o["llama_bench"] = {k: lb[k] for k in ("value", "pp512_tok_s", "pp_stddev", "tg128_tok_s", "tg_stddev", "graphs", "fusions", "wall_s") if k in lb}
        # (the shim's GGML_CDNA4_STATS lines stay in the details record: the line is read from a bounded tail)
    cs = out.get("configs")
    if cs:
        o["configs"] = {k: ({"value": v.get("value"), "workload": (v.get("config") or {}).get("workload", "")[:90], "roofline_frac": (v.get("roofline") or {}).get("frac"),
                             "decode_token_frac": ((v.get("roofline") or {}).get("decode_token") or {}).get("frac"),
                             "prefill_kernel_frac": (((v.get("roofline_prefill") or {}).get("kernel_only")) or {}).get("frac"),
                             "llama_bench": {kk: (v["llama_bench"][kk] if kk != "skipped" else v["llama_bench"][kk][:60] + " ...") for kk in v.get("llama_bench") or {} if kk.endswith("_tok_s") or kk in ("value", "skipped")} if v.get("llama_bench") else None,
                             "cpu_baseline": {kk: v["cpu_baseline"][kk] for kk in ("value", "cores", "kind") if kk in (v.get("cpu_baseline") or {})} if v.get("cpu_baseline") else None}
                            if "error" not in v else v) for k, v in cs.items()}
    ev = out.get("env") or {}
    o["env"] = {k: ev[k] for k in ("hip_runtime", "rocm", "gpu", "cpu_quota", "loadavg") if k in ev}
    o["details"] = "gpurun_out/bench_details.json (and the '[bench details]' line on stderr): per-config records, PMC sources, env, clocks"
    return o


T_START = time.time()


def dry_run(args):
    """`bench.py --gpus N --dry-run` (CPU tier, tests/test_bench_dry_run.py): everything of an N-rank run that does not need a GPU -- the launch contract (RANK / WORLD_SIZE /
    MASTER_* from the environment, one process per rank), the tensor-parallel shard plan of the config (rows of q / k / v / up / gate and the K slices of wo / down per rank, with
    the divisibility the kernels need), the exchange step's shape (two all-reduces per layer: [1, n_embd] f32 per token, [ubatch, n_embd] bf16 per prompt ubatch) run once over
    gloo, the barrier + max-over-ranks timing, and the ONE JSON line of rank 0 with the keys of the contract -- so that the first real multi-GPU run can only fail on hardware facts."""
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1"))
    assert world == args.gpus, "--gpus must match WORLD_SIZE (launch N>1 with torch.distributed.run)"
    if world > 1:
        dist.init_process_group("gloo")
        assert dist.get_world_size() == world
    cfg = CONFIGS[args.config]; s = world if world > 1 else cfg["shard"]
    E, NF, NL = cfg["n_embd"], cfg["n_ff"], cfg["n_layer"]; KV = cfg["n_head_kv"] * cfg["head_dim"]; QD = cfg.get("n_head", E // cfg["head_dim"]) * cfg["head_dim"]
    assert cfg["n_head_kv"] % s == 0 and (NF // s) % 256 == 0 and (E // s) % 256 == 0, "config %s does not shard %d ways" % (args.config, s)
    ty = cfg["types"]; nexp = cfg["n_expert"] or 1

    def nbytes(t, rows, cols):
        return rows * (cols // BLCK_SIZE[t]) * TYPE_SIZE[t]
    per_rank = 0; shapes = {}
    for il in range(NL):
        for name, rows, cols, mult in (("wq", QD // s, E, 1), ("wk", KV // s, E, 1), ("wv", KV // s, E, 1), ("wo", E, QD // s, 1), ("up", NF // s, E, nexp), ("gate", NF // s, E, nexp), ("down", E, NF // s, nexp)):
            per_rank += mult * nbytes(ty(name, il, NL), rows, cols)
            if il == 0:
                shapes[name] = [rows, cols]
    out_bytes = nbytes(ty("output", 0, NL), cfg["n_vocab"], E); per_rank += out_bytes          # output.weight is replicated
    # the exchange step once, as the timed run would: token-size f32 and ubatch-size bf16 partial sums
    tok = torch.full((1, E), float(rank + 1)); ub = torch.full((min(cfg["n_prompt"], N_UBATCH), E), float(rank + 1), dtype=torch.bfloat16)
    t0 = time.perf_counter()
    if world > 1:
        dist.barrier(); dist.all_reduce(tok); dist.all_reduce(ub); dist.barrier()
    el = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    if world > 1:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
    want = world * (world + 1) / 2.0
    assert float(tok[0, 0]) == want and float(ub[0, 0]) == want, "all-reduce over %d ranks summed to %r" % (world, float(tok[0, 0]))
    tot = torch.tensor([float(per_rank - out_bytes)], dtype=torch.float64)
    if world > 1:
        dist.all_reduce(tot)
    if rank == 0:
        NPc = cfg["n_prompt"]
        out = {"metric": "llama-bench pp%d + tg128 tok/s, %s (DRY RUN: no GPU work)" % (NPc, cfg["name"]), "value": 0.0, "unit": "tok/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": round(float(el[0]) * 1e3, 3), "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "none (dry run)", "data": "none (dry run)",
               "config": {"workload": "%s, shard plan only" % cfg["name"], "parallelism": ("tp%d (row-split q/k/v/up/gate, K-split o/down, all-reduce x2 per layer)" % world) if world > 1 else "single GPU",
                          "dry_run": True, "per_rank_shapes_layer0": shapes, "per_rank_weight_bytes": per_rank, "sharded_weight_bytes_all_ranks": int(tot[0]), "replicated_output_bytes": out_bytes,
                          "reduces_per_pass": 2 * NL, "reduce_messages": {"token_f32_bytes": 4 * E, "ubatch_bf16_bytes": 2 * E * min(NPc, N_UBATCH)}},
               "roofline": None, "cpu_baseline": None}
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", default="c2", choices=sorted(CONFIGS.keys()))
    ap.add_argument("--no-extra-configs", action="store_true", help="N = 1, default config: do not append the short c3 / c4shard / c5 runs")
    ap.add_argument("--no-graph", action="store_true", help="do not capture the decode pass in a HIP graph")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-llama-bench", action="store_true", help="skip the end-to-end run of the reference llama-bench binary through the shim")
    ap.add_argument("--no-knob-probe", action="store_true", help=argparse.SUPPRESS)      # (round 3's probe of two opt-in instantiations: both are the defaults now; flag kept so old command lines parse)
    ap.add_argument("--no-pmc", action="store_true", help="do not spawn the rocprofv3 --pmc child that measures roofline.traffic")
    ap.add_argument("--pmc-all", action="store_true", help=argparse.SUPPRESS)      # (now the default; kept so that old command lines still parse)
    ap.add_argument("--no-pmc-extra", action="store_true", help="measure roofline.traffic for the headline config only (skip the rocprofv3 child of c3 / c4shard / c5)")
    ap.add_argument("--ab-lib", default=None, help="A/B: path of ANOTHER build of libggml-hip-cdna4.so (scripts/build_rev.sh <git-rev>); the headline kernels are timed with "
                                                   "both builds in this one process, interleaved, and reported under `ab`")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--cpu-op-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--dry-run", action="store_true", help="no GPU: launch contract, shard plan and exchange step over gloo, the JSON line with the contract's keys (CPU tier test)")
    ap.add_argument("--tp-shapes", type=int, default=0, help="debug: run ONE process with the per-rank shard shapes of an N-way tensor-parallel run (no collectives)")
    args = ap.parse_args()
    if args.dry_run:
        return dry_run(args)
    if args.pmc_child:
        return pmc_child(args)
    if args.cpu_op_child:
        print(json.dumps(cpu_op_level(lambda *a: print(*a, file=sys.stderr), CONFIGS["c2"])), flush=True)
        return

    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1")); local = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, "--gpus must match WORLD_SIZE (launch N>1 with torch.distributed.run)"
    # debug only (exercising the multi-rank control flow on a 1-GPU box): all ranks on one device, gloo instead of RCCL
    dbg_dev = os.environ.get("CDNA4_BENCH_DEBUG_ONE_DEVICE")
    if dbg_dev is not None:
        local = int(dbg_dev)
    if world > 1 and dbg_dev is None and torch.cuda.device_count() <= local:
        # fail loudly: a scaling run with fewer visible GPUs than ranks must not quietly time-share devices
        raise SystemExit("bench.py --gpus %d: rank %d (LOCAL_RANK %d) has no GPU of its own: %d visible" % (args.gpus, rank, local, torch.cuda.device_count()))
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)

    def log(*a):
        if rank == 0:
            print(*a, file=sys.stderr, flush=True)

    import torch.distributed as dist
    if world > 1:
        if dbg_dev is not None:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=device)

    if world > 1 and dbg_dev is None:
        # every rank joined the RCCL communicator on a DISTINCT physical GPU, or the run stops here (no silent fallback to fewer devices): world size as seen by the
        # collective library, and the PCI bus ids of all ranks
        assert dist.get_world_size() == args.gpus, "RCCL world size %d != --gpus %d" % (dist.get_world_size(), args.gpus)
        ids = [None] * world
        try:
            bus = torch.cuda.get_device_properties(local).pci_bus_id if hasattr(torch.cuda.get_device_properties(local), "pci_bus_id") else None
        except Exception:      # noqa: BLE001
            bus = None
        dist.all_gather_object(ids, (os.uname().nodename, bus if bus is not None else local))
        if len(set(ids)) != world:
            raise SystemExit("bench.py --gpus %d: ranks share a device (%s): refusing to report a scaling number" % (args.gpus, ids))
        probe = torch.ones(1, device=device); dist.all_reduce(probe)
        if int(probe.item()) != world:
            raise SystemExit("bench.py --gpus %d: an all-reduce over the communicator summed %d ranks" % (args.gpus, int(probe.item())))
    pkg = _load_package()
    be = pkg.Cdna4Backend(local)
    if world > 1:                   # bootstrap the C-ABI communicator: rank 0's unique id travels over torch.distributed
        idt = torch.zeros(128, dtype=torch.uint8, device=device)
        if rank == 0:
            idt.copy_(torch.frombuffer(bytearray(be.comm_unique_id()), dtype=torch.uint8))
        dist.broadcast(idt, 0)
        try:
            be.comm_init(bytes(idt.cpu().numpy().tobytes()), rank, world)
        except Exception as e:      # keep the scaling run alive: same collective through torch.distributed (also RCCL)
            log("C-ABI communicator failed (%r): reducing through torch.distributed instead" % (e,))
            be.reduce = lambda buf: (dist.all_reduce(buf), buf)[1]
        # Partial sums of up to 8 MiB on the wire go through the one-shot all-reduce over IPC-mapped windows (cdna4_window_*: one launch per reduce, capturable,
        # bf16 conversion of the prompt-size messages inside the launch), the collective library carries anything larger.  The 64-byte handles travel over
        # torch.distributed; the windows are only kept if a reduce through them reproduces the collective library's result (CDNA4_BENCH_REDUCE=rccl: never).
        if os.environ.get("CDNA4_BENCH_REDUCE", "window") == "window":
            from ik_llama_cpp_amd import tp
            if tp.setup_ipc_windows(be, dist, rank, world, device, log):
                if getattr(be.reduce, "__self__", None) is not be:      # (communicator unavailable: torch.distributed carries what the windows do not)
                    big = be.reduce

                    def reduce(buf, wire=None):
                        nb = buf.numel() * (buf.element_size() if wire is None else 2)
                        return be.window_reduce(buf, wire=wire) if nb <= be.window_bytes and nb % 16 == 0 and buf.data_ptr() % 16 == 0 else big(buf)
                    be.reduce = reduce
                log("reduces up to 8 MiB: one-shot over IPC windows")

    if args.tp_shapes and world == 1:
        CONFIGS[args.config] = dict(CONFIGS[args.config], shard=args.tp_shapes)
    if world > 1:
        CONFIGS[args.config] = dict(CONFIGS[args.config], shard=1)         # (c4shard with --gpus 8: the real TP run)

    env = env_info() if rank == 0 else None
    gpu_start = gpu_sample(local) if rank == 0 else None
    res = run_config(args, args.config, be, rank, world, device, log, args.steps, args.warmup, full=True)
    extra = {}
    if world == 1 and args.config == "c2" and not args.no_extra_configs and not args.tp_shapes:
        for key in ("c1", "c3", "c4shard", "c5"):
            try:
                st = 5
                r = run_config(args, key, be, rank, world, device, log, st, 1, full=False)
                extra[key] = {"value": r["value"], "unit": r["unit"], "steps": st, "warmup": 1, "config": r["config"], "roofline": r["roofline"],
                              "roofline_prefill": r["roofline_prefill"]}
            except Exception as e:
                log("extra config %s failed: %r" % (key, e)); extra[key] = {"error": repr(e)[:300]}
                torch.cuda.empty_cache()
    ab = None
    if world == 1 and args.ab_lib:
        try:
            ab = ab_compare(args, pkg, be, device, log)
        except Exception as e:
            log("--ab-lib comparison failed: %r" % (e,)); ab = {"error": repr(e)[:300]}

    if rank == 0:
        cfgname = CONFIGS[args.config]["name"]; NPc = CONFIGS[args.config]["n_prompt"]
        # the mat-mul harness of run_config (every MUL_MAT / FUSED_UP_GATE of the graph through the C ABI, no attention / norm / rope): what the rooflines are measured on
        matmul_only = {"value": res["value"], "unit": "tok/s", "ms_per_step": res["ms_per_step"], "steps": args.steps, "warmup": args.warmup,
                       "what": "pp%d + tg128 over the quantized mat-mul path alone (ctypes harness, decode pass replayed from one HIP graph)" % NPc,
                       "pp%d_tok_s" % NPc: res["config"].get("pp%d_tok_s" % NPc), "tg128_tok_s": res["config"].get("tg128_tok_s"), "config": res["config"]}
        out = {
            "metric": "llama-bench pp%d + tg128 tok/s, %s (quantized mat-mul path only)" % (NPc, cfgname),
            "value": res["value"], "unit": "tok/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": res["ms_per_step"], "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "u8 weights x i8 activations -> i32 block sums, f32 scale accumulate (decode); f16 MFMA, f32 accumulate (prefill)",
            "data": "synthetic (random-bit quant blocks in the config's type mix, N(0,1) activations)",
            "config": res["config"], "roofline": res["roofline"], "roofline_prefill": res["roofline_prefill"], "cpu_baseline": res["cpu_baseline"],
        }
        if extra:
            out["configs"] = extra
        if ab:
            out["ab"] = ab
    be.close()
    if rank == 0:
        if world == 1 and args.config == "c2" and not args.no_llama_bench and not args.tp_shapes:
            torch.cuda.empty_cache()
            # THE METRIC BASELINE.json NAMES: the reference's own llama-bench (unmodified sources) on a full-size synthetic Llama-3-8B Q4_K_M GGUF, every node of the graph on the
            # device through the backend shim.  One "step" = one llama-bench repetition (pp512 pass + 128 decoded tokens, each timed by llama-bench itself between device
            # synchronizations, examples/llama-bench/llama-bench.cpp:2096-2131); llama-bench runs its own warm-up pass of each test first.
            lb = llama_bench_end_to_end(log, reps=max(args.steps, 1))
            out["llama_bench"] = lb
            if lb and lb.get("value"):
                out["matmul_only"] = matmul_only
                out["metric"] = "llama-bench pp%d + tg128 tok/s, %s, 1xMI355X" % (NPc, cfgname)
                out["value"] = lb["value"]; out["ms_per_step"] = round(1e3 * (NPc + 128) / lb["value"], 3)
                out["config"] = {"workload": "%s synthetic GGUF (%s), the reference's llama-bench -p %d -n 128 -ngl 99 -fa 1 -r %d through libggml-cuda-cdna4.so: every graph node "
                                             "(mat-muls, norms, rope, KV writes, attention) on the device, decode steps replayed from HIP graphs" % (cfgname, lb.get("model", "?"), NPc, max(args.steps, 1)),
                                 "parallelism": "single GPU", "pp%d_tok_s" % NPc: lb.get("pp%d_tok_s" % NPc), "pp_stddev": lb.get("pp_stddev"), "tg128_tok_s": lb.get("tg128_tok_s"),
                                 "tg_stddev": lb.get("tg_stddev"), "graphs": lb.get("graphs"), "timed_by": "llama-bench (steps = its -r repetitions; its own warm-up run precedes them)",
                                 "rooflines_measured_on": "the mat-mul harness (`matmul_only`): same kernels, same weights shapes, C ABI"}
            else:
                out["metric"] += " -- llama-bench leg unavailable in this run"
            if lb and lb.get("value") and not args.no_pmc:
                # which kernels a decoded token / a prompt pass of the timed model spends its time in, and how much of the token is idle gaps: kernel traces of the same binary
                tr = llama_bench_trace(log, 0, 48)
                if tr:
                    out["roofline"].setdefault("decode_token", {})["llama_bench_trace"] = tr
                tr = llama_bench_trace(log, 512, 0)
                if tr:
                    out["roofline_prefill"]["pp512_llama_bench_trace"] = tr
            if lb and lb.get("value"):
                # north_star quotes the prefill target on a 4k-token prompt: the same binary with -p 4096 -n 0 at llama-bench's default ubatch (512: eight ubatches, the GEMMs
                # of the headline run at growing attention depth) and as ONE ubatch (-ub 4096 -b 4096: every GEMM at N = 4096, the large-batch route).  pass_frac = tok/s x the
                # mat-mul flops of a token (2 x 6.98 G weights of the 7 x 32 layer matrices = 13.96 GFLOP; attention flops NOT counted) / 2.5 PFLOP/s
                cfg2 = CONFIGS["c2"]; E_, NF_ = cfg2["n_embd"], cfg2["n_ff"]
                fl_tok = 2.0 * cfg2["n_layer"] * (2 * E_ * E_ + 2 * cfg2["n_head_kv"] * cfg2["head_dim"] * E_ + 3 * NF_ * E_)
                pp4k = {}
                for name, xa in (("ub512", []), ("ub4096", ["-ub", "4096", "-b", "4096"])):
                    r4 = run_llama_bench(log, synth_gguf("llama3-8b-q4km", log), 4096, 0, 3, gpu=True, extra_args=xa, timeout=300)
                    if r4 and r4.get("pp4096_tok_s"):
                        pp4k[name] = {"pp4096_tok_s": r4["pp4096_tok_s"], "pp_stddev": r4.get("pp_stddev"), "pass_frac": round(r4["pp4096_tok_s"] * fl_tok / 1e12 / MFMA_F16_PEAK_TFLOPS, 4), "cmd": r4["cmd"]}
                out["roofline_prefill"]["pp4096_llama_bench"] = dict(pp4k, gflop_per_token=round(fl_tok / 1e9, 2),
                                                                     what="whole 4096-token prompt pass end to end (mat-muls, norms, rope, attention, KV writes) against the MFMA peak, mat-mul flops only")
            if "c3" in extra and "error" not in extra["c3"]:
                # BASELINE configs[2] end to end: the same llama-bench on a synthetic 8B GGUF in the IQ2_S / IQ3_S / Q6_K mix of CONFIGS["c3"], and the reference CPU backend on it
                extra["c3"]["llama_bench"] = llama_bench_end_to_end(log, 512, 128, 5, gguf_kind="llama3-8b-iq2m")
                if not args.no_cpu_baseline and time.time() - T_START < 330:
                    extra["c3"]["cpu_baseline"] = cpu_baseline(log, CONFIGS["c3"], "llama3-8b-iq2m", 512, 128, op_level=False, reps=1)
            if "c4shard" in extra and "error" not in extra["c4shard"]:
                extra["c4shard"]["llama_bench"] = {"skipped": "no single-process GGUF expresses ONE rank's shard of a TP = 8 run: the reference's multi-GPU mode (-sm graph) is one process driving "
                                                              "all eight devices, and on a 1-GPU box its eight sub-graphs run back to back on the one GPU (tests/test_gpu_llama.py does that on 2 / 4 / 8 logical "
                                                              "devices for correctness).  The per-rank figure is the mat-mul harness above (shapes of one rank, no collectives); the end-to-end leg needs --gpus 8."}
            if "c5" in extra and "error" not in extra["c5"]:
                # BASELINE configs[4] (single-GPU part): the 32-layer Mixtral-8x7B-shaped file, measured (26 GB; skipped with a note when the run is already long)
                try:
                    if time.time() - T_START < 300:
                        extra["c5"]["llama_bench"], c5cpu = llama_bench_mixtral(log, 3, cpu=not args.no_cpu_baseline)
                        if c5cpu:
                            extra["c5"]["cpu_baseline"] = c5cpu
                    else:
                        extra["c5"]["llama_bench"] = {"skipped": "run already %.0f s long" % (time.time() - T_START)}
                except Exception as e:      # noqa: BLE001
                    log("c5 llama-bench leg failed: %r" % (e,)); extra["c5"]["llama_bench"] = None
            if "c1" in extra and "error" not in extra["c1"]:
                # BASELINE configs[0] is the reference's own CPU case: the reference llama-bench on a Qwen3-0.6B-shaped IQ4_NL GGUF, CPU backend, pp128 / tg32 --
                # and the same file through the shim on the GPU
                extra["c1"]["llama_bench"] = llama_bench_end_to_end(log, 128, 32, 5, gguf_kind="qwen3-0.6b-iq4nl")
                if not args.no_cpu_baseline:
                    extra["c1"]["cpu_baseline"] = cpu_baseline(log, CONFIGS["c1"], "qwen3-0.6b-iq4nl", 128, 32, op_level=False)
        if world > 1 and args.config == "c2" and not args.no_llama_bench and dbg_dev is None:
            # the same metric at N > 1: the reference's llama-bench with -sm graph over the N devices (its multi-GPU mode is ONE process driving every GPU: sub-graphs per device,
            # GGML_OP_REDUCE across them -- peer reads over xGMI through the shim); the one-process-per-GPU RCCL / IPC-window run above stays in the line as `matmul_only`
            out["matmul_only"] = matmul_only
            torch.cuda.empty_cache()
            lb = llama_bench_end_to_end(log, reps=max(args.steps, 1), extra_args=["-sm", "graph"])
            out["llama_bench"] = lb
            if lb and lb.get("value"):
                out["metric"] = "llama-bench pp%d + tg128 tok/s, %s, %dxMI355X (-sm graph)" % (NPc, cfgname, world)
                out["value"] = lb["value"]; out["ms_per_step"] = round(1e3 * (NPc + 128) / lb["value"], 3)
                out["config"] = {"workload": "%s synthetic GGUF, the reference's llama-bench -p %d -n 128 -ngl 99 -fa 1 -sm graph -r %d through libggml-cuda-cdna4.so on %d GPUs (one process, "
                                             "per-device sub-graphs, GGML_OP_REDUCE by peer access)" % (cfgname, NPc, max(args.steps, 1), world),
                                 "parallelism": "tp%d (-sm graph)" % world, "pp%d_tok_s" % NPc: lb.get("pp%d_tok_s" % NPc), "tg128_tok_s": lb.get("tg128_tok_s"),
                                 "matmul_only_is": "one process per GPU, row / K split, RCCL or IPC-window all-reduce x2 per layer (torch.distributed launch)"}
            else:
                out["metric"] += " -- llama-bench -sm graph leg unavailable: the one-process-per-GPU mat-mul harness is the value"
        out["env"] = env
        out["env"]["gpu_at_start"] = gpu_start; out["env"]["gpu_at_end"] = gpu_sample(local)
        print(json.dumps(compact_line(out, log)), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
