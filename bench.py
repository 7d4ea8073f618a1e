#!/usr/bin/env python3
"""bench.py -- the BASELINE.json metric on the hot path: llama-bench pp512 + tg128 for Llama-3-8B Q4_K_M, restricted to
the quantized mat-mul path this repo implements (every MUL_MAT / FUSED_UP_GATE of the model graph, nothing else).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A "step" = one llama-bench repetition of the mat-mul path: one 512-token prompt pass (pp512: every weight matrix applied to a
512-column activation batch, output.weight to the last token only, like llama-bench) followed by 128 single-token passes
(tg128).  Weights: random-bit Q4_K / Q6_K blocks with finite scales, in the Q4_K_M type mix of src/llama-quantize.cpp
(attn_v / ffn_down -> Q6_K on the 16 "use_more_bits" layers, output.weight -> Q6_K), all 4.6 GB resident in HBM before the
timed region; activations: synthetic N(0,1) f32.  value = 640 tokens / step time (whole job, all ranks).

N > 1: tensor parallel exactly like the reference's `-sm graph`: q/k/v/up/gate row-split, o/down K-split, one
all-reduce(sum) of the [4096 x tokens] f32 partials after o and after down (RCCL over xGMI through the C ABI), output.weight
replicated.  Total work is fixed => "scaling": "strong".

Extra objects on the JSON line: `roofline` (dominant kernel = the fused up*gate Q4_K decode GEMV, HIP-event timed live over
the 32 layers' distinct weights) and `cpu_baseline` (the REAL reference CPU kernels from oracle/_ref driven by OpenMP on this
host, bounded sample; rank 0, N=1 only)."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
os.environ.setdefault("OMP_PROC_BIND", "close")
os.environ.setdefault("OMP_PLACES", "cores")

import numpy as np   # noqa: E402
import torch         # noqa: E402

from __graft_entry__ import _load_package   # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E peak (MI355X_MICROARCH.md "Chip-level parameters"); the ONE place this constant lives
MFMA_F16_PEAK_TFLOPS = 2500.0  # dense f16/bf16 MFMA peak

Q4_K, Q6_K, Q8_2_X4 = 12, 14, 99           # enum ggml_type of this fork (ggml.h)
TYPE_SIZE = {Q4_K: 144, Q6_K: 210}
D_OFFS = {Q4_K: (0, 2), Q6_K: (208,)}

# Llama-3-8B (n_embd 4096, n_ff 14336, 32 heads / 8 KV heads x 128, 32 layers, vocab 128256)
N_EMBD, N_FF, N_HEAD_KV, HEAD_DIM, N_LAYER, N_VOCAB = 4096, 14336, 8, 128, 32, 128256
N_PROMPT, N_GEN = 512, 128


def use_more_bits(i, n):       # src/llama-quantize.cpp:312-314
    return i < n // 8 or i >= 7 * n // 8 or (i - n // 8) % 3 == 2


def synth_weights(t, m, k, gen, device):
    """random-bit blocks with finite fp16 super-block scales (any byte pattern is a valid block)."""
    ts = TYPE_SIZE[t]; nb = k // 256
    w = torch.randint(0, 256, (m, nb, ts), dtype=torch.uint8, device=device, generator=gen)
    for off in D_OFFS[t]:
        d = (torch.rand((m, nb), device=device, generator=gen) * 0.02 + 1e-3).to(torch.float16)
        if off != 2:
            d = d * (torch.randint(0, 2, (m, nb), device=device, generator=gen) * 2 - 1).to(torch.float16)
        w[:, :, off:off + 2] = d.view(torch.uint8).view(m, nb, 2)
    return w.view(m, nb * ts)


class Model:
    """The mat-mul path of one Llama-3-8B Q4_K_M forward, sharded for tensor parallel rank `rank` of `world`."""

    def __init__(self, be, rank, world, device):
        self.be, self.rank, self.world, self.dev = be, rank, world, device
        self.shard = world                      # shapes follow `shard`; collectives follow `world`
        self.emit_q8 = os.environ.get("CDNA4_BENCH_EMIT_Q8", "0") == "1"   # measured neutral at N = 1, harmful on TP shards (profiles/r01_notes.md)
        gen = torch.Generator(device=device); gen.manual_seed(1234 + rank)
        s = world
        assert N_HEAD_KV % s == 0 and (N_FF // s) % 256 == 0 and (N_EMBD // s) % 256 == 0
        self.layers = []
        for il in range(N_LAYER):
            tv = Q6_K if use_more_bits(il, N_LAYER) else Q4_K           # attn_v and ffn_down (llama-quantize.cpp:631-632,731-737)
            L = dict(
                wq=(Q4_K, synth_weights(Q4_K, N_EMBD // s, N_EMBD, gen, device)),
                wk=(Q4_K, synth_weights(Q4_K, N_HEAD_KV * HEAD_DIM // s, N_EMBD, gen, device)),
                wv=(tv, synth_weights(tv, N_HEAD_KV * HEAD_DIM // s, N_EMBD, gen, device)),
                wo=(Q4_K, synth_weights(Q4_K, N_EMBD, N_EMBD // s, gen, device)),          # K-split
                up=(Q4_K, synth_weights(Q4_K, N_FF // s, N_EMBD, gen, device)),
                gate=(Q4_K, synth_weights(Q4_K, N_FF // s, N_EMBD, gen, device)),
                down=(tv, synth_weights(tv, N_EMBD, N_FF // s, gen, device)),              # K-split
            )
            self.layers.append(L)
        self.output = (Q6_K, synth_weights(Q6_K, N_VOCAB, N_EMBD, gen, device))           # replicated (llama-build-context.cpp:2499-2530)
        self.bufs = {}

    def weight_bytes(self):
        n = self.output[1].numel()
        for L in self.layers:
            n += sum(v[1].numel() for v in L.values())
        return n

    def _buf(self, name, n, m):
        key = (name, n)
        if key not in self.bufs:
            self.bufs[key] = torch.empty((n, m), dtype=torch.float32, device=self.dev)
        return self.bufs[key]

    def prepare(self, n):
        """activations for a batch of n columns (synthetic, fixed) + output buffers; nothing is allocated in the timed region."""
        g = torch.Generator(device=self.dev); g.manual_seed(99 + n)
        s = self.shard
        self.bufs[("x", n)] = torch.randn((n, N_EMBD), device=self.dev, generator=g)           # layer input (after norm)
        self.bufs[("attn", n)] = torch.randn((n, N_EMBD // s), device=self.dev, generator=g)    # attention output slice (wo input)
        self.bufs[("x1", n)] = torch.randn((1, N_EMBD), device=self.dev, generator=g)
        for name, m in (("q", N_EMBD // s), ("k", N_HEAD_KV * HEAD_DIM // s), ("v", N_HEAD_KV * HEAD_DIM // s), ("o", N_EMBD),
                        ("ffn", N_FF // s), ("down", N_EMBD)):
            self._buf(name, n, m)
        self._buf("logits", 1, N_VOCAB)
        if n == 1:          # decode: the fused up*gate launch also emits ffn_down's int8 input (cdna4_fused_up_gate_q8)
            self.bufs[("ffn_q8", 1)] = torch.empty((1, (N_FF // s) // 128 * 144), dtype=torch.uint8, device=self.dev)
        if n > 32 and self.world > 1:                # prompt-size partial sums travel as bf16 (reduce_type, llama-build-context.cpp:1198-1200)
            self.bufs[("red16", n)] = torch.empty((n, N_EMBD), dtype=torch.bfloat16, device=self.dev)
        self.be.reserve_workspace(512 * N_FF * 2 + (1 << 20))

    def reduce(self, t, n):
        """GGML_OP_REDUCE of a [n, n_embd] partial sum: f32 for n <= 32, bf16 on the wire for prompt batches like the reference
        (src/llama.cpp:8147,8227-8242: reduce_type)."""
        r16 = self.bufs.get(("red16", n))
        if r16 is None or os.environ.get("CDNA4_BENCH_REDUCE_F32") == "1":
            self.be.reduce(t)
        else:
            r16.copy_(t); self.be.reduce(r16); t.copy_(r16)

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
            if n == 1 and self.emit_q8:
                f, fq = be.fused_up_gate_q8(L["up"][0], L["up"][1], L["gate"][1], x, out=self.bufs[("ffn", n)], q8_out=self.bufs[("ffn_q8", 1)])
                d = be.mul_mat(L["down"][0], L["down"][1], fq, out=self.bufs[("down", n)], x_type=Q8_2_X4)
            else:
                f = be.fused_up_gate(L["up"][0], L["up"][1], L["gate"][1], x, out=self.bufs[("ffn", n)])
                d = be.mul_mat(L["down"][0], L["down"][1], f, out=self.bufs[("down", n)])
            if self.world > 1:
                self.reduce(d, n)                                               # GGML_OP_REDUCE after ffn-down
        xl = self.bufs[("x1", n)] if last_only_logits or n == 1 else x
        be.mul_mat(self.output[0], self.output[1], xl, out=self.bufs[("logits", 1)])


def cpu_baseline(log):
    """The reference CPU path (oracle/_ref, real iqk_mul_mat kernels incl. its N>=32 repack path) on this host: bounded sample =
    tg over 4 distinct layers + pp512 over 1 layer, extrapolated to the 32-layer model (+ output.weight) like llama-bench would run it."""
    try:
        from oracle import bindings as ob
        if ob.ref_path() is None:
            return None
        ref = ob.Ref(); orc = ob.Oracle()
    except Exception as e:      # no reference library on this host
        log("cpu_baseline unavailable: %r" % (e,)); return None
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from common import random_block_bytes
    phys = os.cpu_count() or 1
    nth = max(1, min(64, phys // 2 if phys > 16 else phys))
    shapes = [("wq", Q4_K, N_EMBD, N_EMBD), ("wk", Q4_K, 1024, N_EMBD), ("wv", Q6_K, 1024, N_EMBD), ("wo", Q4_K, N_EMBD, N_EMBD),
              ("up", Q4_K, N_FF, N_EMBD), ("gate", Q4_K, N_FF, N_EMBD), ("down", Q6_K, N_EMBD, N_FF)]
    n_tg_layers = 8                        # 1.16 GB of distinct weights per sweep: well beyond the host's L3, like a real token
    layers = [[(t, random_block_bytes(t, m, k, 7 * li + i), m, k) for i, (_, t, m, k) in enumerate(shapes)] for li in range(n_tg_layers)]
    rng = np.random.default_rng(0)

    def run(n, layer_list, reps, stat):
        times = []
        acts = {}
        for k in (N_EMBD, N_FF):
            x = rng.standard_normal((n, k)).astype(np.float32)
            acts[k] = {vdt: ref.quantize_activations(vdt, x) for vdt in (ob.Q8_2_X4,)}
        outs = {m: np.zeros((n, m), np.float32) for m in (N_EMBD, 1024, N_FF)}
        for _ in range(reps):
            t0 = time.perf_counter()
            for L in layer_list:
                for (t, w, m, k) in L:
                    vdt = ob.vec_dot_type(t)
                    ref.mul_mat_omp(orc, t, w, acts[k][vdt], vdt, n, k, outs[m], nth)
            times.append(time.perf_counter() - t0)
        return stat(times) / len(layer_list)     # seconds per layer
    run(1, layers, 2, min)                                        # warm the thread team
    t_tg_layer = run(1, layers, 15, lambda v: float(np.median(v)))
    run(N_PROMPT, layers[:2], 1, min)                             # first touch of the work buffers
    t_pp_layer = run(N_PROMPT, layers[:2], 4, min)
    wout = random_block_bytes(Q6_K, N_VOCAB, N_EMBD, 99)
    xq = ref.quantize_activations(ob.Q8_2_X4, rng.standard_normal((1, N_EMBD)).astype(np.float32)); lo = np.zeros((1, N_VOCAB), np.float32)
    t_out = 1e30
    for _ in range(3):
        t0 = time.perf_counter(); ref.mul_mat_omp(orc, Q6_K, wout, xq, ob.Q8_2_X4, 1, N_EMBD, lo, nth); t_out = min(t_out, time.perf_counter() - t0)
    t_tg = N_LAYER * t_tg_layer + t_out
    t_pp = N_LAYER * t_pp_layer + t_out
    total = t_pp + N_GEN * t_tg
    return {"value": round((N_PROMPT + N_GEN) / total, 2), "unit": "tok/s", "cores": nth, "kind": "reference",
            "pp512_tok_s": round(N_PROMPT / t_pp, 1), "tg128_tok_s": round(1.0 / t_tg, 2),
            "sample": "reference iqk_mul_mat (oracle/_ref %s build) on the same mat-mul sequence: tg timed over %d distinct layers "
                      "(median of 15 sweeps) and pp512 over 2 layers (best of 4 after a warm-up) + output.weight, extrapolated x%d layers; %d OpenMP threads"
                      % (ref.variant, n_tg_layers, N_LAYER, nth)}


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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--no-graph", action="store_true", help="do not capture the decode pass in a HIP graph")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--tp-shapes", type=int, default=0, help="debug: run ONE process with the per-rank shard shapes of an N-way tensor-parallel run (no collectives)")
    ap.add_argument("--roofline-only", action="store_true", help="skip the timed steps; only the per-kernel roofline sweeps (for rocprofv3 --pmc passes)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1")); local = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, "--gpus must match WORLD_SIZE (launch N>1 with torch.distributed.run)"
    # debug only (exercising the multi-rank control flow on a 1-GPU box): all ranks on one device, gloo instead of RCCL
    dbg_dev = os.environ.get("CDNA4_BENCH_DEBUG_ONE_DEVICE")
    if dbg_dev is not None:
        local = int(dbg_dev)
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

    shard_world = args.tp_shapes if (args.tp_shapes and world == 1) else world
    model = Model(be, rank, shard_world, device)
    if shard_world != world:
        model.world = 1            # shapes of a shard, no reduce
    model.prepare(N_PROMPT); model.prepare(1)
    log("weights resident: %.3f GB on rank 0 (%s)" % (model.weight_bytes() / 1e9, be.description()))

    # ---- decode pass captured in a HIP graph (~130 launches / token)
    graph = None
    model.forward(1, False); torch.cuda.synchronize()
    # N > 1: eager launches by default.  With 64 all-reduces per token the host (193 calls, ~8.5 us each) is about as fast as the
    # devices (measured with --tp-shapes: eager 911-926 tok/s per rank without collectives), and a capture that fails inside a
    # collective can leave the process unusable.  CDNA4_BENCH_TP_GRAPH=1 captures the all-reduces with the kernels (RCCL supports stream
    # capture; thread-local capture mode so that RCCL's helper threads cannot invalidate it).
    # (gloo's CUDA path joins its own streams into a capture and cannot be captured: the debug mode runs eagerly)
    if not args.no_graph and (world == 1 or (os.environ.get("CDNA4_BENCH_TP_GRAPH", "0") == "1" and dbg_dev is None)):
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
    if world > 1:           # every rank must take the same path (a graph on some ranks and eager on others would still match collectives,
        flag = torch.tensor([1 if graph is not None else 0], device=device)     # but keep the runs comparable)
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

    def decode_token():
        if graph is not None:
            graph.replay()
        elif plan is not None:
            plan.replay(be._check)
        else:
            model.forward(1, False)

    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]

    def step():
        ev[0].record()
        model.forward(N_PROMPT, True)          # pp512
        ev[1].record()
        for _ in range(N_GEN):                 # tg128
            decode_token()
        ev[2].record()

    def sync_all():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(0 if args.roofline_only else args.warmup):
        step()
    sync_all()
    t0 = time.perf_counter()
    pp_ms = tg_ms = 1e-9
    for _ in range(0 if args.roofline_only else args.steps):
        step()
        torch.cuda.synchronize()
        pp_ms += ev[0].elapsed_time(ev[1]); tg_ms += ev[1].elapsed_time(ev[2])
    sync_all()
    elapsed = time.perf_counter() - t0
    tt = torch.tensor([elapsed, pp_ms, tg_ms], dtype=torch.float64, device=device)
    if world > 1:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    elapsed, pp_ms, tg_ms = [float(v) for v in tt.cpu()]
    ms_per_step = elapsed * 1e3 / args.steps
    tokens = N_PROMPT + N_GEN

    # ---- N > 1: the exchange step on its own ("reduce time broken out"): the 2 x n_layer GGML_OP_REDUCEs of one token
    # ([1, n_embd] f32) and of one 512-token batch (bf16 on the wire), back to back, max over ranks; outside the timed region
    reduce_info = None
    if world > 1:
        o1 = model.bufs[("o", 1)]; opp = model.bufs[("o", N_PROMPT)]
        o1.zero_(); opp.zero_()                      # (repeated in-place sums of zeros stay finite)
        er0, er1, er2 = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        for _ in range(4):
            model.reduce(o1, 1); model.reduce(opp, N_PROMPT)
        sync_all()
        er0.record()
        for _ in range(2 * N_LAYER):
            model.reduce(o1, 1)
        er1.record()
        for _ in range(2 * N_LAYER):
            model.reduce(opp, N_PROMPT)
        er2.record(); sync_all()
        rt = torch.tensor([er0.elapsed_time(er1), er1.elapsed_time(er2)], dtype=torch.float64, device=device)
        dist.all_reduce(rt, op=dist.ReduceOp.MAX)
        r_tok, r_pp = [float(v) for v in rt.cpu()]
        reduce_info = {"per_token_ms": round(r_tok, 4), "per_pp512_ms": round(r_pp, 4), "reduces_per_pass": 2 * N_LAYER,
                       "share_of_tg_time": round(r_tok / max(tg_ms / (args.steps * N_GEN), 1e-9), 4),
                       "share_of_pp_time": round(r_pp / max(pp_ms / args.steps, 1e-9), 4),
                       "wire": "f32 [1, %d] per token; bf16 [%d, %d] per prompt batch" % (N_EMBD, N_PROMPT, N_EMBD)}

    # ---- roofline of the dominant kernel: fused up*gate Q4_K decode GEMV (46 % of the decode weight bytes), timed with
    # HIP events on the launch stream over the 32 layers' DISTINCT weights (cold L2 / Infinity Cache: 2.1 GB per sweep).
    x1 = model.bufs[("x", 1)]; ffn = model.bufs[("ffn", 1)]
    def sweep():
        for L in model.layers:
            be.fused_up_gate(Q4_K, L["up"][1], L["gate"][1], x1, out=ffn)
    sweep(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    nsweep = 5
    e0.record()
    for _ in range(nsweep):
        sweep()
    e1.record(); torch.cuda.synchronize()
    k_ms = e0.elapsed_time(e1) / (nsweep * N_LAYER)
    m_loc = N_FF // shard_world
    alg_bytes = 2 * m_loc * (N_EMBD // 256) * 144 + 4 * N_EMBD + 4 * m_loc          # M*K*bpw/8 (x2 matrices) + 4*K*N + 4*M*N
    ach = alg_bytes / (k_ms * 1e-3) / 1e9
    # HBM bytes per launch from the committed rocprofv3 --pmc FETCH_SIZE pass of this same sweep (profiles/r01_pmc_fetch_size.json:
    # counters cannot be read from inside the process; gfx950 FETCH_SIZE x2 correction applied there), N=1 shape only
    traffic = None
    try:
        if world == 1:
            pm = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_fetch_size.json")))["kernels"]
            traffic = [v["hbm_read_bytes_per_launch_corrected"] for kname, v in pm.items() if "gemv_kernel<12, 1, true" in kname][0]
    except Exception:
        traffic = None
    roofline = {"bound": "hbm", "kernel": "gemv_kernel<Q4_K,1,fused up*gate> %dx%d x2" % (m_loc, N_EMBD), "achieved": round(ach, 1),
                "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": traffic,
                "bytes_per_launch": alg_bytes, "avg_launch_us": round(k_ms * 1e3, 2)}
    # secondary: the prefill MFMA kernel on the same weights (N=512)
    xp = model.bufs[("x", N_PROMPT)]; ffp = model.bufs[("ffn", N_PROMPT)]
    be.fused_up_gate(Q4_K, model.layers[0]["up"][1], model.layers[0]["gate"][1], xp, out=ffp); torch.cuda.synchronize()
    e0.record()
    for L in model.layers:
        be.fused_up_gate(Q4_K, L["up"][1], L["gate"][1], xp, out=ffp)
    e1.record(); torch.cuda.synchronize()
    g_ms = e0.elapsed_time(e1) / N_LAYER
    fl = 2.0 * 2 * m_loc * N_EMBD * N_PROMPT
    roofline_prefill = {"bound": "mfma", "kernel": "gemm_mfma_kernel<Q4_K,fused up*gate> N=512", "achieved": round(fl / (g_ms * 1e-3) / 1e12, 1),
                        "peak": MFMA_F16_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(fl / (g_ms * 1e-3) / 1e12 / MFMA_F16_PEAK_TFLOPS, 4),
                        "avg_launch_us": round(g_ms * 1e3, 1)}

    # BASELINE.json states the MFMA target on a 4k-token prefill: same fused launch at N = 4096 (one ubatch of pp4096), 4 layers' weights
    try:
        n4k = 4096
        g4 = torch.Generator(device=device); g4.manual_seed(7)
        x4 = torch.randn((n4k, N_EMBD), device=device, generator=g4); f4 = torch.empty((n4k, m_loc), device=device)
        be.reserve_workspace(n4k * N_EMBD * 2 + (1 << 20))
        be.fused_up_gate(Q4_K, model.layers[0]["up"][1], model.layers[0]["gate"][1], x4, out=f4); torch.cuda.synchronize()
        e0.record()
        for L in model.layers[:4]:
            be.fused_up_gate(Q4_K, L["up"][1], L["gate"][1], x4, out=f4)
        e1.record(); torch.cuda.synchronize()
        g4_ms = e0.elapsed_time(e1) / 4
        fl4 = 2.0 * 2 * m_loc * N_EMBD * n4k
        roofline_prefill["n4096"] = {"achieved": round(fl4 / (g4_ms * 1e-3) / 1e12, 1), "frac": round(fl4 / (g4_ms * 1e-3) / 1e12 / MFMA_F16_PEAK_TFLOPS, 4),
                                     "avg_launch_us": round(g4_ms * 1e3, 1)}
        del x4, f4
    except Exception as e:      # (memory-constrained shard configurations): the N = 512 figure above stands alone
        log("4k-token prefill roofline skipped: %r" % (e,))

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(log)

    if rank == 0:
        out = {
            "metric": "llama-bench pp512 + tg128 tok/s, Llama-3-8B Q4_K_M (quantized mat-mul path only)",
            "value": round(tokens * args.steps / elapsed, 2), "unit": "tok/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "u8 weights x i8 activations -> i32 block sums, f32 scale accumulate (decode); f16 MFMA, f32 accumulate (prefill)",
            "data": "synthetic (random-bit Q4_K/Q6_K blocks in the Q4_K_M type mix, N(0,1) activations)",
            "config": {"workload": "Llama-3-8B Q4_K_M on %dxMI355X: pp512 + tg128, every MUL_MAT/FUSED_UP_GATE of the graph (225 weight matrices, "
                                   "4.616 GB), no attention/norm/rope ops" % world,
                       "parallelism": "tp%d (row-split q/k/v/up/gate, K-split o/down, RCCL all-reduce x2 per layer)" % world if world > 1 else "single GPU",
                       "pp512_tok_s": round(N_PROMPT * args.steps / (pp_ms * 1e-3), 1), "tg128_tok_s": round(N_GEN * args.steps / (tg_ms * 1e-3), 1),
                       "decode_hip_graph": graph is not None, "weight_bytes_per_rank": model.weight_bytes(), "reduce": reduce_info},
            "roofline": roofline, "roofline_prefill": roofline_prefill, "cpu_baseline": cpu,
        }
        print(json.dumps(out), flush=True)
    be.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
