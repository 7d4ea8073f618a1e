"""Tensor-parallel sharding of the mat-mul path (the reference's `-sm graph`).

Mirrors the split planning of src/llama-load-tensors.cpp:5452-5499: q/k/v/up/gate are split along ne[1] (rows; no
communication), o/down along ne[0] (K; each rank produces a full-size partial sum) and every sharded block ends in ONE
all-reduce(sum) -- the GGML_OP_REDUCE node (ggml.c:6166-6189).  Host logic only: it runs unchanged on CPU (tests use gloo +
the oracle as the compute) and on GPU (Cdna4Backend + RCCL through the C ABI)."""
import os
from .cdna4 import BLCK_SIZE, TYPE_SIZE, BASE_OF


def split_sizes(total, world, granularity, weights=None):
    """Split `total` into `world` contiguous parts, each a multiple of `granularity` (unequal `tensor_split` weights allowed).
    Row splits use granularity = head_dim*gqa for attention (whole KV-head groups, llama-load-tensors.cpp:5459-5465) or 4 for
    _R4 types; K splits use the quant block size (256, or 32 for IQ4_NL) (:5466-5477)."""
    if total % granularity:
        raise ValueError("size %d is not a multiple of the split granularity %d" % (total, granularity))
    units = total // granularity
    if weights is None:
        weights = [1.0] * world
    if len(weights) != world or min(weights) < 0 or sum(weights) <= 0:
        raise ValueError("bad tensor_split weights")
    acc = 0.0; tot = float(sum(weights)); bounds = [0]
    for w in weights:
        acc += w
        bounds.append(int(round(units * acc / tot)))
    bounds[-1] = units
    sizes = [(bounds[i + 1] - bounds[i]) * granularity for i in range(world)]
    if min(sizes) < 0:
        raise ValueError("degenerate split")
    return sizes


def offsets(sizes):
    out = [0]
    for s in sizes:
        out.append(out[-1] + s)
    return out


def shard_rows(w, sizes, rank):
    """row-split (split_dim = 1): rank's contiguous slice of rows of a [M, row_size] quantized weight."""
    o = offsets(sizes)
    return w[o[rank]:o[rank + 1]]


def shard_k(w, t, k_sizes, rank):
    """K-split (split_dim = 0) of quantized rows: whole blocks.  Plain layouts: a byte-column slice of every row (behind the row's meta bytes, which every
    shard keeps: the row scale of the _KS types multiplies each partial sum).  Row-interleaved _R4 layouts: 4 consecutive rows share 4 * row_size bytes made of
    interleaved blocks of 4 * type_size bytes per block of elements, so the slice is taken over the block index of every 4-row group --
    the same bytes as splitting the base-type tensor and repacking each shard (tests/test_tp_gloo.py)."""
    import numpy as np
    from .cdna4 import ROW_META
    bs, ts = BLCK_SIZE[t], TYPE_SIZE[t]
    o = offsets(k_sizes)
    if o[rank] % bs or o[rank + 1] % bs:
        raise ValueError("K split must fall on quant block boundaries")
    b0, b1 = o[rank] // bs, o[rank + 1] // bs
    if t in BASE_OF:
        m, rs = w.shape
        if m % 4:
            raise ValueError("_R4 tensors come in groups of 4 rows")
        g = np.ascontiguousarray(w).reshape(m // 4, (4 * rs) // (4 * ts), 4 * ts)[:, b0:b1]
        return np.ascontiguousarray(g).reshape(m, (b1 - b0) * ts)
    meta = ROW_META.get(t, 0)
    if meta:
        return np.concatenate([w[:, :meta], w[:, meta + b0 * ts:meta + b1 * ts]], axis=1)
    return w[:, b0 * ts:b1 * ts]


class ShardedFFN:
    """One rank's share of the FFN block: fused up*gate (row-split) -> down (K-split) -> all-reduce(sum).
    `matmul(t, w, x)` and `fused_up_gate(t, wu, wg, x)` are the compute callables (backend or oracle), `all_reduce(buf)` the
    collective (RCCL via Cdna4Backend.reduce, or torch.distributed on CPU)."""

    def __init__(self, t_up, w_up, w_gate, t_down, w_down, n_ff, world, rank, fused_up_gate, matmul, all_reduce):
        ff_sizes = split_sizes(n_ff, world, BLCK_SIZE[t_down])          # the ff slice is down's K slice: block aligned
        self.w_up = shard_rows(w_up, ff_sizes, rank); self.w_gate = shard_rows(w_gate, ff_sizes, rank)
        self.w_down = shard_k(w_down, t_down, ff_sizes, rank)
        self.t_up, self.t_down = t_up, t_down
        self.fused_up_gate, self.matmul, self.all_reduce = fused_up_gate, matmul, all_reduce

    def forward(self, x):
        h = self.fused_up_gate(self.t_up, self.w_up, self.w_gate, x)    # [n, ff/world]
        part = self.matmul(self.t_down, self.w_down, h)                 # [n, n_embd] partial sum
        return self.all_reduce(part)                                    # GGML_OP_REDUCE: every rank ends with the full sum


class ShardedAttention:
    """One rank's share of the attention block: q / k / v row-split by WHOLE KV-head groups (a rank owns some KV heads and the gqa q heads that read them: the attention itself
    needs no communication, llama-load-tensors.cpp:5459-5465), attn_output K-split over the same q dimensions, ONE all-reduce(sum) at the end (GGML_OP_REDUCE).
    `matmul(t, w, x)` is the compute callable, `attention(q, k, v, n_head, n_head_kv, head_dim)` the (causal) attention over this rank's heads, `all_reduce(buf)` the collective.
    `weights`: unequal shares of the KV heads (tensor_split)."""

    def __init__(self, t_q, wq, t_k, wk, t_v, wv, t_o, wo, n_head, n_head_kv, head_dim, world, rank, matmul, attention, all_reduce, weights=None):
        if n_head % n_head_kv:
            raise ValueError("n_head must be a multiple of n_head_kv")
        gqa = n_head // n_head_kv
        kv_heads = split_sizes(n_head_kv, world, 1, weights)                                   # KV heads per rank (a rank may get none with extreme weights: rejected below)
        if min(kv_heads) == 0:
            raise ValueError("every rank needs at least one KV head (n_head_kv %d over %d ranks)" % (n_head_kv, world))
        q_rows = [h * gqa * head_dim for h in kv_heads]; kv_rows = [h * head_dim for h in kv_heads]
        if any(r % BLCK_SIZE[t_o] for r in q_rows):
            raise ValueError("attn_output's K slices (%s) must fall on its quant block boundaries (%d)" % (q_rows, BLCK_SIZE[t_o]))
        self.wq = shard_rows(wq, q_rows, rank); self.wk = shard_rows(wk, kv_rows, rank); self.wv = shard_rows(wv, kv_rows, rank)
        self.wo = shard_k(wo, t_o, q_rows, rank)
        self.t = (t_q, t_k, t_v, t_o); self.n_head, self.n_head_kv, self.head_dim = kv_heads[rank] * gqa, kv_heads[rank], head_dim
        self.matmul, self.attention, self.all_reduce = matmul, attention, all_reduce

    def forward(self, x):
        t_q, t_k, t_v, t_o = self.t
        q = self.matmul(t_q, self.wq, x); k = self.matmul(t_k, self.wk, x); v = self.matmul(t_v, self.wv, x)
        o = self.attention(q, k, v, self.n_head, self.n_head_kv, self.head_dim)              # [n, n_head_local * head_dim]
        return self.all_reduce(self.matmul(t_o, self.wo, o))                                  # partial sum over this rank's q dimensions -> full sum on every rank


def setup_ipc_windows(be, dist, rank, world, device, log=lambda *a: None, max_bytes=8 << 20):
    """One process per GPU: create this rank's IPC window (cdna4_window_create), exchange the handles over torch.distributed, attach the peers' windows and
    keep them only if an all-reduce through them -- f32, and f32 with a bf16 wire -- reproduces dist.all_reduce on EVERY rank.  Returns True with the windows
    attached on all ranks, or False with them released on all ranks.  Every torch.distributed collective in here is executed by every rank whatever failed
    locally (a rank that skipped one would hang the others); the window kernels themselves wait for their peers with a bound."""
    import torch

    def agree(ok):
        f = torch.tensor([1 if ok else 0], device=device); dist.all_reduce(f, op=dist.ReduceOp.MIN); return int(f.item()) == 1
    try:
        mine = be.window_create(rank, world, max_bytes)
    except Exception as e:      # noqa: BLE001
        log("IPC window: %r" % (e,)); mine = None
    handles = [None] * world
    dist.all_gather_object(handles, mine)
    ok = all(h is not None for h in handles)
    if ok:
        try:
            for r in range(world):
                if r != rank:
                    be.window_attach(r, handles[r])
        except Exception as e:  # noqa: BLE001
            log("IPC window attach: %r" % (e,)); ok = False
    ok = agree(ok)
    if ok:
        x = torch.sin(torch.arange(8192, device=device, dtype=torch.float32) * (rank + 1))
        y = x.clone(); z = x.clone(); w = x.clone()
        dist.all_reduce(z)
        try:
            be.window_reduce(y, check=True); be.window_reduce(w, check=True, wire=torch.bfloat16)
            if device.type == "cuda":
                torch.cuda.synchronize()
            ok = bool(torch.allclose(y, z, rtol=0, atol=1e-5 * world) and torch.allclose(w, z, rtol=0, atol=2e-2 * world))
        except Exception as e:  # noqa: BLE001
            log("IPC window reduce: %r" % (e,)); ok = False
        ok = agree(ok)
    if ok and world > 2 and max_bytes >= (4 << 20):
        # prompt-size messages of more than two ranks take the TWO-SHOT form (reduce-scatter + all-gather, from CDNA4_WINDOW_TWO_SHOT_MIN bytes on the wire): validate it too, on a
        # 4 MiB message in both wire types.  If only this form fails the windows stay, for messages below the cross-over (the collective library carries the rest).
        two_shot_min = int(os.environ.get("CDNA4_WINDOW_TWO_SHOT_MIN", 256 * 1024))
        x = torch.sin(torch.arange(1 << 20, device=device, dtype=torch.float32) * (0.001 * (rank + 1)))
        y = x.clone(); z = x.clone(); w = x.clone()
        dist.all_reduce(z)
        try:
            be.window_reduce(y, check=True); be.window_reduce(w, check=True, wire=torch.bfloat16)
            if device.type == "cuda":
                torch.cuda.synchronize()
            big_ok = bool(torch.allclose(y, z, rtol=0, atol=1e-5 * world) and torch.allclose(w, z, rtol=0, atol=2e-2 * world))
        except Exception as e:  # noqa: BLE001
            log("IPC window two-shot reduce: %r" % (e,)); big_ok = False
        if not agree(big_ok):
            log("IPC windows: the two-shot form did not validate; windows kept for messages below %d bytes on the wire" % two_shot_min)
            be.window_bytes = min(be.window_bytes, two_shot_min - 16)
    if not ok:
        log("IPC windows unavailable: the collective library for every reduce")
        try:
            be.window_free()
        except Exception:       # noqa: BLE001
            pass
    return ok
