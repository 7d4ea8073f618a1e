"""world_size-2 tensor-parallel tests on CPU (gloo): the sharding plan + reduce contract of the N>1 path, with the oracle as the
compute (the HIP kernels are exercised per-GPU by the -m gpu tests; the collective on GPU is RCCL through the C ABI)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import load_package
from common import activations, random_block_bytes
from oracle import bindings as ob


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, t_up, t_down, n_ff, n_embd, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    pkg = load_package(); from ik_llama_cpp_amd import tp
    orc = ob.Oracle()
    wu = random_block_bytes(t_up, n_ff, n_embd, 1); wg = random_block_bytes(t_up, n_ff, n_embd, 2)
    wd = random_block_bytes(t_down, n_embd, n_ff, 3); x = activations(3, n_embd, 4)

    def all_reduce(buf):
        tt = torch.from_numpy(buf); dist.all_reduce(tt); return tt.numpy()
    ffn = tp.ShardedFFN(t_up, wu, wg, t_down, wd, n_ff, world, rank,
                        fused_up_gate=lambda t, a, b, xx: orc.fused_up_gate(t, 10, a, b, xx),
                        matmul=lambda t, w, xx: orc.mul_mat(t, np.ascontiguousarray(w), xx), all_reduce=all_reduce)
    got = ffn.forward(x)
    if rank == 0:
        # unsharded reference of the same block
        h = orc.fused_up_gate(t_up, 10, wu, wg, x)
        want = orc.mul_mat(t_down, wd, h)
        q.put((got, want))
    dist.barrier(); dist.destroy_process_group()


@pytest.mark.parametrize("t_up,t_down", [(ob.Q4_K, ob.Q6_K), (ob.IQ4_NL, ob.Q4_K)], ids=["q4_K+q6_K", "iq4_nl+q4_K"])
def test_sharded_ffn_matches_unsharded(t_up, t_down):
    world, n_ff, n_embd = 2, 1024, 512
    ctx = mp.get_context("spawn"); q = ctx.Queue(); port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, t_up, t_down, n_ff, n_embd, q)) for r in range(world)]
    [p.start() for p in procs]
    got, want = q.get(timeout=120)
    [p.join(60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    # each rank re-quantizes ITS slice of the hidden activations per 32/256 block exactly like the unsharded run does (K-split on
    # block boundaries), so the partial sums add up to the unsharded result up to f32 summation order
    assert np.allclose(got, want, rtol=1e-5, atol=1e-5 * np.abs(want).max())


def _causal_attention(q, k, v, n_head, n_head_kv, hd):
    """plain numpy causal attention over [n, heads * hd] rows (f64 softmax): the compute stand-in of the sharded block test"""
    n = q.shape[0]; gqa = n_head // n_head_kv; out = np.zeros((n, n_head * hd), np.float32)
    mask = np.triu(np.full((n, n), -np.inf), 1)
    for h in range(n_head):
        qh = q[:, h * hd:(h + 1) * hd].astype(np.float64); kh = k[:, (h // gqa) * hd:(h // gqa + 1) * hd].astype(np.float64); vh = v[:, (h // gqa) * hd:(h // gqa + 1) * hd].astype(np.float64)
        sc = qh @ kh.T / np.sqrt(hd) + mask; sc -= sc.max(axis=1, keepdims=True); p = np.exp(sc); p /= p.sum(axis=1, keepdims=True)
        out[:, h * hd:(h + 1) * hd] = (p @ vh).astype(np.float32)
    return out


def _layer_worker(rank, world, port, weights, q):
    """one transformer layer (attention block + FFN block with their residuals), sharded over `world` ranks: two all-reduces"""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    load_package(); from ik_llama_cpp_amd import tp
    orc = ob.Oracle()
    n_embd, hd, n_ff, n = 512, 128, 1024, 5
    n_head_kv = 4 if weights is None else 8; n_head = 2 * n_head_kv           # equal shares: one KV head per rank; tensor_split 3:1:2:2 over 8 KV heads
    T = ob.Q4_K
    wq = random_block_bytes(T, n_head * hd, n_embd, 11); wk = random_block_bytes(T, n_head_kv * hd, n_embd, 12); wv = random_block_bytes(ob.Q6_K, n_head_kv * hd, n_embd, 13)
    wo = random_block_bytes(T, n_embd, n_head * hd, 14)
    wu = random_block_bytes(T, n_ff, n_embd, 15); wg = random_block_bytes(T, n_ff, n_embd, 16); wd = random_block_bytes(ob.Q6_K, n_embd, n_ff, 17)
    x = activations(n, n_embd, 18)
    n_reduce = [0]

    def all_reduce(buf):
        n_reduce[0] += 1; tt = torch.from_numpy(np.ascontiguousarray(buf)); dist.all_reduce(tt); return tt.numpy()
    mm = lambda t, w, xx: orc.mul_mat(t, np.ascontiguousarray(w), xx)      # noqa: E731
    att = tp.ShardedAttention(T, wq, T, wk, ob.Q6_K, wv, T, wo, n_head, n_head_kv, hd, world, rank, mm, _causal_attention, all_reduce, weights=weights)
    ffn = tp.ShardedFFN(T, wu, wg, ob.Q6_K, wd, n_ff, world, rank, fused_up_gate=lambda t, a, b, xx: orc.fused_up_gate(t, 10, a, b, xx), matmul=mm, all_reduce=all_reduce)
    h1 = x + att.forward(x); got = h1 + ffn.forward(h1)
    if rank == 0:
        a = _causal_attention(mm(T, wq, x), mm(T, wk, x), mm(ob.Q6_K, wv, x), n_head, n_head_kv, hd)
        r1 = x + mm(T, wo, a); want = r1 + mm(ob.Q6_K, wd, orc.fused_up_gate(T, 10, wu, wg, r1))
        q.put((got, want, n_reduce[0], att.n_head_kv))
    dist.barrier(); dist.destroy_process_group()


@pytest.mark.parametrize("weights", [None, [3.0, 1.0, 2.0, 2.0]], ids=["equal", "tensor_split"])
def test_sharded_layer_world_4(weights):
    """world 4 (gloo): attention block with KV-head granularity (4 KV heads over 4 ranks: one each; q heads follow their KV head; attn_output K-split on block boundaries) + FFN
    block; two all-reduces per layer; the result equals the unsharded layer up to the summation order of the partial sums.  tensor_split 3:1:2:2 over 8 KV heads: ranks
    with 3 / 1 / 2 / 2 KV heads (the planner rounds shares to whole KV heads, llama-load-tensors.cpp:5459-5465)."""
    world = 4
    ctx = mp.get_context("spawn"); q = ctx.Queue(); port = _free_port()
    procs = [ctx.Process(target=_layer_worker, args=(r, world, port, weights, q)) for r in range(world)]
    [p.start() for p in procs]
    got, want, n_reduce, kvh = q.get(timeout=180)
    [p.join(60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    assert n_reduce == 2 and kvh >= 1
    # the sharded attn_output / ffn_down re-quantize THEIR K slice of the activations per block exactly like the unsharded run (slices on block boundaries)
    assert np.allclose(got, want, rtol=2e-5, atol=2e-5 * np.abs(want).max())


def test_attention_split_rejects_ranks_without_a_kv_head():
    load_package(); from ik_llama_cpp_amd import tp
    w = np.zeros((8, 144), np.uint8)
    with pytest.raises(ValueError):
        tp.ShardedAttention(ob.Q4_K, w, ob.Q4_K, w, ob.Q4_K, w, ob.Q4_K, w, 8, 2, 128, 4, 0, None, None, None)


def test_split_planner():
    pkg = load_package(); from ik_llama_cpp_amd import tp
    assert tp.split_sizes(28672, 8, 256) == [3584] * 8                   # Llama-3-70B ffn @ TP8
    assert tp.split_sizes(14336, 4, 256) == [3584] * 4
    assert sum(tp.split_sizes(14336, 3, 256)) == 14336 and all(s % 256 == 0 for s in tp.split_sizes(14336, 3, 256))
    assert tp.split_sizes(4096, 2, 1024, weights=[3, 1]) == [3072, 1024]  # unequal tensor_split, KV-head-group granularity
    with pytest.raises(ValueError):
        tp.split_sizes(1000, 2, 256)
    w = np.zeros((8, ob.row_size(ob.Q4_K, 1024)), np.uint8)
    assert tp.shard_k(w, ob.Q4_K, [512, 512], 1).shape == (8, 288)
    with pytest.raises(ValueError):
        tp.shard_k(w, ob.Q4_K, [500, 524], 0)


@pytest.mark.parametrize("t", ob.BASE_TYPES, ids=lambda t: ob.NAMES[t])
def test_k_split_of_r4_tensors_equals_repacked_split(t):
    """K-splitting a row-interleaved tensor = splitting the base tensor and interleaving each shard; and the partial sums of the shards add up to the
    unsharded product (the reduce contract of the K-split mat-muls)."""
    pkg = load_package(); from ik_llama_cpp_amd import tp
    orc = ob.Oracle(); m, k = 16, 1024
    w = random_block_bytes(t, m, k, 11); r4 = orc.repack_r4(t, w, k); sizes = [256, 512, 256]
    for rank in range(3):
        base_shard = np.ascontiguousarray(tp.shard_k(w, t, sizes, rank))
        assert np.array_equal(tp.shard_k(r4, ob.R4_OF[t], sizes, rank), orc.repack_r4(t, base_shard, sizes[rank]))
    with pytest.raises(ValueError):
        tp.shard_k(r4[:6], ob.R4_OF[t], sizes, 0)


@pytest.mark.parametrize("t", [ob.IQ4_KS, ob.IQ2_KS, ob.Q4_0, ob.MXFP4], ids=lambda t: ob.NAMES[t])
def test_k_split_partial_sums_add_up(t):
    """row-scaled types keep their row meta in every shard; 32-block types split on 32-element boundaries"""
    pkg = load_package(); from ik_llama_cpp_amd import tp
    orc = ob.Oracle(); m, k = 8, 1024
    w = random_block_bytes(t, m, k, 12); x = activations(2, k, 13); sizes = [512, 256, 256]; off = [0, 512, 768, 1024]
    want, _ = orc.mul_mat_f64(t, w, x)
    got = sum(orc.mul_mat_f64(t, np.ascontiguousarray(tp.shard_k(w, t, sizes, r)), np.ascontiguousarray(x[:, off[r]:off[r + 1]]))[0] for r in range(3))
    assert np.allclose(got, want, rtol=1e-12, atol=1e-12 * np.abs(want).max())


# ---- tp.setup_ipc_windows: the handshake must end with the SAME verdict on every rank and never leave a rank alone in a collective, whatever fails where
class _FakeWindows:
    """stands in for Cdna4Backend: the 'windows' reduce through torch.distributed; `fail` = (rank, stage) injects a failure on one rank"""
    def __init__(self, rank, fail):
        self.rank, self.fail, self.freed = rank, fail, False

    def _hit(self, stage):
        return self.fail == (self.rank, stage)

    def window_create(self, rank, world, max_bytes):
        if self._hit("create"):
            raise RuntimeError("no IPC here")
        return b"h" * 64

    def window_attach(self, peer, handle):
        assert handle == b"h" * 64
        if self._hit("attach"):
            raise RuntimeError("cannot map the peer")

    def window_reduce(self, buf, check=False, wire=None):
        if self._hit("raise"):
            raise RuntimeError("a peer did not arrive")
        if wire is not None:
            buf.copy_(buf.to(wire).float())
        if self.fail[1] != "raise":          # (a real peer that never calls costs the others their bounded wait, not a collective)
            dist.all_reduce(buf)
        if self._hit("wrong"):
            buf.add_(1.0)
        return buf

    def window_free(self):
        self.freed = True


def _handshake_worker(rank, world, port, fail, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    load_package(); from ik_llama_cpp_amd import tp
    be = _FakeWindows(rank, fail)
    ok = tp.setup_ipc_windows(be, dist, rank, world, torch.device("cpu"))
    t = torch.ones(1); dist.all_reduce(t)           # the ranks are still in step afterwards
    q.put((rank, ok, be.freed, float(t.item())))
    dist.barrier(); dist.destroy_process_group()


@pytest.mark.parametrize("fail", [(-1, "none"), (1, "create"), (0, "attach"), (1, "wrong"), (0, "raise")], ids=lambda f: "%s@%d" % (f[1], f[0]))
def test_ipc_window_handshake_agrees_on_every_rank(fail):
    world = 2
    ctx = mp.get_context("spawn"); q = ctx.Queue(); port = _free_port()
    procs = [ctx.Process(target=_handshake_worker, args=(r, world, port, fail, q)) for r in range(world)]
    [p.start() for p in procs]
    res = sorted(q.get(timeout=120) for _ in range(world))
    [p.join(60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    want = fail[1] == "none"
    for rank, ok, freed, t in res:
        assert ok == want and freed == (not want) and t == world, (rank, ok, freed, t)
