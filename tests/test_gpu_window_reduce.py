"""-m gpu: the one-shot all-reduce over IPC-mapped windows (cdna4_window_*: the one-process-per-GPU form of the reference's P2P one-shot reduce, ggml-cuda/reduce.cu:448-533)
with TWO RANKS = two processes, both on device 0 (the pool's boxes have one GPU; RCCL refuses two ranks on one device, HIP IPC does not) -- or one device per rank
where the node has more.  Each rank reduces a sequence of
messages -- the decode size (one token of n_embd floats), a prompt-size f16 message, a bf16 one -- and compares with the sum computed on the host; a third scenario checks
that a missing peer produces an error after the bounded wait instead of a hang."""
import multiprocessing as mp
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC (the host driver supports nothing else); inherited by the spawned ranks
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


TWO_SHOT_MIN = 256 * 1024          # csrc/reduce.inc: messages of at least this many bytes on the wire take the two-shot (reduce-scatter + all-gather) form ...


def _two_shot(nbytes, world):      # ... with more than 2 ranks (or with CDNA4_WINDOW_TWO_SHOT_ANY_WORLD set)
    return nbytes >= int(os.environ.get("CDNA4_WINDOW_TWO_SHOT_MIN", TWO_SHOT_MIN)) and (world > 2 or "CDNA4_WINDOW_TWO_SHOT_ANY_WORLD" in os.environ)


def _rank_main(rank, world, q_out, q_in, scenario, res):
    try:
        sys.path.insert(0, ROOT)
        import torch
        from __graft_entry__ import _load_package
        pkg = _load_package()
        dev = rank % torch.cuda.device_count()          # one GPU (the pool's boxes): both ranks on device 0; a multi-GPU node: one device per rank, the windows cross xGMI
        torch.cuda.set_device(dev)
        be = pkg.Cdna4Backend(dev)
        handle = be.window_create(rank, world, 8 << 20)
        q_out.put((rank, handle))
        peers = {}
        while len(peers) < world - 1:
            r, h = q_in.get(timeout=60); peers[r] = h
        for r, h in peers.items():
            be.window_attach(r, h)
        ok = True; note = ""
        if scenario == "reduce":
            # token-size (one-shot), prompt-size (two-shot when it applies), a ragged vector count (slices of unequal length), alternating so that both forms share the parities
            msgs = [(4096, torch.float32), (8192, torch.float32), (512 * 4096, torch.float16), (64 * 4096, torch.bfloat16), (4096, torch.float32), (100 * 1028, torch.float32)] * 3
            for i, (n, dt) in enumerate(msgs):
                parts = [torch.from_numpy(np.random.default_rng(1000 * i + r).standard_normal(n).astype(np.float32)) for r in range(world)]
                mine = parts[rank].to(dt).cuda()
                be.window_reduce(mine, check=True)
                want = parts[0].to(dt).float()
                for p in parts[1:]:
                    want = want + p.to(dt).float()                           # f32 accumulate in rank order: what every rank computes, bit for bit
                want = want.to(dt)
                got = mine.cpu()
                if not bool((got == want).all()):
                    ok = False; note += " msg %d (%d x %s): max err %g" % (i, n, dt, float((got.float() - want.float()).abs().max()))
            for i, wire in enumerate((torch.bfloat16, torch.float16)):      # f32 buffers, 16-bit wire: sum of the rounded partials, accumulated in f32
                n = 512 * 4096
                parts = [torch.from_numpy(np.random.default_rng(5000 + 10 * i + r).standard_normal(n).astype(np.float32)) for r in range(world)]
                mine = parts[rank].cuda()
                be.window_reduce(mine, check=True, wire=wire)
                want = parts[0].to(wire).float()
                for p in parts[1:]:
                    want = want + p.to(wire).float()
                if _two_shot(n * 2, world):                                  # two-shot: the reduced slices travel in the wire type too
                    want = want.to(wire).float()
                if not bool((mine.cpu() == want).all()):
                    ok = False; note += " wire %s: max err %g" % (wire, float((mine.cpu() - want).abs().max()))
        elif scenario == "graph":              # the epoch lives in device memory: a captured sequence of reduces replays correctly
            parts = [torch.from_numpy(np.random.default_rng(77 + r).standard_normal(4096).astype(np.float32)) for r in range(world)]
            x = parts[rank].cuda(); y = torch.empty_like(x); z = torch.empty_like(x)
            be.window_reduce(y.copy_(x), check=True)                  # (warm-up outside the capture)
            st = torch.cuda.Stream()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=st):
                for _ in range(4):
                    y.copy_(x); be.window_reduce(y)
                z.copy_(y); z.mul_(0.5); be.window_reduce(z)
            want = sum(parts)
            t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
            for it in range(20):
                y.zero_(); z.zero_()
                if it == 10:
                    t0.record()
                g.replay()
            t1.record(); torch.cuda.synchronize()
            ok = bool((y.cpu() == want).all()) and bool((z.cpu() - 0.5 * world * want).abs().max() <= 1e-5 * float(want.abs().max()) * world)
            note = "%.1f us per captured reduce (copy + reduce, both ranks on one device)" % (t0.elapsed_time(t1) * 1e3 / 50)
        elif scenario == "missing_peer":       # rank 1 never calls: rank 0 must get an error, not a hang
            if rank == 0:
                x = torch.ones(4096, device="cuda")
                try:
                    be.window_reduce(x, check=True); ok = False
                except pkg.Cdna4Error:
                    ok = True
                res.put((rank, ok, "")); ok = None
            else:
                q_in.get(timeout=150)          # keep this rank's window mapped until rank 0 has given up
        be.window_free(); be.close()
        if ok is not None:
            res.put((rank, ok, note))
    except Exception as e:          # noqa: BLE001
        res.put((rank, False, repr(e)))


def _run(scenario, world=2, env=None):
    ctx = mp.get_context("spawn")
    old = {k: os.environ.get(k) for k in (env or {})}
    os.environ.update(env or {})                     # inherited by the spawned ranks
    try:
        return _run_ranks(ctx, scenario, world)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def _run_ranks(ctx, scenario, world):
    qs = [ctx.Queue() for _ in range(world)]; out = ctx.Queue(); res = ctx.Queue()
    procs = [ctx.Process(target=_rank_main, args=(r, world, out, qs[r], scenario, res)) for r in range(world)]
    for p in procs:
        p.start()
    handles = [out.get(timeout=180) for _ in range(world)]
    for r, h in handles:
        for o in range(world):
            if o != r:
                qs[o].put((r, h))
    results = [res.get(timeout=180)]
    if scenario == "missing_peer":
        for r in range(1, world):
            qs[r].put(("done", None))
    results += [res.get(timeout=180) for _ in range(world - 1)]
    for p in procs:
        p.join(timeout=60)
        if p.is_alive():
            p.kill()
    return sorted(results)


def test_two_ranks_on_one_device_reduce_through_ipc_windows():
    for rank, ok, err in _run("reduce"):
        assert ok, (rank, err)


def test_two_ranks_two_shot_protocol():
    """the reduce-scatter + all-gather form with two ranks (it is only the default from three ranks on)"""
    for rank, ok, err in _run("reduce", env={"CDNA4_WINDOW_TWO_SHOT_ANY_WORLD": "1", "CDNA4_WINDOW_TWO_SHOT_MIN": "65536"}):
        assert ok, (rank, err)


def _four_ranks_possible():
    """four ranks need four devices -- or CDNA4_TEST_4RANKS_ONE_DEVICE=1: four processes time-sharing ONE GPU work most of the time, but the hardware scheduler does not
    promise to keep four processes' spinning kernels co-resident, and a rank that is descheduled for longer than the bounded wait (~1 s) turns into a (correctly reported)
    'peer did not arrive' error: measured on the pool's one-GPU boxes, profiles/r03_notes.md.  Two ranks on one device are reliable and cover both protocols."""
    import torch
    return torch.cuda.device_count() >= 4 or os.environ.get("CDNA4_TEST_4RANKS_ONE_DEVICE") == "1"


def test_four_ranks_one_shot_and_two_shot():
    if not _four_ranks_possible():
        pytest.skip("needs 4 GPUs (or CDNA4_TEST_4RANKS_ONE_DEVICE=1)")
    """four ranks (four processes time-sharing device 0 on the pool's one-GPU boxes): token-size messages one-shot, prompt-size ones two-shot (every rank reduces its quarter
    and pulls the other three), identical bits on every rank"""
    for rank, ok, err in _run("reduce", world=4):
        assert ok, (rank, err)


def test_four_ranks_captured_reduces_replay():
    if not _four_ranks_possible():
        pytest.skip("needs 4 GPUs (or CDNA4_TEST_4RANKS_ONE_DEVICE=1)")
    for rank, ok, note in _run("graph", world=4):
        assert ok, (rank, note)


def test_missing_peer_times_out_with_an_error():
    for rank, ok, err in _run("missing_peer"):
        assert ok, (rank, err)


def test_captured_reduces_replay():
    for rank, ok, note in _run("graph"):
        print(note)
        assert ok, (rank, note)
