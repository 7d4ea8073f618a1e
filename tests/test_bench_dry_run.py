"""CPU tier: `bench.py --gpus N --dry-run` under the driver's launch command (torch.distributed.run, one process per rank, 127.0.0.1 rendezvous) -- the launch contract, the
tensor-parallel shard plan and the exchange step of an N-GPU run without a GPU (VERDICT r05 "do this" 10), parsed from the run's own output."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CONTRACT_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config")


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0)); return s.getsockname()[1]


def run_bench(n, config):
    cmd = [sys.executable]
    if n > 1:
        cmd += ["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1", "--master-port", str(free_port())]
    cmd += [os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "2", "--warmup", "1", "--dry-run", "--config", config]
    env = dict(os.environ); env.pop("RANK", None); env.pop("WORLD_SIZE", None); env.pop("LOCAL_RANK", None)
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr.decode(errors="replace")[-2000:]
    lines = [ln for ln in r.stdout.decode().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, "rank 0 prints ONE JSON line, got %d" % len(lines)
    return json.loads(lines[0])


@pytest.mark.parametrize("n", [1, 2, 4])
def test_dry_run_line_and_shard_plan(n):
    one = run_bench(1, "c2") if n > 1 else None
    out = run_bench(n, "c2")
    for k in CONTRACT_KEYS:
        assert k in out, k
    assert out["n_gpus"] == n and out["steps"] == 2 and out["warmup"] == 1 and out["higher_is_better"] is True and out["config"]["dry_run"] is True
    c = out["config"]
    assert ("tp%d" % n in c["parallelism"]) if n > 1 else c["parallelism"] == "single GPU"
    sh = c["per_rank_shapes_layer0"]
    assert sh["wq"] == [4096 // n, 4096] and sh["wk"] == [1024 // n, 4096] and sh["up"] == [14336 // n, 4096] and sh["down"] == [4096, 14336 // n] and sh["wo"] == [4096, 4096 // n]
    if n > 1:      # row / K splits partition the sharded matrices exactly; output.weight is replicated
        assert c["sharded_weight_bytes_all_ranks"] == one["config"]["sharded_weight_bytes_all_ranks"]
        assert c["per_rank_weight_bytes"] - c["replicated_output_bytes"] == one["config"]["sharded_weight_bytes_all_ranks"] // n
    assert c["reduces_per_pass"] == 64 and c["reduce_messages"] == {"token_f32_bytes": 16384, "ubatch_bf16_bytes": 2 * 4096 * 512}


def test_dry_run_70b_shard_config_on_eight_ranks_is_refused_below_its_divisibility():
    """Llama-3-70B (c4shard: 8 KV heads) shards 8 ways; the 8B config does not shard 16 ways (8 KV heads): the plan stops loudly instead of producing a number"""
    out = run_bench(8, "c4shard")
    assert out["n_gpus"] == 8 and out["config"]["per_rank_shapes_layer0"]["up"] == [28672 // 8, 8192]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "16", "--master-addr", "127.0.0.1", "--master-port", str(free_port()),
           os.path.join(ROOT, "bench.py"), "--gpus", "16", "--dry-run"]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300, cwd=ROOT)
    assert r.returncode != 0 and b"does not shard" in r.stderr
