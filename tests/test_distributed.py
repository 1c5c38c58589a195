"""CPU, world_size 2 over gloo: the multi-GPU layer (pcr_amd/shard.py) used by bench.py --
shard assignment, logits gather order, barrier-bracketed timing with max-reduce, BN-buffer broadcast."""
import os
import subprocess
import sys
import textwrap

import pytest
import torch

from conftest import ROOT
from pcr_amd import shard


def test_shard_range_partitions():
    for n in (0, 1, 7, 512, 513):
        for world in (1, 2, 3, 8):
            spans = [shard.shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


WORKER = textwrap.dedent("""
    import os, sys, time
    sys.path.insert(0, os.path.join(%r, "point-cloud-reid_amd"))
    import torch, torch.distributed as dist
    from pcr_amd import shard
    rank, local, world = shard.init(backend="gloo")
    assert world == 2 and shard.is_dist()
    n = 11
    s1 = torch.arange(n * 6, dtype=torch.float32).view(n, 2, 3)
    s2 = s1 + 1000
    a, b = shard.shard_pairs(s1, s2)
    lo, hi = shard.shard_range(n, rank, world)
    assert a.shape[0] == hi - lo and torch.equal(a, s1[lo:hi]) and torch.equal(b, s2[lo:hi])
    fake_logits = a.sum(dim=(1, 2)) - b.sum(dim=(1, 2))          # stands in for the per-pair model output
    full = shard.gather_logits(fake_logits, n)
    assert torch.equal(full, (s1.sum(dim=(1, 2)) - s2.sum(dim=(1, 2))))
    calls = []
    def step():
        time.sleep(0.02 * (rank + 1))                               # rank 1 is slower
        calls.append(1)
        return len(calls)
    dt, last = shard.timed(step, steps=3, warmup=1)
    assert len(calls) == 4 and last == 4
    assert dt >= 3 * 0.04 * 0.9, dt                                 # the max over ranks, not rank 0's own time
    bn = torch.nn.BatchNorm1d(4)
    with torch.no_grad():
        bn.running_mean.fill_(float(rank + 1)); bn.running_var.fill_(float(10 * (rank + 1)))
    shard.broadcast_buffers(bn, src=0)
    assert float(bn.running_mean[0]) == 1.0 and float(bn.running_var[0]) == 10.0
    dist.barrier()
    dist.destroy_process_group()
    sys.stdout.write("rank %%d ok\\n" %% rank); sys.stdout.flush()
""")


def test_two_rank_gloo(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER % ROOT)
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(port), str(script)]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=240, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "rank 0 ok" in r.stdout and "rank 1 ok" in r.stdout


def _run_bench(*argv, env=None):
    cmd = [sys.executable, os.path.join(ROOT, "bench.py")] + list(argv)
    return subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env or dict(os.environ))


def test_bench_gpus_flag_launches_that_many_ranks():
    """`python bench.py --gpus 2` (no torchrun environment) starts 2 ranks itself -- here over gloo with the
    --dry-run step -- and relays rank 0's single JSON line"""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    r = _run_bench("--gpus", "2", "--dry-run", "--steps", "3", "--warmup", "1", env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["config"]["rccl_ranks"] == 2 and rec["steps"] == 3
    # N = 1 stays a single process
    r = _run_bench("--gpus", "1", "--dry-run", "--steps", "2", "--warmup", "0", env=env)
    assert r.returncode == 0 and json.loads(r.stdout.strip().splitlines()[-1])["n_gpus"] == 1


def test_bench_rejects_a_world_size_that_contradicts_gpus():
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = _run_bench("--gpus", "4", "--dry-run", env=env)
    assert r.returncode != 0 and "WORLD_SIZE" in (r.stdout + r.stderr)
