"""CPU: the mmcv-free training loop pieces (pcr_amd/train.py) -- cyclic lr / momentum values, one-bucket gradient
exchange over gloo (2 ranks == 1 rank on the concatenated batch), clipping, accumulation, mmcv-layout checkpoints."""
import math
import os
import subprocess
import sys
import textwrap

import pytest
import torch

from conftest import ROOT
from pcr_amd import train


def test_cyclic_schedule_values():
    # configs_reid/_base_/schedules/cyclic_*.py: lr 3e-4, target_ratio (10, 1e-4), one cycle, 40 % up
    v = lambda it: train.cyclic_value(3e-4, it, 100)
    assert v(0) == pytest.approx(3e-4, rel=1e-12)
    assert v(20) == pytest.approx(1.65e-3, rel=1e-12)                  # half way up the cosine: mean of 3e-4 and 3e-3
    assert v(40) == pytest.approx(3e-3, rel=1e-12)                     # peak, first iteration of the down phase
    assert v(70) == pytest.approx(0.5 * (3e-3 + 3e-8), rel=1e-12)      # half way down
    assert v(99) == pytest.approx(3e-8 + 0.5 * (3e-3 - 3e-8) * (math.cos(math.pi * 59 / 60) + 1), rel=1e-12)
    m = lambda it: train.cyclic_value(0.9, it, 100, target_ratio=(0.85 / 0.95, 1.0))
    assert m(0) == pytest.approx(0.9) and m(40) == pytest.approx(0.9 * 0.85 / 0.95) and m(99) < 0.9
    # two cycles: the second one restarts
    assert train.cyclic_value(1.0, 50, 100, cyclic_times=2) == pytest.approx(1.0)


class Toy(torch.nn.Module):
    """a model with the train_step interface and one parameter that never receives a gradient"""

    def __init__(self):
        super().__init__()
        torch.manual_seed(3)
        self.a = torch.nn.Linear(6, 8)
        self.b = torch.nn.Linear(8, 1)
        self.unused = torch.nn.Parameter(torch.ones(5))

    def train_step(self, data, optimizer):
        y = self.b(torch.tanh(self.a(data["x"]))).squeeze(1)
        loss = torch.nn.functional.binary_cross_entropy_with_logits(y, data["t"])
        return dict(loss=loss, log_vars={"loss": float(loss.detach())}, num_samples=len(data["t"]))


def _batches():
    g = torch.Generator().manual_seed(11)
    return [dict(x=torch.randn(8, 6, generator=g), t=(torch.rand(8, generator=g) > 0.5).float()) for _ in range(4)]


def _run_single(cumulative_iters=1):
    m = Toy()
    tr = train.Trainer(m, max_iters=4, lr=1e-2, grad_clip=0.5, cumulative_iters=cumulative_iters)
    outs = [tr.step(b) for b in _batches()]
    return m, tr, outs


def test_trainer_single_process_and_checkpoint(tmp_path):
    m, tr, outs = _run_single()
    assert tr.iter == 4 and all(math.isfinite(float(o["loss"])) for o in outs)
    assert outs[0]["lr"] == pytest.approx(1e-2) and outs[1]["lr"] > outs[0]["lr"]
    assert all(o["grad_norm"] > 0 for o in outs)
    assert m.unused.grad is None and len(tr.bucket.live) == 4       # the never-used parameter is not in the bucket
    path = str(tmp_path / "ck.pth")
    tr.save(path)
    ck = torch.load(path, weights_only=False)
    assert set(ck) == {"meta", "state_dict", "optimizer"} and ck["meta"]["iter"] == 4
    m2 = Toy()
    with torch.no_grad():
        m2.a.weight.zero_()
    tr2 = train.Trainer(m2, max_iters=4, lr=1e-2)
    tr2.load(path)
    assert tr2.iter == 4 and all(torch.equal(p, q) for p, q in zip(m.parameters(), m2.parameters()))
    # accumulation over 2 iterations = one step on the mean of the two losses
    _, tr3, outs3 = _run_single(cumulative_iters=2)
    assert "grad_norm" not in outs3[0] and "grad_norm" in outs3[1]


def test_graph_signature_refuses_what_a_replay_cannot_refresh_and_sees_the_training_arithmetic():
    """a list input with a non-tensor member -> None (eager with a warning), not an AttributeError; the
    captured iteration's arithmetic (train_ops.TRAIN_PRECISION) is part of the signature"""
    from pcr_amd import train_ops
    tr = train.Trainer(Toy(), max_iters=2, lr=1e-2)
    good = dict(x=[torch.zeros(3), torch.zeros(3)], flag=True)
    assert tr._graph_signature(good, {"x"}) is not None
    assert tr._graph_signature(dict(x=[torch.zeros(3), "oops"]), {"x"}) is None
    assert tr._graph_signature(dict(x=[torch.zeros(3), torch.zeros(4)]), {"x"}) is None
    a = tr._graph_signature(good, {"x"})
    prev = train_ops.set_train_precision("f32" if train_ops.TRAIN_PRECISION != "f32" else "bf16x3")
    try:
        assert tr._graph_signature(good, {"x"}) != a
    finally:
        train_ops.set_train_precision(prev)


WORKER = textwrap.dedent("""
    import os, sys
    sys.path.insert(0, os.path.join(%r, "point-cloud-reid_amd"))
    sys.path.insert(0, os.path.join(%r, "tests"))
    import torch, torch.distributed as dist
    from pcr_amd import shard, train
    import test_train_loop as T
    rank, local, world = shard.init(backend="gloo")
    m = T.Toy()
    tr = train.Trainer(m, max_iters=4, lr=1e-2, grad_clip=0.5)
    for b in T._batches():
        lo, hi = shard.shard_range(8, rank, world)
        tr.step(dict(x=b["x"][lo:hi], t=b["t"][lo:hi]))
    assert tr.bucket.nbytes() == 4 * (6 * 8 + 8 + 8 + 1)             # the unused parameter is not in the bucket
    ref, _, _ = T._run_single()                                      # the same four steps on the whole batch
    for p, q in zip(m.parameters(), ref.parameters()):
        assert torch.allclose(p, q, atol=1e-6), (p - q).abs().max()
    dist.barrier()
    dist.destroy_process_group()
    sys.stdout.write("rank %%d ok\\n" %% rank); sys.stdout.flush()
""")


def test_two_rank_gradient_exchange(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER % (ROOT, ROOT))
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


class ToyParsed(Toy):
    """the same model logging through the reference's path: ReIDNet._parse_losses (mmdet's all-reduce of the log scalars)"""

    def train_step(self, data, optimizer):
        from mmdet3d.models.ReIDNet import ReIDNet
        y = self.b(torch.tanh(self.a(data["x"]))).squeeze(1)
        loss, log_vars = ReIDNet._parse_losses({"reid_loss": torch.nn.functional.binary_cross_entropy_with_logits(y, data["t"])})
        return dict(loss=loss, log_vars=log_vars, num_samples=len(data["t"]))


WORKER_LOG = textwrap.dedent("""
    import os, sys
    sys.path.insert(0, os.path.join(%r, "point-cloud-reid_amd"))
    sys.path.insert(0, os.path.join(%r, "tests"))
    import torch, torch.distributed as dist
    from pcr_amd import shard, train
    import test_train_loop as T
    rank, local, world = shard.init(backend="gloo")
    calls = []
    real = dist.all_reduce
    def counted(t, *a, **k):
        calls.append(t.numel())
        return real(t, *a, **k)
    dist.all_reduce = counted
    m = T.ToyParsed()
    tr = train.Trainer(m, max_iters=4, lr=1e-2, grad_clip=0.5)
    for b in T._batches():
        lo, hi = shard.shard_range(8, rank, world)
        n0 = len(calls)
        out = tr.step(dict(x=b["x"][lo:hi], t=b["t"][lo:hi]))
        assert len(calls) == n0 + 1, calls                          # ONE collective per iteration: gradients + log scalars
        mine = torch.tensor([float(out["loss"])], dtype=torch.float64)
        both = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(both, mine)
        mean = float((both[0] + both[1]) / 2)
        assert abs(out["log_vars"]["loss"] - mean) < 1e-6 and abs(out["log_vars"]["reid_loss"] - mean) < 1e-6
        assert abs(float(both[0]) - float(both[1])) > 1e-4          # (the ranks really saw different shards)
    assert tr.bucket.tail == 2 and calls[-1] == (6 * 8 + 8 + 8 + 1) + 2
    # outside a Trainer the model keeps mmdet's behaviour: its own all-reduce of the log scalars
    n0 = len(calls)
    out = m.train_step(dict(x=T._batches()[0]["x"], t=T._batches()[0]["t"]), None)
    assert len(calls) == n0 + 1 and calls[-1] == 2 and not out["log_vars"].deferred()
    # gradient accumulation: no exchange every iteration, so the scalars keep their own collective
    tr2 = train.Trainer(T.ToyParsed(), max_iters=4, lr=1e-2, cumulative_iters=2)
    n0 = len(calls)
    tr2.step(dict(x=T._batches()[0]["x"], t=T._batches()[0]["t"]))
    assert calls[n0:] == [2], calls[n0:]
    dist.barrier()
    dist.destroy_process_group()
    sys.stdout.write("rank %%d ok\\n" %% rank); sys.stdout.flush()
""")


def test_two_rank_log_scalars_ride_in_the_gradient_bucket(tmp_path):
    """the log-scalar all-reduce of `_parse_losses` is hoisted out of the (capturable) forward +
    backward into the tail of the gradient bucket -- one collective per iteration, same logged values"""
    script = tmp_path / "worker.py"
    script.write_text(WORKER_LOG % (ROOT, ROOT))
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
