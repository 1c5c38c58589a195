"""GPU: the data-parallel training pieces with DEVICE tensors -- two ranks (gloo; both on cuda:0, RCCL refuses two ranks
on one device) run the bucketed gradient exchange (foreach pack / unpack) and the HIP norm + clip + AdamW launches;
the result must equal one rank on the concatenated batch.  The 8-GPU RCCL run itself is the driver's."""
import os
import subprocess
import sys
import textwrap

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

WORKER = textwrap.dedent("""
    import os, sys
    sys.path.insert(0, os.path.join(%r, "point-cloud-reid_amd"))
    sys.path.insert(0, os.path.join(%r, "tests"))
    import torch, torch.distributed as dist
    from pcr_amd import shard, train
    from pcr_amd.optim import FusedAdamW
    import test_train_loop as T
    rank, local, world = shard.init(backend="gloo")
    torch.cuda.set_device(0)

    def run(sharded):
        m = T.Toy().cuda()
        tr = train.Trainer(m, max_iters=4, lr=1e-2, grad_clip=0.5)
        assert isinstance(tr.optimizer, FusedAdamW) and tr.fused
        norms = []
        for b in T._batches():
            lo, hi = shard.shard_range(8, rank, world) if sharded else (0, 8)
            out = tr.step(dict(x=b["x"][lo:hi].cuda(), t=b["t"][lo:hi].cuda()))
            norms.append(float(out["grad_norm"]))
        return m, tr, norms

    class NoDist:      # the single-rank reference: same process, exchange switched off
        def __enter__(self):
            self.f = shard.is_dist
            shard.is_dist = lambda: False
        def __exit__(self, *a):
            shard.is_dist = self.f
    m, tr, norms = run(True)
    assert tr.bucket.nbytes() == 4 * (6 * 8 + 8 + 8 + 1) and tr.bucket.flat.is_cuda
    with NoDist():
        ref, _, ref_norms = run(False)
    for p, q in zip(m.parameters(), ref.parameters()):
        assert torch.allclose(p, q, atol=1e-6), float((p - q).abs().max())
    assert all(abs(a - b) < 1e-5 * max(1.0, abs(b)) for a, b in zip(norms, ref_norms)), (norms, ref_norms)
    dist.barrier()
    dist.destroy_process_group()
    sys.stdout.write("rank %%d ok\\n" %% rank); sys.stdout.flush()
""")


def test_two_ranks_on_device_tensors(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER % (ROOT, ROOT))
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(port), str(script)]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "rank 0 ok" in r.stdout and "rank 1 ok" in r.stdout


@pytest.mark.parametrize("workload", ["pt128", "pt128_train", "ssg1024", "pt1024"])
def test_bench_runs_two_ranks_on_the_gpu(workload):
    """bench.py's multi-rank path end to end on the GPU (pair sharding, barriers, max-over-ranks timing, rank 0's JSON
    line, the gradient bucket under the training workload): two ranks on cuda:0 over gloo through the test hooks"""
    import json
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0", PCR_BENCH_TEST_BACKEND="gloo",
               PCR_BENCH_TEST_SAME_DEVICE="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2",
           "--steps", "2", "--warmup", "1", "--workload", workload, "--pairs", "32", "--no-cpu-baseline"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["value"] > 0 and d["scaling"] == "weak" and "roofline" in d
    assert d["config"].get("rccl_ranks", 2) == 2
    # the companions of the default run and the CPU baseline belong to the one-GPU line only; every rank's own time is
    # in the line (a straggler shows as a spread; ms_per_step is their maximum)
    assert "also" not in d and "cpu_baseline" not in d
    assert len(d["per_rank_ms_per_step"]) == 2 and max(d["per_rank_ms_per_step"]) == pytest.approx(d["ms_per_step"])


def test_default_two_rank_run_carries_the_training_companion():
    """the default `bench.py --gpus N` line at N > 1 also records the training step (forward + backward +
    ONE gradient-bucket all-reduce + clip + AdamW on every rank), so that a scaling run exercises the collective
    north_star names; the last stdout line stays compact"""
    import json
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0", PCR_BENCH_TEST_BACKEND="gloo",
               PCR_BENCH_TEST_SAME_DEVICE="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2",
           "--steps", "2", "--warmup", "2"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    last = r.stdout.strip().splitlines()[-1]
    assert len(last) < 8000
    d = json.loads(last)
    assert d["n_gpus"] == 2 and d["config"]["workload"].startswith("pt1024") and "cpu_baseline" not in d
    (tr,) = d["also"]
    assert tr["name"] == "pt128_train" and tr["n_gpus"] == 2 and tr["value"] > 0 and tr["pairs"] == 256


def test_training_companion_replays_its_graph_on_two_ranks():
    """with N > 1 ranks the training step keeps its HIP-graph replay (the logged loss scalars ride in
    the tail of the gradient bucket instead of being all-reduced inside the captured region)"""
    import json
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0", PCR_BENCH_TEST_BACKEND="gloo",
               PCR_BENCH_TEST_SAME_DEVICE="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2",
           "--steps", "3", "--warmup", "3", "--workload", "pt128_train", "--pairs", "32", "--no-cpu-baseline"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert d["n_gpus"] == 2 and d["config"]["launch"] == "hipgraph", d["config"]
    assert "single-process mode" not in r.stderr


TRAIN_WORKER = textwrap.dedent("""
    import os, sys
    sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "point-cloud-reid_amd")); sys.path.insert(0, os.path.join(%r, "oracle"))
    import torch, torch.distributed as dist
    from pcr_amd import shard, train
    from pcr_amd import testing as T
    import bench
    rank, local, world = shard.init(backend="gloo")
    torch.cuda.set_device(0)
    pairs, n = 8, 128

    def batch(step):
        s1, s2 = T.synthetic_pairs(pairs, n, seed=100 * step + rank, kind="randn")
        ids1 = torch.arange(pairs)
        ids2 = torch.where(torch.arange(pairs) %% 2 == 0, ids1, ids1 + pairs)
        zero = torch.zeros(1, dtype=torch.long, device="cuda")
        return dict(sparse_1=list(s1.cuda()), sparse_2=list(s2.cuda()), dense_1=list(s1.cuda()), dense_2=list(s2.cuda()),
                    label_1=[zero] * pairs, label_2=[zero] * pairs,
                    id_1=[i.view(1).cuda() for i in ids1], id_2=[i.view(1).cuda() for i in ids2])

    def run(graph):
        model, _ = bench.build_pt_model([128, 64, 32])
        model.train()
        tr = train.Trainer(model, max_iters=8, lr=3e-4, grad_clip=1.0, graph=graph)
        tr.graph_warmup = 1
        losses, logged, mine = [], [], []
        for step in range(5):
            out = tr.step(batch(step))
            losses.append(out["loss"].detach().clone())
            logged.append(float(out["log_vars"]["loss"]))
            mine.append(float(out["loss"]))
        return model, tr, torch.stack(losses), logged, mine

    m_g, tr_g, loss_g, log_g, mine_g = run(True)
    assert tr_g.graph and tr_g._g is not None, "the two-rank trainer fell back to eager"
    m_e, tr_e, loss_e, log_e, mine_e = run(False)
    # same launches replayed or issued one by one: the trajectories agree
    assert torch.allclose(loss_g, loss_e, rtol=1e-5, atol=1e-6), (loss_g, loss_e)
    for p, q in zip(m_g.parameters(), m_e.parameters()):
        assert torch.allclose(p, q, rtol=1e-4, atol=1e-6), float((p - q).abs().max())
    # the logged loss is the MEAN over the ranks (mmdet's _parse_losses), delivered by the bucket's tail
    t = torch.tensor(mine_g, dtype=torch.float64)
    both = [torch.zeros_like(t) for _ in range(world)]
    dist.all_gather(both, t)
    mean = (both[0] + both[1]) / 2
    assert not torch.allclose(both[0], both[1]), "the ranks saw the same data"
    assert torch.allclose(torch.tensor(log_g, dtype=torch.float64), mean, rtol=1e-5), (log_g, mean)
    assert torch.allclose(torch.tensor(log_e, dtype=torch.float64), mean, rtol=1e-5), (log_e, mean)
    assert tr_g.bucket.tail == 2 and tr_g.bucket.nbytes() == tr_e.bucket.nbytes()      # [reid_loss, loss] behind the gradients
    dist.barrier()
    dist.destroy_process_group()
    sys.stdout.write("rank %%d ok\\n" %% rank); sys.stdout.flush()
""")


def test_two_rank_graph_replay_equals_the_two_rank_eager_trajectory(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(TRAIN_WORKER % (ROOT, ROOT, ROOT))
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(port), str(script)]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-6000:]
    assert "rank 0 ok" in r.stdout and "rank 1 ok" in r.stdout
