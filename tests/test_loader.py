"""CPU: the training-side data path (pcr_amd/loader.py) -- the training pair rule on an on-disk toy crop directory,
mmdet's DistributedGroupSampler restated, virtual DataLoader workers, and the epoch loop: two gloo ranks fed by the
loader train to the same weights as one process stepping on the concatenated batches."""
import os
import subprocess
import sys
import textwrap

import numpy as np
import pytest
import torch

from conftest import ROOT
from pcr_amd import loader as LD
from pcr_amd import pairs as PR


def make_crops(root, n_true=14, n_fp=6, seed=0):
    """<root>/<token>/<observation>/pts_xyz.bin, the reference's layout (object_loader_base.py:247-269)"""
    g = np.random.default_rng(seed)
    meta = {}
    for i in range(n_true + n_fp):
        fp = i >= n_true
        tok = ("FP_%02d" if fp else "obj_%02d") % i
        meta[tok] = dict(cls=i % 2, fp=fp)
        for obs in range(int(g.integers(2, 6))):
            d = os.path.join(root, tok, str(obs))
            os.makedirs(d)
            n = int(g.integers(3, 200))
            (g.standard_normal((n, 3)).astype(np.float32) + i).tofile(os.path.join(d, "pts_xyz.bin"))
    return meta


def dataset(root, meta, seed=0):
    crops = LD.CropDirectory(root)
    table = crops.table(meta, num_classes=2)
    np.random.seed(seed)                      # set_seeds(cfg.seed) precedes the dataset's shuffle in the reference
    return LD.TrainPairs(table, crops.read, subsample_sparse=32, subsample_dense=8)


def test_distributed_group_sampler_shards_and_reseeds():
    flags = np.zeros(37, dtype=np.uint8)
    per_rank = [LD.DistributedGroupSampler(flags, samples_per_gpu=4, num_replicas=3, rank=r, seed=5) for r in range(3)]
    assert all(len(s) == 16 for s in per_rank)                 # ceil(37 / 4 / 3) * 4
    got = [list(s) for s in per_rank]
    flat = sum(got, [])
    assert len(flat) == 48 and set(flat) == set(range(37))     # padded by repetition, every sample present
    assert got == [list(s) for s in per_rank]                  # same epoch: same order
    for s in per_rank:
        s.set_epoch(1)
    assert [list(s) for s in per_rank] != got
    # whole batches stay together through the batch permutation: a rank's list is made of runs of 4 that are also
    # runs of the un-permuted padded list
    one = LD.DistributedGroupSampler(flags, 4, 1, 0, seed=5)
    g = torch.Generator()
    g.manual_seed(5)
    base = np.arange(37)[torch.randperm(37, generator=g).numpy()].tolist()
    base = base + base[:40 - 37]
    runs = {tuple(base[i:i + 4]) for i in range(0, 40, 4)}
    lst = list(one)
    assert all(tuple(lst[i:i + 4]) in runs for i in range(0, 40, 4))


def test_training_pair_rule(tmp_path):
    meta = make_crops(str(tmp_path))
    ds = dataset(str(tmp_path), meta)
    t = ds.table
    # (the reference keeps objects with MORE than two observations: `temp > 2`, reidentification_base.py:214)
    assert len(ds) == sum(1 for o in t.objects if not o["fp"] and len(o["frames"]) >= 3)
    np.random.seed(1)
    items = [ds[i % len(ds)] for i in range(200)]
    pos = [it for it in items if it["id_1"] == it["id_2"]]
    neg = [it for it in items if it["id_1"] != it["id_2"]]
    assert 60 < len(pos) < 140                                  # the coin
    for it in items:
        assert it["sparse_1"].shape == (32, 3) and it["sparse_2"].shape == (32, 3) and it["dense_2"].shape == (8, 3)
    fps = [it for it in neg if it["id_2"] == -1]
    assert fps and all(it["label_2"] == it["label_1"] + 2 for it in fps)          # FP classes are offset by num_classes
    assert all(it["label_2"] == it["label_1"] for it in neg if it["id_2"] != -1)  # true negatives: same class
    # crops are object-centred around their index (make_crops): a positive's two clouds share a centre, a negative's do not
    c = lambda p: float(np.mean(p))                              # noqa: E731
    assert all(abs(c(it["sparse_1"]) - c(it["sparse_2"])) < 0.9 for it in pos)
    assert sum(abs(c(it["sparse_1"]) - c(it["sparse_2"])) > 0.9 for it in neg) > 0.9 * len(neg)
    np.random.seed(1)
    again = [ds[i % len(ds)] for i in range(200)]
    assert all(np.array_equal(a["sparse_2"], b["sparse_2"]) and a["id_2"] == b["id_2"] for a, b in zip(items, again))


def test_epoch_loader_is_reproducible_and_worker_seeded(tmp_path):
    meta = make_crops(str(tmp_path))
    mk = lambda rank, workers: LD.EpochLoader(dataset(str(tmp_path), meta), samples_per_gpu=4, num_replicas=2,   # noqa: E731
                                              rank=rank, seed=7, num_workers=workers)
    a = [b for b in mk(0, 2).epoch(0)]
    b = [b for b in mk(0, 2).epoch(0)]
    assert len(a) == len(mk(0, 2)) and len(a) >= 1
    same = lambda x, y: all(torch.equal(p, q) for k in x for p, q in zip(x[k], y[k]))     # noqa: E731
    assert all(same(x, y) for x, y in zip(a, b))
    other_rank = [b for b in mk(1, 2).epoch(0)]
    assert not all(same(x, y) for x, y in zip(a, other_rank))
    assert set(a[0]) == {"sparse_1", "sparse_2", "dense_1", "dense_2", "label_1", "label_2", "id_1", "id_2"}
    assert LD.worker_seed(4, 3, 2, 10) == 4 * 3 + 2 + 10
    # the caller's generator is left where it was (the virtual workers swap numpy's state in and out)
    ld = mk(0, 2)
    np.random.seed(123)
    before = np.random.get_state()[1].copy()
    list(ld.epoch(1))
    assert np.array_equal(np.random.get_state()[1], before)


def test_val_pairs_literal_exclusion_flag():
    from test_data_format import _object_table
    t = _object_table()
    pos, neg = PR.build_val_pairs(t, 4, seed=0)
    pos_l, neg_l = PR.build_val_pairs(_object_table(), 4, seed=0, literal_exclusion=True)
    assert pos == pos_l and neg != neg_l
    for p, n in zip(pos_l, neg_l):       # the reference's rule as written: the object indexed by the observation number
        assert n["tok2"] != t.objects[p["o1"]]["token"]


class TinyReID(torch.nn.Module):
    """a model with ReIDNet's train_step interface on the loader's data dict (CPU, a few parameters)"""

    def __init__(self):
        super().__init__()
        torch.manual_seed(4)
        self.enc = torch.nn.Linear(3, 8)
        self.head = torch.nn.Linear(16, 1)

    def train_step(self, data, optimizer):
        s1, s2 = torch.stack(data["sparse_1"]), torch.stack(data["sparse_2"])
        f = torch.cat([torch.tanh(self.enc(s1)).max(1)[0], torch.tanh(self.enc(s2)).max(1)[0]], dim=1)
        y = self.head(f).squeeze(1)
        t = (torch.cat(data["id_1"]) == torch.cat(data["id_2"])).float()
        loss = torch.nn.functional.binary_cross_entropy_with_logits(y, t)
        return dict(loss=loss, log_vars={}, num_samples=len(t))


def single_process_reference(root, meta, epochs, world=2, spg=4):
    """what the two ranks compute together: at every step the ranks' batches concatenated, one step on the mean loss"""
    from pcr_amd import train
    m = TinyReID()
    loaders = [LD.EpochLoader(dataset(root, meta), spg, world, r, seed=7, num_workers=2) for r in range(world)]
    tr = train.Trainer(m, max_iters=epochs * len(loaders[0]), lr=1e-2, grad_clip=1.0)
    for ep in range(epochs):
        for parts in zip(*[ld.epoch(ep) for ld in loaders]):
            tr.step({k: sum((p[k] for p in parts), []) for k in parts[0]})
    return m


WORKER = textwrap.dedent("""
    import os, sys, json
    sys.path.insert(0, os.path.join(%(root)r, "point-cloud-reid_amd"))
    sys.path.insert(0, os.path.join(%(root)r, "tests"))
    import torch, torch.distributed as dist
    from pcr_amd import shard, train, loader as LD
    import test_loader as T
    rank, local, world = shard.init(backend="gloo")
    meta = json.load(open(os.path.join(%(data)r, "..", "meta.json")))
    m = T.TinyReID()
    ld = LD.EpochLoader(T.dataset(%(data)r, meta), 4, world, rank, seed=7, num_workers=2)
    tr = train.Trainer(m, max_iters=2 * len(ld), lr=1e-2, grad_clip=1.0)
    losses = LD.run_epochs(tr, ld, 2)
    assert tr.epoch == 2 and tr.iter == 2 * len(ld) and len(losses) == 2 * len(ld)
    ref = T.single_process_reference(%(data)r, meta, 2)
    for p, q in zip(m.parameters(), ref.parameters()):
        assert torch.allclose(p, q, atol=1e-6), float((p - q).abs().max())
    dist.barrier()
    dist.destroy_process_group()
    sys.stdout.write("rank %%d ok\\n" %% rank); sys.stdout.flush()
""")


def test_two_ranks_train_two_epochs_from_disk_like_one(tmp_path):
    import json
    import socket
    data = tmp_path / "crops"
    data.mkdir()
    meta = make_crops(str(data))
    (tmp_path / "meta.json").write_text(json.dumps(meta))
    script = tmp_path / "worker.py"
    script.write_text(WORKER % dict(root=ROOT, data=str(data)))
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(port), str(script)]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "rank 0 ok" in r.stdout and "rank 1 ok" in r.stdout
