"""Training-side data path of the ReID datasets without mmcv / mmdet: the training pair rule, the
rank-sharded batch order and the epoch loop that feeds `pcr_amd.train.Trainer` from an on-disk crop directory.

What is restated, and from where:

* `TrainPairs.__getitem__` -- `ReIDDatasetNuscenesFP.__getitem__` (mmdet3d/datasets/reidentification_nuscenes.py:37-72)
  with `get_random_other_even_train` (reidentification_base.py:316-357), `get_random_frame` /
  `get_random_frame_even` / `get_class_list_density` (object_loader_base.py:149-152, 201-238) and `return_item`
  (reidentification_base.py:427-438): a coin flip between a POSITIVE (two different observations of the indexed object)
  and a NEGATIVE (one observation of it against an observation of another object -- a second coin flip decides between
  a true object of the same class and a false-positive detection of that class -- drawn from a point-count bucket
  sampled from the positive object's own bucket distribution).  Same sequence of calls into numpy's GLOBAL generator
  (`np.random.choice`, `np.random.randn` for the stand-in dense cloud of a false positive, `np.random.randint` inside
  `subsamplePC`), over the object table of pcr_amd/pairs.py.
* `DistributedGroupSampler` -- mmdet's sampler of that name (mmdet 2.x, `mmdet/datasets/samplers/group_sampler.py`;
  third-party, the reference vendors only its call site: bugfix/data_loader_builder.py:155-164, and pins no version):
  per epoch a `torch.Generator` seeded with epoch + seed permutes every group, pads it to a multiple of
  samples_per_gpu * world, permutes whole batches, and rank r takes the r-th contiguous share.  The ReID datasets put
  every sample in group 0 (`self.flag = np.zeros`, reidentification_base.py:65-67).
* `worker_seed` / `EpochLoader` -- `worker_init_fn` (bugfix/data_loader_builder.py:195-199): worker w of rank r seeds
  numpy with num_workers * r + w + seed; DataLoader hands batch i of a rank to worker i mod num_workers, and (workers
  not being persistent, the reference's default) every epoch starts from freshly seeded workers.  Here the workers are
  virtual: the loader swaps numpy's global state per batch, so the samples are the ones those processes would draw, in
  one process and in a reproducible order.

PARITY: the pair rule is pinned -- `tests/golden/pairs_toy.npz` holds the items the reference's own
`ReIDDatasetNuscenesFP` returns on a toy crop directory (oracle/make_golden.py gen_pairs; the dataset and loader classes
imported unmodified, stand-ins only for the absent lamtk / mmcv / mmdet imports), and `tests/test_pairs_golden.py`
requires `TrainPairs` to return the same tensors, labels and ids item for item and to leave numpy's generator in the same
state.  The sampler stays a restatement of mmdet 2.x (third party, not vendored by the reference, no version pinned
there): its tests pin the sharding invariants, reproducibility under a seed, and that two ranks fed by this loader train
to the same weights as one process stepping on the concatenated batches.
"""
import math

import numpy as np
import torch

from . import data as D
from .pairs import BUCKETS


def worker_seed(num_workers, rank, worker_id, seed):
    return num_workers * rank + worker_id + seed


class DistributedGroupSampler:
    def __init__(self, flags, samples_per_gpu=1, num_replicas=1, rank=0, seed=0):
        self.flag = np.asarray(flags, dtype=np.int64)
        self.samples_per_gpu, self.num_replicas, self.rank = int(samples_per_gpu), int(num_replicas), int(rank)
        self.seed = 0 if seed is None else int(seed)
        self.epoch = 0
        self.group_sizes = np.bincount(self.flag)
        self.num_samples = 0
        for size in self.group_sizes:
            self.num_samples += int(math.ceil(size * 1.0 / self.samples_per_gpu / self.num_replicas)) * self.samples_per_gpu
        self.total_size = self.num_samples * self.num_replicas

    def set_epoch(self, epoch):
        self.epoch = int(epoch)

    def __len__(self):
        return self.num_samples

    def __iter__(self):
        g = torch.Generator()
        g.manual_seed(self.epoch + self.seed)
        indices = []
        for i, size in enumerate(self.group_sizes):
            if size > 0:
                indice = np.where(self.flag == i)[0]
                indice = indice[list(torch.randperm(int(size), generator=g).numpy())].tolist()
                extra = int(math.ceil(size * 1.0 / self.samples_per_gpu / self.num_replicas)) * \
                    self.samples_per_gpu * self.num_replicas - len(indice)
                tmp = indice.copy()
                for _ in range(extra // size):
                    indice.extend(tmp)
                indice.extend(tmp[:extra % size])
                indices.extend(indice)
        assert len(indices) == self.total_size
        spg = self.samples_per_gpu
        indices = [indices[j] for i in list(torch.randperm(len(indices) // spg, generator=g))
                   for j in range(i * spg, (i + 1) * spg)]
        offset = self.num_samples * self.rank
        return iter(indices[offset:offset + self.num_samples])


def _class_list_density(pool, cls, density):
    """object_loader_base.py:217-238: walk down from `density` to a bucket holding at least two objects of the class,
    then (from the bottom) up; -> (list of (token, observations), bucket index)"""
    per = pool.get(cls, {})
    while len(per.get(BUCKETS[density], [])) <= 1:
        density -= 1
        if density == -1:
            density = 0
            while len(per.get(BUCKETS[density], [])) <= 1:
                density += 1
                if density >= len(BUCKETS):
                    raise ValueError("no bucket holds two objects of class %r" % (cls,))
    return per[BUCKETS[density]], density


def _frame_even(obj, density):
    """object_loader_base.py:201-215: an observation of `obj` in bucket `density`, else the nearest lower, else the
    lowest non-empty one"""
    b = obj["buckets"]
    while len(b.get(BUCKETS[density], [])) == 0:
        density -= 1
        if density == -1:
            density = 0
            while len(b.get(BUCKETS[density], [])) == 0:
                density += 1
                if density >= len(BUCKETS):
                    raise ValueError("object %r has no observation" % obj["token"])
    return np.random.choice(b[BUCKETS[density]], 1, replace=False)[0]


class TrainPairs:
    """table: pcr_amd.pairs.ObjectTable; `read(token, observation) -> float32 [n, 3]` loads one sparse crop (e.g.
    `CropDirectory.read`); `read_dense(token)` the aggregated cloud as the reference's complete loader returns it
    (None: the sparse crop stands in).  Item i is built around true object `idx[i]` -- the objects with a tracked class
    and more than two usable observations, shuffled once under the global generator at construction
    (reidentification_base.py:201-250, `shuffle=False` keeps the table order)."""

    def __init__(self, table, read, subsample_sparse, subsample_dense=0, read_dense=None, ids=None, shuffle=True):
        self.table, self.read, self.read_dense = table, read, read_dense
        self.ns, self.nd = int(subsample_sparse), int(subsample_dense)
        self.idx = table.shuffled_index() if shuffle else np.asarray(table.true_index, dtype=np.int64)
        self.ids = ids if ids is not None else {o["token"]: i for i, o in enumerate(table.objects)}
        self.flag = np.zeros(len(self), dtype=np.uint8)
        nb = len(BUCKETS)
        for o in table.objects:
            dist = np.array([len(o["buckets"].get(b, [])) for b in BUCKETS], dtype=np.float64)
            o["distribution"] = dist / dist.sum() if dist.sum() > 0 else np.full(nb, 1.0 / nb)

    def __len__(self):
        return len(self.idx)

    def _dense(self, tok, fallback):
        return self.read_dense(tok) if self.read_dense is not None else fallback

    def __getitem__(self, i):
        t = self.table
        obj = t.objects[self.idx[i]]
        tok, cls = obj["token"], obj["cls"]
        nums = np.asarray(obj["nums"])
        if np.random.choice([0, 1]) == 1:
            a, b = np.random.choice(nums, 2, replace=False)
            s1, s2 = self.read(tok, int(a)), self.read(tok, int(b))
            d1 = self._dense(tok, s1)
            return self._item(s1, s2, d1, d1, cls, cls, self.ids[tok], self.ids[tok])
        a = np.random.choice(nums, 1, replace=False)[0]
        s1 = self.read(tok, int(a))
        d1 = self._dense(tok, s1)
        density = np.random.choice(np.arange(len(BUCKETS)), p=obj["distribution"])
        use_tp = np.random.choice([0, 1]) == 1
        pool = t.tp if use_tp else t.fp
        cls2 = cls if use_tp else cls + t.num_classes
        cands, density = _class_list_density(pool, cls, density)
        other = tok
        while other == tok:
            other = cands[np.random.choice(len(cands), 1)[0]][0]
        oobj = t.by_token[other]
        if oobj.get("fp"):
            d2, id2 = np.random.randn(self.nd, 3), -1      # (reference: a false positive has no aggregated cloud)
        else:
            d2, id2 = None, self.ids[other]
        b = _frame_even(oobj, density)
        s2 = self.read(other, int(b))
        if d2 is None:
            d2 = self._dense(other, s2)
        return self._item(s1, s2, d1, d2, cls, cls2, self.ids[tok], id2)

    def _item(self, s1, s2, d1, d2, l1, l2, id1, id2, **extra):
        """return_item (reidentification_base.py:427-438): every cloud arrives point-major [n, 3], is turned
        channel-major and resampled by subsamplePC, sparse clouds first.  (With the reference's FakeCompleteLoader,
        which returns zeros of shape (3, n), the same two calls see n 'channels' x 3 'points', cut to three rows and
        resample from three columns: a `read_dense` that returns that shape reproduces it, tests/test_pairs_golden.py.)"""
        cm = lambda p: np.moveaxis(np.asarray(p), 0, 1)      # noqa: E731
        s1, s2 = D.subsample_pc(cm(s1), self.ns), D.subsample_pc(cm(s2), self.ns)
        d1, d2 = D.subsample_pc(cm(d1), self.nd), D.subsample_pc(cm(d2), self.nd)
        out = dict(sparse_1=s1, sparse_2=s2, dense_1=d1, dense_2=d2, label_1=l1, label_2=l2, id_1=id1, id_2=id2)
        out.update(extra)
        return out


class ValPairs:
    """the validation dataset over a pair set of pcr_amd.pairs.build_val_pairs: positives first, then negatives
    (ReIDDatasetNuscenesFPVal.__getitem__, reidentification_nuscenes.py:108-145), items with the `size_*` / `vis_*`
    keys of return_item_size_vis (reidentification_base.py:455-483).  `vis_to_cls_id` maps nuScenes visibility tokens
    1..4 to 0..3; the reference SWAPS the two visibility entries (`vis1, vis2 = DC(to_tensor(v2)), DC(to_tensor(v1))`,
    :470) and so does this."""

    VIS = {1: 0, 2: 1, 3: 2, 4: 3}

    def __init__(self, table, positives, negatives, read, subsample_sparse, subsample_dense=0, read_dense=None, ids=None,
                 visibility=None):
        self.tp = TrainPairs(table, read, subsample_sparse, subsample_dense, read_dense, ids, shuffle=False)
        self.table, self.pairs = table, list(positives) + list(negatives)
        self.visibility = visibility or {}
        self.flag = np.zeros(len(self), dtype=np.uint8)

    def __len__(self):
        return len(self.pairs)

    def __getitem__(self, i):
        p = self.pairs[i]
        t, tp = self.table, self.tp
        s1, s2 = tp.read(p["tok1"], p["o1"]), tp.read(p["tok2"], p["o2"])
        d1 = tp._dense(p["tok1"], s1)
        o2 = t.by_token[p["tok2"]]
        if p["tok2"] == p["tok1"] and p["match"]:
            d2, id2 = d1, tp.ids[p["tok1"]]
        elif o2.get("fp"):
            d2, id2 = np.random.randn(tp.nd, 3), -1
        else:
            d2, id2 = tp._dense(p["tok2"], s2), tp.ids[p["tok2"]]
        v1 = self.VIS.get(self.visibility.get(p["tok1"], {}).get(int(p["o1"]), -1), -1)
        v2 = self.VIS.get(self.visibility.get(p["tok2"], {}).get(int(p["o2"]), -1), -1)
        return tp._item(s1, s2, d1, d2, p["cls1"], p["cls2"], tp.ids[p["tok1"]], id2,
                        size_1=np.asarray(s1).shape[0], size_2=np.asarray(s2).shape[0], vis_1=v2, vis_2=v1)


class CropDirectory:
    """`<root>/<token>/<observation>/pts_xyz.bin` (object_loader_base.py:247-269) -> crops; `table()` scans the tree
    into the object table the pair rules work on (class / false-positive flag from `meta[token]`)"""

    def __init__(self, root, load_fraction=1.0):
        self.root, self.load_fraction = str(root), load_fraction

    def read(self, token, observation):
        return D.load_points(self.root, token, observation, load_fraction=self.load_fraction)

    def table(self, meta, num_classes):
        import os
        from .pairs import ObjectTable
        objs = []
        for tok in sorted(os.listdir(self.root)):
            frames = {}
            for obs in sorted(os.listdir(os.path.join(self.root, tok)), key=int):
                f = os.path.join(self.root, tok, obs, "pts_xyz.bin")
                frames[int(obs)] = int(os.stat(f).st_size // 12)
            m = meta[tok]
            objs.append(dict(token=tok, cls=int(m["cls"]), fp=bool(m.get("fp", False)), frames=frames))
        return ObjectTable(objs, num_classes)


class EpochLoader:
    """the batches one rank sees in one epoch: sampler order, `samples_per_gpu` items per batch, batch i built under
    virtual worker i mod num_workers (num_workers = 0: the caller's global generator, as a worker-less DataLoader)"""

    def __init__(self, dataset, samples_per_gpu, num_replicas=1, rank=0, seed=0, num_workers=0, device="cpu"):
        self.dataset, self.spg, self.rank, self.seed = dataset, int(samples_per_gpu), int(rank), seed
        self.num_workers, self.device = int(num_workers), device
        self.sampler = DistributedGroupSampler(dataset.flag, samples_per_gpu, num_replicas, rank, seed=seed)

    def __len__(self):
        return len(self.sampler) // self.spg

    def epoch(self, epoch):
        self.sampler.set_epoch(epoch)
        order = list(self.sampler)
        states = None
        if self.num_workers > 0:       # fresh workers every epoch, each seeded by worker_init_fn
            states = []
            for w in range(self.num_workers):
                states.append(np.random.RandomState(worker_seed(self.num_workers, self.rank, w, self.seed or 0)).get_state())
        for bi in range(len(self)):
            ids = order[bi * self.spg:(bi + 1) * self.spg]
            if states is not None:
                outer = np.random.get_state()
                np.random.set_state(states[bi % self.num_workers])
            try:
                samples = [self.dataset[i] for i in ids]
            finally:
                if states is not None:
                    states[bi % self.num_workers] = np.random.get_state()
                    np.random.set_state(outer)
            yield D.collate_pairs(samples, device=self.device)


def run_epochs(trainer, loader, epochs, start_epoch=0, on_step=None):
    """mmcv's EpochBasedRunner loop for this path: per epoch the sampler is re-seeded (DistSamplerSeedHook), every batch
    is one Trainer.step (forward + backward + bucket exchange + clip + AdamW); -> list of per-step losses (lazy)"""
    outs = []
    for ep in range(start_epoch, start_epoch + epochs):
        trainer.epoch = ep
        for batch in loader.epoch(ep):
            out = trainer.step(batch)
            outs.append(out["loss"].detach())
            if on_step is not None:
                on_step(ep, out)
        trainer.epoch = ep + 1
    return outs
