"""Object table and validation pair-set construction of the ReID datasets, without mmcv / lamtk.

Reference, restated call for call on numpy's GLOBAL generator:
* `ObjectTable` -- `ObjectLoaderSparseBase.get_filtered_nums` in `filter_mode='pts'`, `get_buckets`, `get_all_buckets`
  (mmdet3d/datasets/object_loader_base.py:99-107, 154-199) and `ReIDDatasetBase.collect_dataset_idx`
  (reidentification_base.py:201-250): the observations of an object with at least `min_points` points; objects with
  MORE THAN TWO such observations and a tracked class form `idx` (the reference's comment says "at least two", its
  code says `temp > 2`: the code is followed); false positives need one observation; `shuffle_idx` permutes `idx`.
* `build_val_pairs(even=True)` -- `ReIDDatasetNuscenesFPValEven.{before,after}_collect_dataset_idx_hook`
  (reidentification_nuscenes.py:206-249: `set_seeds(validation_seed)`, the shuffle, up to `max_combinations` shuffled
  2-combinations per object, one negative per positive by `get_random_other_even_val`, reidentification_base.py:361-425:
  same class, true object or false positive by a coin flip, an observation from the point-count bucket of the
  positive's second observation or the nearest lower bucket that can supply one).
* `build_val_pairs(even=False)` -- `ReIDDatasetNuscenesFPVal.after_collect_dataset_idx_hook` (:88-105) with
  `get_random_other` / `get_random_frame` (reidentification_base.py:264-279, object_loader_base.py:149-152).

PARITY: pinned.  `tests/golden/pairs_toy.npz` holds what the reference's own classes produce on a toy object table
(oracle/make_golden.py gen_pairs, oracle/ref_datasets.py: the classes imported unmodified, stand-ins only for the absent
lamtk / mmcv / mmdet imports); `tests/test_pairs_golden.py` holds this module to it element by element.

The reference's exclusion rule, as written: it passes `taken_idx=x['o1']` -- the positive's first OBSERVATION NUMBER --
where an object index is meant (reidentification_nuscenes.py:100, 236), so the object it refuses as a partner is the one
whose index happens to equal that observation number (`self.obj_tokens[taken_idx]`, reidentification_base.py:392,423;
`other == taken_idx`, :276), not the positive's own object: the reference can pair an object with itself as a
"negative" (8-10 of 51 negatives in the fixture).  `literal_exclusion=True` reproduces that (and raises IndexError where
the reference would); the default excludes the positive's own object, which is what the reference's docstrings say.
"""
import itertools

import numpy as np

BUCKETS = [(2 ** x, 2 ** (x + 1)) for x in range(20)]


def bucket_of(num_pts):
    """index of the power-of-two bucket holding `num_pts` (special_log: 0 points -> the LAST bucket, index -1)"""
    return -1 if num_pts == 0 else int(np.log2(num_pts))


class ObjectTable:
    """objects: list of dict(token, cls (int; -1 = not a tracked class), frames {observation number: number of
    points}, fp (bool)), in the order of the reference's `obj_infos` (= `obj_tokens`)"""

    def __init__(self, objects, num_classes, min_points=1, min_observations=3):
        self.objects = list(objects)
        self.num_classes = num_classes
        self.by_token = {o["token"]: o for o in self.objects}
        for o in self.objects:
            # get_filtered_nums, filter_mode 'pts': observation numbers in ascending order with >= min_points points
            o["nums"] = [n for n in sorted(o["frames"], key=int) if o["frames"][n] >= min_points]
            o["buckets"] = {}
            for n in o["nums"]:
                o["buckets"].setdefault(BUCKETS[bucket_of(o["frames"][n])], []).append(n)
        self.true_index = [i for i, o in enumerate(self.objects)
                           if len(o["nums"]) >= min_observations and not o.get("fp") and o["cls"] != -1]
        self.fp_index = [i for i, o in enumerate(self.objects) if len(o["nums"]) > 0 and o.get("fp") and o["cls"] != -1]
        # class -> bucket -> [(token, observations in the bucket)] over ALL objects (the loader's all_buckets)
        self.tp, self.fp = self.pools(range(len(self.objects)))

    def pools(self, index):
        """get_all_buckets over the objects `index` (in that order): (true objects, false positives), each class ->
        bucket -> [(token, number of observations in the bucket)]; objects of untracked classes are skipped"""
        tp, fp = {}, {}
        for i in index:
            o = self.objects[i]
            if o["cls"] == -1:
                continue
            dst = fp if o.get("fp") else tp
            for b, frames in o["buckets"].items():
                dst.setdefault(o["cls"], {}).setdefault(b, []).append((o["token"], len(frames)))
        return tp, fp

    def shuffled_index(self):
        """collect_dataset_idx's `shuffle_idx`: one np.random.permutation over the true-object index"""
        idx = np.asarray(self.true_index, dtype=np.int64)
        return idx[np.random.permutation(len(idx))]


def _excluded_token(table, own_token, o1, literal):
    if not literal:
        return own_token
    return table.objects[o1]["token"]           # (IndexError where the reference raises it)


def _other_even(table, tp_pool, fp_pool, excluded, cls, pts):
    """get_random_other_even_val: -> (token, class label, observation)"""
    b_idx = bucket_of(pts)
    b = BUCKETS[b_idx]
    if np.random.choice([0, 1]) == 1:
        out_cls = cls
        while True:                             # a bucket holding one object cannot supply a partner: one bucket down
            cands = tp_pool.get(cls, {}).get(b)
            if cands is not None and len(cands) != 1:
                break
            b_idx -= 1
            b = BUCKETS[b_idx]                  # (running below -20 raises IndexError, as in the reference)
    else:
        out_cls = cls + table.num_classes
        while True:
            cands = fp_pool.get(cls, {}).get(b)
            if cands is not None:
                break
            b_idx -= 1
            b = BUCKETS[b_idx]
        if len(cands) == 0:
            raise ValueError("no false-positive partner of class %d" % cls)
    other = excluded
    while other == excluded:
        other = cands[np.random.choice(len(cands), 1)[0]][0]
    frame = np.random.choice(table.by_token[other]["buckets"][b], 1)[0]
    return other, out_cls, int(frame)


def build_val_pairs(table, max_combinations, seed=0, literal_exclusion=False, even=True):
    """-> (positives, negatives): lists of dict(tok1, o1, tok2, o2, cls1, cls2, match); len(negatives) == len(positives).
    even=True: the FPValEven rule (seeds numpy with `seed` itself, like set_seeds(validation_seed)); even=False: the
    FPVal rule, which runs on the caller's generator state (`seed` is ignored: the reference's class does not seed).
    literal_exclusion: see the module docstring."""
    if even:
        np.random.seed(seed)
    idx = table.shuffled_index()
    positives = []
    for i in idx:
        o = table.objects[i]
        combs = list(itertools.combinations(o["nums"], r=2))
        np.random.shuffle(combs)
        for a, b in combs[:max_combinations]:
            positives.append(dict(tok1=o["token"], o1=int(a), tok2=o["token"], o2=int(b), cls1=o["cls"], cls2=o["cls"],
                                  pts2=o["frames"][b], match=1))
    negatives = []
    if even:
        tp_pool, _ = table.pools(idx)
        _, fp_pool = table.pools(table.fp_index)
        for p in positives:
            excluded = _excluded_token(table, p["tok1"], p["o1"], literal_exclusion)
            other, cls2, frame = _other_even(table, tp_pool, fp_pool, excluded, p["cls1"], p["pts2"])
            negatives.append(dict(tok1=p["tok1"], o1=p["o1"], tok2=other, o2=frame, cls1=p["cls1"], cls2=cls2, match=0))
        return positives, negatives
    own = {table.objects[i]["token"]: int(i) for i in idx}
    for p in positives:
        same = np.array([i for i in idx if table.objects[i]["cls"] == p["cls1"]])
        if len(same) == 1:
            raise ValueError("class %d has one object: no partner" % p["cls1"])
        taken = p["o1"] if literal_exclusion else own[p["tok1"]]
        other = taken
        while other == taken:
            other = np.random.choice(same, 1)[0]
        oo = table.objects[other]
        frame = np.random.choice(oo["nums"], 1, replace=False)[0]
        negatives.append(dict(tok1=p["tok1"], o1=p["o1"], tok2=oo["token"], o2=int(frame), cls1=p["cls1"],
                              cls2=p["cls1"], match=0))
    return positives, negatives
