"""Validation pair-set construction of the ReID datasets on a plain object table.

Reference: `ReIDDatasetNuscenesFPVal.after_collect_dataset_idx_hook` (mmdet3d/datasets/reidentification_nuscenes.py:
209-249, seeded by `set_seeds(validation_seed)`, datasets/utils.py:536-541) with `get_random_other_even_val`
(reidentification_base.py:361-425) and the power-of-two point-count buckets of the object loader
(object_loader_base.py:154-171, 241-244).  The rule: every object contributes up to `max_combinations` POSITIVE pairs
(two observations of itself, a shuffled prefix of all 2-combinations); every positive gets ONE NEGATIVE: its first
observation against an observation of another object -- a coin flip decides between a true object of the same class
and a false-positive detection of that class -- drawn from the same point-count bucket as the positive's second
observation (falling to lower buckets while the bucket is empty), so positives and negatives are balanced 50/50 and
matched in sparsity.

PARITY: unpinned.  The reference's dataset classes sit on `lamtk.aggregation.loader.Loader`, which is absent here, so
no fixture can be recorded from them; this module makes the same KIND of numpy global-RNG calls (np.random.seed,
shuffle, choice) in the same order over equivalent tables, which is what makes the sets reproducible under a seed.

One deliberate difference, selectable: the reference passes `taken_idx=x['o1']` -- the positive's first OBSERVATION
NUMBER -- to `get_random_other_even_val` (reidentification_nuscenes.py:236), which then refuses the object
`self.obj_tokens[taken_idx]` (reidentification_base.py:392,423): the object whose INDEX happens to equal that
observation number, not the positive's own object.  So the reference can pair an object with itself as a "negative"
and needlessly excludes an unrelated one.  `build_val_pairs(..., literal_exclusion=False)` (default) excludes the
positive's own object, which is what the docstring of the reference function says it does;
`literal_exclusion=True` reproduces the reference's rule as written (an observation number beyond the object list,
where the reference would raise IndexError, excludes nothing).  Under one seed the two give different negative sets.
"""
import itertools

import numpy as np

BUCKETS = [(2 ** x, 2 ** (x + 1)) for x in range(20)]


def bucket_of(num_pts):
    """index of the power-of-two bucket holding `num_pts` (special_log: 0 points -> the LAST bucket, index -1)"""
    return -1 if num_pts == 0 else int(np.log2(num_pts))


class ObjectTable:
    """objects: list of dict(token, cls (int), frames {observation number: number of points}, fp (bool))"""

    def __init__(self, objects, num_classes):
        self.objects = list(objects)
        self.num_classes = num_classes
        self.by_token = {o["token"]: o for o in self.objects}
        for o in self.objects:
            o["buckets"] = {}
            for n, pts in o["frames"].items():
                o["buckets"].setdefault(BUCKETS[bucket_of(pts)], []).append(n)
        # class -> bucket -> [(token, observations in the bucket)], separately for true objects and false positives
        self.tp, self.fp = {}, {}
        for o in self.objects:
            dst = self.fp if o.get("fp") else self.tp
            for b, frames in o["buckets"].items():
                dst.setdefault(o["cls"], {}).setdefault(b, []).append((o["token"], frames))


def _other_even(table, token, cls, pts):
    """an observation in the bucket of `pts` (or the nearest lower non-empty one) of an object other than `token`
    (None: nothing excluded); -> (token, cls, frame)"""
    b_idx = bucket_of(pts)
    use_tp = np.random.choice([0, 1]) == 1
    pool = table.tp if use_tp else table.fp
    out_cls = cls if use_tp else cls + table.num_classes
    while True:
        cands = pool.get(cls, {}).get(BUCKETS[b_idx])
        # a true-positive bucket holding only the object itself cannot supply a partner: go one bucket down
        if cands and not (use_tp and len(cands) == 1) and not (len(cands) == 1 and cands[0][0] == token):
            break
        b_idx -= 1
        if b_idx < -len(BUCKETS):
            raise ValueError("no %s partner of class %d for an observation of %d points"
                             % ("true-object" if use_tp else "false-positive", cls, pts))
    other = token
    while other == token:
        other = cands[np.random.choice(len(cands), 1)[0]][0]
    frame = np.random.choice(table.by_token[other]["buckets"][BUCKETS[b_idx]], 1)[0]
    return other, out_cls, int(frame)


def build_val_pairs(table, max_combinations, seed=0, literal_exclusion=False):
    """-> (positives, negatives): lists of dict(tok1, o1, tok2, o2, cls1, cls2, match); len(negatives) == len(positives)
    literal_exclusion: see the module docstring"""
    np.random.seed(seed)
    positives = []
    for o in table.objects:
        if o.get("fp"):
            continue
        combs = list(itertools.combinations(sorted(o["frames"]), r=2))
        np.random.shuffle(combs)
        for a, b in combs[:max_combinations]:
            positives.append(dict(tok1=o["token"], o1=int(a), tok2=o["token"], o2=int(b), cls1=o["cls"], cls2=o["cls"],
                                  pts2=o["frames"][b], match=1))
    negatives = []
    for p in positives:
        excluded = p["tok1"]
        if literal_exclusion:
            excluded = table.objects[p["o1"]]["token"] if p["o1"] < len(table.objects) else None
        other, cls2, frame = _other_even(table, excluded, p["cls1"], p["pts2"])
        negatives.append(dict(tok1=p["tok1"], o1=p["o1"], tok2=other, o2=frame, cls1=p["cls1"], cls2=cls2, match=0))
    return positives, negatives
