"""CPU: the reference's on-disk object-crop format and its resampling (pcr_amd/data.py)."""
import numpy as np
import torch

from pcr_amd import data


def test_load_points_and_fraction(tmp_path):
    pts = np.arange(30, dtype=np.float32).reshape(10, 3)
    d = tmp_path / "objs" / "veh_7" / "3"
    d.mkdir(parents=True)
    pts.tofile(str(d / "pts_xyz.bin"))
    got = data.load_points(str(tmp_path), "objs/veh_7", 3)
    assert got.dtype == np.float32 and np.array_equal(got, pts)
    got = data.load_points(str(tmp_path), "objs/veh_7", "3", load_fraction=0.3)      # the last int(10 * 0.3) = 3 points
    assert np.array_equal(got, pts[7:])


def test_subsample_semantics():
    rng = np.random.RandomState(0)
    pc = np.arange(5 * 40, dtype=np.float64).reshape(5, 40)           # 5 channels, 40 points
    out = data.subsample_pc(pc, 16, rng)
    assert out.shape == (16, 3)
    cols = {tuple(c) for c in pc[:3].T.tolist()}
    assert all(tuple(r) in cols for r in out.tolist())               # every row is one of the input points
    assert data.subsample_pc(pc[:3, :16], 16, rng).tolist() == pc[:3, :16].T.tolist()   # exact size: untouched
    assert np.array_equal(data.subsample_pc(pc[:, :2], 8, rng), np.zeros((8, 3)))       # <= 2 points: zeros
    assert data.subsample_pc(pc, 0).shape == (40, 5)
    big = data.subsample_pc(pc[:3, :4], 64, rng)                                          # upsampling: duplicates
    assert len({tuple(r) for r in big.tolist()}) <= 4


def test_collate_pairs_keys():
    s = [dict(sparse_1=np.zeros((8, 3)), sparse_2=np.ones((8, 3)), id_1=4, id_2=4),
         dict(sparse_1=np.zeros((8, 3)), sparse_2=np.ones((8, 3)), id_1=4, id_2=9)]
    d = data.collate_pairs(s)
    assert set(d) == {"sparse_1", "sparse_2", "dense_1", "dense_2", "label_1", "label_2", "id_1", "id_2"}
    assert len(d["sparse_1"]) == 2 and d["sparse_1"][0].shape == (8, 3) and d["sparse_1"][0].dtype == torch.float32
    assert [int(i) for i in d["id_2"]] == [4, 9] and d["id_1"][0].shape == (1,)


def _object_table(seed=0):
    import numpy as np
    from pcr_amd.pairs import ObjectTable
    g = np.random.default_rng(seed)
    objs = []
    for i in range(40):
        fp = i >= 30
        objs.append(dict(token=("FP_%d" if fp else "obj_%d") % i, cls=int(i % 3), fp=fp,
                         frames={int(n): int(g.integers(1, 600)) for n in range(int(g.integers(3, 9)))}))
    return ObjectTable(objs, num_classes=3)


def test_val_pair_set_construction():
    """the reference's validation pair rule (reidentification_nuscenes.py:209-249) on a synthetic object table"""
    from pcr_amd.pairs import BUCKETS, bucket_of, build_val_pairs
    t = _object_table()
    pos, neg = build_val_pairs(t, max_combinations=4, seed=0)
    assert len(pos) == len(neg) and len(pos) == sum(min(4, len(o["frames"]) * (len(o["frames"]) - 1) // 2)
                                                   for o in t.objects if not o["fp"])      # balanced 50 / 50
    for p, n in zip(pos, neg):
        assert p["tok1"] == p["tok2"] and p["o1"] != p["o2"] and p["match"] == 1
        assert n["tok1"] == p["tok1"] and n["o1"] == p["o1"] and n["tok2"] != n["tok1"] and n["match"] == 0
        other = t.by_token[n["tok2"]]
        assert other["cls"] == p["cls1"]                                  # same class (true object or FP of that class)
        assert n["cls2"] == p["cls1"] + (3 if other["fp"] else 0)
        assert n["o2"] in other["frames"]
    # the partner observation comes from the positive's point-count bucket or the nearest non-empty one below (like the
    # reference, an index running below bucket 0 wraps around to the top of the list: rare)
    same_or_lower = [bucket_of(t.by_token[n["tok2"]]["frames"][n["o2"]]) <= bucket_of(p["pts2"]) for p, n in zip(pos, neg)]
    assert sum(same_or_lower) > 0.8 * len(pos)
    for p, n in zip(pos, neg):
        pass
    again = build_val_pairs(_object_table(), max_combinations=4, seed=0)
    assert again == (pos, neg)                                            # reproducible under the seed
    other_seed = build_val_pairs(_object_table(), max_combinations=4, seed=1)
    assert other_seed != (pos, neg)
    assert BUCKETS[bucket_of(1)] == (1, 2) and BUCKETS[bucket_of(599)] == (512, 1024)
