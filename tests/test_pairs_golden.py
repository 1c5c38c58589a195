"""CPU: the training pair rule (pcr_amd/loader.py TrainPairs), both validation pair sets (pcr_amd/pairs.py) and the
validation items against what the REFERENCE's own dataset classes return on a toy crop directory
(tests/golden/pairs_toy.npz, recorded by oracle/make_golden.py gen_pairs from ReIDDatasetNuscenesFP / ...FPVal /
...FPValEven over ObjectLoaderSparseBase: reidentification_nuscenes.py:16-249, reidentification_base.py:201-483,
object_loader_base.py:75-245).  The crops are rebuilt here from the same seeds; every comparison is exact."""
import json
import os

import numpy as np
import pytest

from conftest import ROOT
from pcr_amd import loader as LD
from pcr_amd import pairs as PR

GOLD = os.path.join(ROOT, "tests", "golden", "pairs_toy.npz")
CLS = {"vehicle.car": 0, "human.pedestrian.adult": 1}


def crop_points(token, obs, npts):      # oracle/ref_datasets.py crop_points, restated (the oracle is not imported here)
    h = (sum(ord(c) * (i + 1) for i, c in enumerate(token)) * 1009 + int(obs) * 9176 + 12345) % (2 ** 31 - 1)
    return np.random.RandomState(h).randn(int(npts), 3).astype(np.float32)


@pytest.fixture(scope="module")
def world(tmp_path_factory):
    g = np.load(GOLD)
    meta = json.loads(str(g["meta"]))
    objs = json.loads(str(g["objects"]))
    root = str(tmp_path_factory.mktemp("crops"))
    for o in objs:
        for n, npts in o["frames"].items():
            d = os.path.join(root, o["token"], n)
            os.makedirs(d)
            crop_points(o["token"], int(n), npts).tofile(os.path.join(d, "pts_xyz.bin"))
    crops = LD.CropDirectory(root)
    info = {o["token"]: dict(cls=CLS.get(o["class_name"], -1), fp=o["fp"]) for o in objs}
    vis = {o["token"]: {int(k): v for k, v in o["visibility"].items()} for o in objs}

    def table():
        return crops.table(info, num_classes=2)
    return g, meta, objs, crops, table, vis


def test_object_table_matches_the_loader_of_the_reference(world):
    g, meta, objs, crops, table, _ = world
    t = table()
    assert [o["token"] for o in t.objects] == [o["token"] for o in objs]            # obj_tokens order
    for o, ref in zip(t.objects, objs):
        assert o["frames"] == {int(k): v for k, v in ref["frames"].items()}          # point counts from the file sizes
    assert sorted(t.true_index) == sorted(g["train_idx"].tolist())                   # `temp > 2`, tracked class, not FP


def test_training_items_equal_the_reference_item_for_item(world):
    g, meta, objs, crops, table, _ = world
    np.random.seed(meta["seed"])                      # the state the reference's constructor shuffled under
    ds = LD.TrainPairs(table(), crops.read, meta["ns"], meta["nd"], read_dense=lambda tok: np.zeros((3, meta["nd"])))
    assert np.array_equal(ds.idx, g["train_idx"])
    assert [ds.table.objects[i]["cls"] for i in ds.idx] == g["train_classes"].tolist()
    np.random.seed(meta["item_seed"])
    k = 0
    for _ in range(meta["passes"]):
        for i in range(len(ds)):
            it = ds[i]
            for key in ("sparse_1", "sparse_2", "dense_1", "dense_2"):
                assert np.array_equal(np.asarray(it[key], dtype=np.float32), g["train_" + key][k]), (key, k)
            for key in ("label_1", "label_2", "id_1", "id_2"):
                assert int(it[key]) == int(g["train_" + key][k][0]), (key, k)
            k += 1
    assert k == len(g["train_id_1"]) and (g["train_id_2"] == -1).sum() > 5 and (g["train_id_1"] != g["train_id_2"]).sum() > 15
    # same number and kind of draws: the generator ends where the reference's ended
    assert np.array_equal(np.random.randint(0, 2 ** 31 - 1, size=4), g["train_rng_after"])


def _as_arrays(t, pos, neg):
    toks = [o["token"] for o in t.objects]
    p = np.array([[toks.index(x["tok1"]), x["o1"], x["o2"], x["cls1"]] for x in pos], dtype=np.int64)
    n = np.array([[toks.index(x["tok1"]), x["o1"], toks.index(x["tok2"]), x["o2"], x["cls1"], x["cls2"]] for x in neg],
                 dtype=np.int64)
    return p, n


@pytest.mark.parametrize("kind", ["val_even", "val"])
def test_validation_pair_sets_equal_the_reference(world, kind):
    g, meta, objs, crops, table, _ = world
    t = table()
    np.random.seed(meta["seed"])                      # (FPVal runs on the caller's state; FPValEven seeds itself)
    pos, neg = PR.build_val_pairs(t, meta["max_combinations"], seed=meta["seed"], literal_exclusion=True,
                                  even=(kind == "val_even"))
    p, n = _as_arrays(t, pos, neg)
    assert np.array_equal(p, g[kind + "_pos"])
    assert np.array_equal(n, g[kind + "_neg"])
    assert (n[:, 0] == n[:, 2]).sum() > 0             # the reference's literal rule pairs objects with themselves
    # the documented intent (default): never the positive's own object, everything else drawn the same way
    np.random.seed(meta["seed"])
    pos2, neg2 = PR.build_val_pairs(table(), meta["max_combinations"], seed=meta["seed"], even=(kind == "val_even"))
    p2, n2 = _as_arrays(t, pos2, neg2)
    assert np.array_equal(p2, p) and (n2[:, 0] != n2[:, 2]).all()


def test_validation_items_carry_size_and_swapped_visibility(world):
    g, meta, objs, crops, table, vis = world
    t = table()
    np.random.seed(meta["seed"])
    pos, neg = PR.build_val_pairs(t, meta["max_combinations"], literal_exclusion=True, even=False)
    ds = LD.ValPairs(t, pos, neg, crops.read, meta["ns"], meta["nd"], read_dense=lambda tok: np.zeros((3, meta["nd"])),
                     visibility=vis)
    assert len(ds) == 2 * len(pos)
    np.random.seed(3)
    for name, j in (("p", 0), ("n", len(pos))):
        it = ds[j]
        for key in ("sparse_1", "sparse_2", "dense_1", "dense_2"):
            assert np.array_equal(np.asarray(it[key], dtype=np.float32), g["val_item_%s_%s" % (name, key)]), (name, key)
        for key in ("label_1", "label_2", "id_1", "id_2", "size_1", "size_2", "vis_1", "vis_2"):
            assert int(it[key]) == int(g["val_item_%s_%s" % (name, key)][0]), (name, key)
