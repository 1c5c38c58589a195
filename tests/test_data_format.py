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
