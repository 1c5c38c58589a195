"""CPU: the C oracle (oracle/pcr_oracle.c) against the golden vectors of the reference's Python
twins (tests/golden/ops_python_twins.npz) and against brute-force numpy definitions."""
import numpy as np
import pytest

from conftest import load_golden
from pcr_amd import testing as T
import point_ops as P


def _clouds(n_clouds, n, seed, kind="randn"):
    return T.synthetic_clouds(n_clouds, n, seed, kind).numpy()


def _take(xyz, idx):
    return np.take_along_axis(xyz, idx[..., None].astype(np.int64).repeat(3, -1), 1)


def test_against_python_twins():
    g = load_golden("ops_python_twins")
    m = g["meta"]
    xyz = _clouds(m["clouds"], m["n"], m["seed"], m["kind"])
    f = P.fps(xyz, m["m"])
    assert (f == g["fps"]).all()
    c = _take(xyz, f)
    assert (P.ball_query(0.0, m["radius"], m["nsample"], xyz, c) == g["ball"]).all()
    k, d2 = P.knn(m["nsample"], xyz, c)
    assert (np.sort(k, -1) == g["knn_sorted"]).all()
    assert (np.diff(d2, axis=-1) >= 0).all()           # heap-sorted ascending


def test_fps_block_rule():
    # largest power of two <= n, capped at 1024 (furthest_point_sample_cuda.cu:11-15)
    for n, want in [(1, 1), (2, 2), (3, 2), (100, 64), (128, 128), (1000, 512), (1024, 1024), (4096, 1024), (5000, 1024)]:
        assert P.fps_block(n) == want


def test_fps_tie_rule_lower_tid_then_lower_k():
    # furthest_point_sample_cuda.cu:17-23,56-71: ties go to the smallest tid (= k mod block), and
    # inside one tid to the smallest k -- NOT to the smallest index.
    n = 12                                            # block = 8: tid t scans k = t, t+8
    xyz = np.zeros((1, n, 3), np.float32)
    xyz[0, 1:, 0] = 1.0                               # point 0 at the origin, all others at x=1
    idx = P.fps(xyz, 3)
    # step 1: every k>=1 ties at distance 1; tid 0 holds k=8 (its k=0 has distance 0) and beats
    # tid 1 (k=1) in the merge tree.  step 2: everything is at distance 0 -> tid 0, k=0.
    assert idx[0].tolist() == [0, 8, 0]


def test_ball_query_predicate_and_padding():
    xyz = np.array([[[0, 0, 0], [0.5, 0, 0], [1.0, 0, 0], [0.2, 0, 0], [3, 0, 0]]], np.float32)
    c = np.array([[[0, 0, 0], [10, 0, 0]]], np.float32)
    idx = P.ball_query(0.0, 1.0, 4, xyz, c)
    # d2 < max_r^2 is strict: the point at distance exactly 1.0 is excluded; pad with first hit
    assert idx[0, 0].tolist() == [0, 1, 3, 0]
    assert idx[0, 1].tolist() == [0, 0, 0, 0]         # nothing in range: stays zero
    idx = P.ball_query(0.3, 1.0, 3, xyz, c)           # min radius excludes 0.2 but d2==0 always passes
    assert idx[0, 0].tolist() == [0, 1, 0]


def test_knn_limits():
    xyz = _clouds(1, 64, 3)
    with pytest.raises(ValueError):
        P.knn(101, xyz, xyz)
    idx, d2 = P.knn(64, xyz, xyz[:, :4])
    assert sorted(idx[0, 0].tolist()) == list(range(64))


def test_knn_prefix_is_sorted_by_distance_then_index():
    xyz = _clouds(2, 96, 11, "dup")
    idx = P.knn_prefix(xyz, 40, 16)
    for b in range(2):
        for s in range(40):
            d = xyz[b] - xyz[b, s]
            d2 = (d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2]     # float32, unfused
            want = np.lexsort((np.arange(96), d2))[:16]
            assert (idx[b, s] == want).all()


def test_gather_group_interp_roundtrip():
    g = np.random.default_rng(0)
    feat = g.standard_normal((2, 5, 33)).astype(np.float32)
    idx = g.integers(0, 33, (2, 7)).astype(np.int32)
    out = P.gather_fwd(feat, idx)
    assert (out == np.take_along_axis(feat, idx[:, None].astype(np.int64).repeat(5, 1), 2)).all()
    gi = g.integers(0, 33, (2, 7, 4)).astype(np.int32)
    go = P.group_fwd(feat, gi)
    assert go.shape == (2, 5, 7, 4)
    assert (go[1, 3, 2, 1] == feat[1, 3, gi[1, 2, 1]])
    # backward = transpose of forward: <fwd(f), g> == <f, bwd(g)>
    gr = g.standard_normal(go.shape).astype(np.float32)
    lhs = (go.astype(np.float64) * gr).sum()
    rhs = (feat.astype(np.float64) * P.group_bwd(gr, gi, 33)).sum()
    assert abs(lhs - rhs) < 1e-3
    gr = g.standard_normal(out.shape).astype(np.float32)
    assert abs((out.astype(np.float64) * gr).sum() - (feat.astype(np.float64) * P.gather_bwd(gr, idx, 33)).sum()) < 1e-3
    # three_nn / interpolate
    unk = g.standard_normal((2, 9, 3)).astype(np.float32)
    kn = g.standard_normal((2, 6, 3)).astype(np.float32)
    d2, i3 = P.three_nn(unk, kn)
    full = ((unk[:, :, None] - kn[:, None]) ** 2).sum(-1)
    assert (i3 == np.argsort(full, -1, kind="stable")[:, :, :3]).all()
    w = g.uniform(size=(2, 9, 3)).astype(np.float32)
    f2 = g.standard_normal((2, 4, 6)).astype(np.float32)
    o = P.three_interp_fwd(f2, i3, w)
    assert np.allclose(o[0, 1, 2], (w[0, 2] * f2[0, 1, i3[0, 2]]).sum(), atol=1e-6)
    gr = g.standard_normal(o.shape).astype(np.float32)
    assert abs((o.astype(np.float64) * gr).sum() - (f2.astype(np.float64) * P.three_interp_bwd(gr, i3, w, 6)).sum()) < 1e-3
