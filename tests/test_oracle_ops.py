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


# ---- round 6: the tie / boundary rules of the CUDA kernels, pinned by a vector --------------
# tests/golden/ops_cuda_semantics.npz holds the outputs of oracle/cuda_sim.py -- a thread-faithful simulation of the .cu
# kernels' EXECUTION (one object per CUDA thread, the shared arrays, the barrier phases; furthest_point_sample_cuda.cu:
# 17-141,213-331, ball_query_cuda.cu:11-54, knn_cuda.cu:27-94, three_nn_cuda.cu:11-65), written independently of the C
# oracle -- on inputs with exact ties (integer / dyadic lattices, duplicated points), non-power-of-two N, N > 1024, points
# at exactly r, d2 == 0 below min_r, heaps full of equal distances, fewer than three known points.
def _cuda_cases():
    import json
    g = load_golden("ops_cuda_semantics")
    meta = g["meta"] if isinstance(g["meta"], dict) else json.loads(str(g["meta"]))
    return g, meta


def run_cuda_case(name, m, g, fps, fps_dist, ball_query, knn, three_nn):
    """-> list of (got, want) array pairs for one case, through the five callables (C oracle here, HIP ops on the GPU)"""
    k = m["kind"]
    if k == "fps":
        return [(fps(g[name + "_xyz"], m["m"]), g[name + "_idx"])]
    if k == "fps_dist":
        return [(fps_dist(g[name + "_dist"], m["m"]), g[name + "_idx"])]
    if k == "ball":
        return [(ball_query(m["min_r"], m["max_r"], m["k"], g[name + "_xyz"], g[name + "_centres"]), g[name + "_idx"])]
    if k == "knn":
        i, d = knn(m["k"], g[name + "_xyz"], g[name + "_centres"])
        return [(i, g[name + "_idx"]), (d, g[name + "_d2"])]
    d, i = three_nn(g[name + "_unknown"], g[name + "_known"])
    return [(i, g[name + "_idx"]), (d, g[name + "_d2"])]


def test_c_oracle_equals_the_simulated_execution_of_the_cuda_kernels():
    g, meta = _cuda_cases()
    assert {m["kind"] for m in meta.values()} == {"fps", "fps_dist", "ball", "knn", "three_nn"} and len(meta) >= 18
    for name, m in meta.items():
        for got, want in run_cuda_case(name, m, g, P.fps, P.fps_dist, P.ball_query, P.knn, P.three_nn):
            assert got.dtype == want.dtype and np.array_equal(got, want), name
    # the cases really exercise what the Python twins cannot: ties broken away from the lowest index, a block of 1024
    # threads over 3000 points, the strict upper boundary, d2 == 0 below min_r, an infinite third neighbour
    assert g["fps_n12_idx"].tolist() == [[0, 8, 0]] and meta["fps_n3000"]["block"] == 1024 and meta["fps_n100"]["block"] == 64
    xyz, c, row = g["bq_min025_r030_xyz"][0], g["bq_min025_r030_centres"][0], g["bq_min025_r030_idx"][0, 0]
    assert np.array_equal(xyz[0], xyz[64]) and np.array_equal(c[0], xyz[0]) and {0, 64} <= set(row.tolist())   # d2 == 0 < min_r^2
    at_min = [k for k in row.tolist() if abs(np.linalg.norm(xyz[k] - c[0]) - 0.25) < 1e-7]
    assert at_min and len(set(row.tolist())) == 2 + len(set(at_min))     # d2 == min_r^2 is inside, nothing else is
    assert (g["bq_r050_idx"][0, 10] == 0).all()                          # the far centre: nothing in range, zeros
    d = np.linalg.norm(g["bq_r050_xyz"][0][:, None] - g["bq_r050_centres"][0][None], axis=-1)
    assert (d == 0.5).sum() > 10                                         # points at exactly r exist (and are excluded)
    assert np.isinf(g["nn3_two_known_d2"][..., 2]).all() and (g["nn3_two_known_idx"][..., 2] == 0).all()
    tied = g["knn_k8_d2"][0]
    assert (np.diff(tied, axis=-1) == 0).sum() > 20                      # equal distances inside the returned heaps


def test_the_vectors_are_what_the_simulator_produces():
    """the committed file is reproducible from oracle/cuda_sim.py (cheap cases re-simulated here), and the simulator's
    launch shape follows the launcher's opt_n_threads"""
    import cuda_sim as S
    g, meta = _cuda_cases()
    for n, want in [(1, 1), (3, 2), (12, 8), (100, 64), (1000, 512), (1024, 1024), (3000, 1024)]:
        assert S.launch_block_size(n) == want
    for name in ("fps_n12", "fps_n12_dup", "fps_n100", "fpsd_n40_asym", "bq_min050_r075", "knn_k8", "nn3_lattice"):
        for got, want in run_cuda_case(name, meta[name], g, S.fps, S.fps_with_dist, S.ball_query, S.knn, S.three_nn):
            assert np.array_equal(got, want), name
