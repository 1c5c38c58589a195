"""GPU: D-FPS + ball query of the picked centres as ONE launch (pcr_fps_ball_query_rows_f32, round 5) against the two
separate launches it replaces (pcr_fps_f32 -- itself bit-exact against the C oracle of furthest_point_sample_cuda.cu --
and pcr_ball_query_rows_f32): pick order, centre coordinates, hit counts and the SA kernel's row table entry for entry,
on random, duplicate-heavy and integer-lattice clouds (ties in the sampling AND on the ball's boundary), ragged sizes;
and the module takes the fused path with bit-identical logits."""
import numpy as np
import pytest
import torch

from pcr_amd import testing as T

pytestmark = pytest.mark.gpu


def _clouds(kind, B, N, seed):
    g = torch.Generator().manual_seed(seed)
    if kind == "lattice":      # integer lattice: equal distances everywhere (FPS tie rule, points ON the ball's surface)
        return torch.randint(0, 5, (B, N, 3), generator=g).float() * 0.25
    return T.synthetic_clouds(B, N, seed=seed, kind=kind)


def _used_rows(cnt, K):
    """per (cloud, item of 16 centres): rows the table defines = sum ceil2(max(cnt,1)) rounded up to 32"""
    B, M = cnt.shape
    nit = (M + 15) // 16
    c = np.maximum(cnt, 1)
    c = (c + 1) // 2 * 2
    pad = np.zeros((B, nit * 16), dtype=np.int64)
    pad[:, :M] = c
    tot = pad.reshape(B, nit, 16).sum(axis=2)
    return (tot + 31) // 32 * 32


@pytest.mark.parametrize("kind", ["box", "randn", "dup", "lattice"])
@pytest.mark.parametrize("B,N,M,K,r", [(5, 1024, 512, 32, 0.2), (3, 512, 128, 64, 0.4), (4, 300, 75, 16, 0.3),
                                       (2, 100, 17, 32, 0.5), (3, 1024, 100, 2, 0.15), (130, 256, 64, 32, 0.3)])
def test_fused_sampling_and_query_equal_the_separate_launches(kind, B, N, M, K, r):
    from mmdet3d import ops
    from mmdet3d.ops import point_ops as PO
    assert PO.fps_ball_query_rows_ok(N, M, K)
    xyz = _clouds(kind, B, N, seed=N + M).cuda().contiguous()
    idx, new_xyz, cnt, rows = PO.fps_ball_query_rows(xyz, M, r, K)
    want_idx = ops.furthest_point_sample(xyz, M)
    assert torch.equal(idx, want_idx)
    want_xyz = ops.gather_points(xyz.transpose(1, 2).contiguous(), want_idx).transpose(1, 2).contiguous()
    assert torch.equal(new_xyz, want_xyz)
    _, want_cnt, want_rows = PO.ball_query_rows(r, K, xyz, want_xyz)
    assert torch.equal(cnt, want_cnt)
    used = _used_rows(cnt.cpu().numpy(), K)
    nit = (M + 15) // 16
    a = rows.view(B, nit, 16 * K, 4).cpu().numpy().view(np.uint32)
    b = want_rows.view(B, nit, 16 * K, 4).cpu().numpy().view(np.uint32)
    for bi in range(B):
        for it in range(nit):
            u = int(used[bi, it])
            assert np.array_equal(a[bi, it, :u], b[bi, it, :u]), (bi, it)
    # the running minimum distances are not an output of the op; the sampling state is checked through the pick order


@pytest.mark.parametrize("prec", ["bf16x3", "bf16"])
def test_the_module_takes_the_fused_path_with_identical_logits(prec):
    import bench
    from pcr_amd import engine
    from mmdet3d.ops import pointnet_modules as PM
    model, _ = bench.build_model("ssg", None)
    s1, s2 = T.synthetic_pairs(6, 1024, seed=11, kind="dup")
    names = []
    engine.PROFILE = []
    with torch.no_grad(), engine.precision(prec):
        a = bench.hot_path(model, s1.cuda(), s2.cuda())
        names = [r[0].split("[")[0] for r in engine.PROFILE]
        engine.PROFILE = None
        prev, PM._NO_FPS_BQ = PM._NO_FPS_BQ, True
        try:
            b = bench.hot_path(model, s1.cuda(), s2.cuda())
        finally:
            PM._NO_FPS_BQ = prev
    # SA1 (64/64/128: the wave-autonomous ragged kernel reads the row table) takes the fused launch; SA2 (128/128/256: the
    # cout-split kernel builds its tiles from index tensor + counts) keeps the separate ones
    assert names.count("fps_ball_query") == 1 and names.count("fps") == 1 and names.count("ball_query") == 1, names
    assert torch.equal(a, b)
    with torch.no_grad(), engine.precision("f32"):       # the f32 SA kernels read the index tensor: separate launches
        engine.PROFILE = []
        bench.hot_path(model, s1.cuda(), s2.cuda())
        names = [r[0].split("[")[0] for r in engine.PROFILE]
        engine.PROFILE = None
    assert "fps_ball_query" not in names and names.count("fps") == 2
