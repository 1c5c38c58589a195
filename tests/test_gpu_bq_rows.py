"""GPU: the ball query's compact row table (pcr_ball_query_rows_f32, include/pcr.h) and the wave-autonomous ragged SA
kernel that reads it (sa_stream_rag_kernel<.., TAB>).  Reference semantics: ball_query_cuda.cu:11-54 (first K indices in
ascending order, padded with the first hit) and point_sa_module.py:166-216 (group, 3-layer shared MLP, max over K).
* idx / cnt written next to the table equal pcr_ball_query_cnt_f32's, bit for bit (and both equal the C oracle elsewhere:
  test_gpu_point_ops.py);
* the table holds exactly the rows {neighbour, point - centre} of the first ceil2(max(cnt, 1)) entries of every idx row,
  centre after centre per run of 16 centres, zero entries up to a multiple of 32 rows;
* an SA launch fed with the table returns the bits of the launch fed with idx + cnt, which returns the bits of the
  K-row launch -- over cloud sizes (4 / 8 / 16 points per lane), empty balls, full balls, centre counts that are not
  multiples of 16, layers with and without a feature table."""
import numpy as np
import pytest
import torch
import torch.nn as nn

pytestmark = pytest.mark.gpu


def _layer(cin, couts, g):
    convs, bns, a = [], [], cin
    for c in couts:
        convs.append(nn.Conv2d(a, c, 1))
        bn = nn.BatchNorm2d(c)
        bn.running_mean.copy_(torch.randn(c, generator=g) * 0.1)
        bn.running_var.copy_(torch.rand(c, generator=g) + 0.5)
        bn.weight.data.copy_(torch.rand(c, generator=g) + 0.5)
        bn.bias.data.copy_(torch.randn(c, generator=g) * 0.1)
        bn.eval()
        bns.append(bn)
        a = c
    return convs, bns


def _table_from_idx(xyz, centres, idx, cnt, K):
    """the table as include/pcr.h words it, from idx / cnt (numpy, f32 subtraction like the kernel's)"""
    B, M, _ = centres.shape
    nitem = (M + 15) // 16
    tab = np.zeros((B, nitem, 16 * K, 4), dtype=np.float32)
    written = np.zeros((B, nitem, 16 * K), dtype=bool)
    for b in range(B):
        for it in range(nitem):
            r = 0
            for c in range(it * 16, min(M, it * 16 + 16)):
                n = max(int(cnt[b, c]), 1)
                n = (n + 1) & ~1
                for k in range(n):
                    i = int(idx[b, c, k])
                    tab[b, it, r, 0] = np.int32(i).view(np.float32)
                    tab[b, it, r, 1:] = xyz[b, i] - centres[b, c]
                    written[b, it, r] = True
                    r += 1
            pad = (r + 31) & ~31
            written[b, it, r:pad] = True    # zero entries
    return tab, written


@pytest.mark.parametrize("B,N,M,K,radius", [(3, 1024, 512, 32, 0.2), (9, 512, 128, 64, 0.4), (2, 200, 37, 16, 0.3),
                                            (1, 1000, 100, 32, 1e-4), (2, 64, 16, 32, 10.0), (17, 700, 50, 48, 0.25),
                                            (96, 1024, 512, 32, 0.08)])   # (50 k centres: rare hazards show at this size)
def test_row_table_matches_the_index_tensor(B, N, M, K, radius):
    from mmdet3d.ops.point_ops import ball_query_cnt, ball_query_rows
    from pcr_amd import _lib as L
    g = torch.Generator().manual_seed(B * 1000 + N + K)
    xyz = torch.rand(B, N, 3, generator=g)
    pick = torch.stack([torch.randperm(N, generator=g)[:M] for _ in range(B)])
    centres = torch.gather(xyz, 1, pick.unsqueeze(-1).expand(-1, -1, 3)).contiguous()
    idx0, cnt0 = ball_query_cnt(0.0, radius, K, xyz.cuda(), centres.cuda())
    idx1, cnt1, rows = ball_query_rows(radius, K, xyz.cuda(), centres.cuda(), want_idx=True)
    none_idx, cnt2, rows2 = ball_query_rows(radius, K, xyz.cuda(), centres.cuda())
    assert none_idx is None
    assert torch.equal(idx0, idx1) and torch.equal(cnt0, cnt1) and torch.equal(cnt0, cnt2)
    assert rows.numel() == L.load().pcr_ball_query_rows_floats(B, M, K) == B * ((M + 15) // 16) * 16 * K * 4
    want, written = _table_from_idx(xyz.numpy(), centres.numpy(), idx0.cpu().numpy(), cnt0.cpu().numpy(), K)
    for got in (rows, rows2):
        got = got.cpu().numpy().reshape(want.shape)
        assert np.array_equal(got.view(np.int32)[written], want.view(np.int32)[written])
    for _ in range(3):                       # the same launch again: the same bits
        again = ball_query_rows(radius, K, xyz.cuda(), centres.cuda())[2].cpu().numpy().reshape(want.shape)
        assert np.array_equal(again.view(np.int32)[written], want.view(np.int32)[written])
    if radius < 1e-3:
        assert int(cnt0.max()) == 1          # a centre is its own (only) neighbour
    if radius > 5:
        assert int(cnt0.min()) == K


@pytest.mark.parametrize("couts,D,B,N,M,K,radius", [((64, 64, 128), 0, 6, 1024, 512, 32, 0.2),
                                                    ((64, 64, 128), 0, 3, 1024, 100, 32, 0.2),
                                                    ((32, 32, 32), 0, 9, 300, 77, 16, 0.3),
                                                    ((64, 64, 64), 16, 4, 512, 64, 32, 0.25),
                                                    ((32, 32, 32), 8, 2, 256, 33, 16, 1e-4),
                                                    ((64, 64, 128), 0, 2, 128, 48, 32, 10.0)])
def test_sa_launch_from_the_row_table_equals_the_indexed_launch(couts, D, B, N, M, K, radius):
    from mmdet3d.ops.point_ops import ball_query_cnt, ball_query_rows
    from pcr_amd import engine
    g = torch.Generator().manual_seed(B * 100 + N + K + D)
    xyz = torch.rand(B, N, 3, generator=g).cuda()
    feat = torch.randn(B, D, N, generator=g).cuda() if D else None
    cidx = torch.stack([torch.randperm(N, generator=g)[:M] for _ in range(B)]).int().cuda()
    centres = torch.gather(xyz, 1, cidx.long().unsqueeze(-1).expand(-1, -1, 3)).contiguous()
    convs, bns = _layer(3 + D, couts, g)
    plan = engine.SaPlan(convs, bns, torch.device("cuda"), 1)
    idx, cnt = ball_query_cnt(0.0, radius, K, xyz, centres)
    _, cnt_r, rows = ball_query_rows(radius, K, xyz, centres)
    for prec in ("bf16x3", "bf16"):
        with engine.precision(prec), torch.no_grad():
            assert plan.wants_row_table(N, K, 0.0)
            a = plan.run(xyz, feat, idx, centre_idx=cidx, cnt=cnt)
            b = plan.run(xyz, feat, None, centre_idx=cidx, cnt=cnt_r, rows=rows, K=K)
            b_pm = plan.run(xyz, feat, None, centre_idx=cidx, cnt=cnt_r, rows=rows, K=K, out_point_major=True)
            krow = plan.run(xyz, feat, idx, centre_idx=cidx)
        assert torch.equal(a, b) and torch.equal(a, krow) and torch.equal(b_pm.contiguous(), a), prec
    with engine.precision("f32"):
        assert not plan.wants_row_table(N, K, 0.0)      # the f32 unit has no wave-autonomous kernels


def test_module_forward_takes_the_row_table_path():
    """PointSAModule (SSG's first layer) through the table equals the module with skip_repeats off (K-row launch)"""
    from mmdet3d.ops import PointSAModule
    from pcr_amd import engine
    torch.manual_seed(3)
    m = PointSAModule(mlp_channels=[0, 64, 64, 128], num_point=128, radius=0.2, num_sample=32).cuda().eval()
    for mod in m.modules():
        if isinstance(mod, nn.BatchNorm2d):
            mod.running_mean.normal_(0, 0.1)
            mod.running_var.uniform_(0.5, 1.5)
    xyz = torch.rand(5, 1024, 3).cuda()
    seen = []
    real = engine.SaPlan.run

    def spy(self, *a, **kw):
        seen.append(kw.get("rows") is not None)
        return real(self, *a, **kw)
    engine.SaPlan.run = spy
    try:
        with torch.no_grad(), engine.precision("bf16x3"):
            _, f1, i1 = m(xyz)
            m.skip_repeats = False
            _, f2, i2 = m(xyz)
    finally:
        engine.SaPlan.run = real
        m.skip_repeats = True
    assert seen == [True, False]
    assert torch.equal(i1, i2) and torch.equal(f1.contiguous(), f2.contiguous())
