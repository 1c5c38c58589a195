"""GPU: tables WITH the coordinate term for the wave-autonomous K-row SA kernel (pcr_dense_pm_xyz_f32, pcr_sa_params.pq_has_xyz,
ABI 17).  The decomposed first layer of sample_and_group_edge's MLP (models/pointnet2_utils.py:242-288, 333-360) is
`relu(Wa (x_i - x_c) + P[i] + Q[c] + shift)`; in the bf16 modes the per-point tables now carry the coordinate term and the
shift -- P'[i] = P[i] + Wa x_i, Q'[c] = Q[c] - Wa x_c + shift, the coordinate products as exact f32 fmas -- and the launch's
first layer is one add per element (no coordinate loads, no layer-1 MFMAs).  The two forms differ by the rounding of
`Wa x_i - Wa x_c` against `Wa (x_i - x_c)`, i.e. by ~1e-7 of the layer's scale: held here against each other, against plain
torch fp32, and the library must refuse such tables for any launch that would not read them as such."""
import ctypes

import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

from pcr_amd import _lib as L
from pcr_amd import engine

pytestmark = pytest.mark.gpu


def _layer(D, c, seed, mode=0):
    g = torch.Generator().manual_seed(seed)
    torch.manual_seed(seed)                      # (the convolutions' default initialisation draws from the global generator)
    cin = 3 + (2 * D if mode == 0 else D)
    convs = [nn.Conv2d(a, b, 1) for a, b in ((cin, c), (c, c), (c, c))]
    bns = [nn.BatchNorm2d(c) for _ in range(3)]
    for bn in bns:
        bn.running_mean.copy_(torch.randn(c, generator=g) * 0.1)
        bn.running_var.copy_(torch.rand(c, generator=g) + 0.5)
        bn.weight.data.copy_(torch.rand(c, generator=g) + 0.5)
        bn.bias.data.copy_(torch.randn(c, generator=g) * 0.1)
        bn.eval()
    return convs, bns, engine.SaPlan(convs, bns, torch.device("cuda"), mode)


def _torch_reference(convs, bns, xyz, feat, idx):
    """sample_and_group_edge + the three conv / BN / ReLU layers + max over K, plain torch fp32 on the host"""
    B, S, K = idx.shape
    D = feat.shape[1]
    li = idx.long().cpu()
    xyz, feat = xyz.cpu(), feat.cpu()
    f_pm = feat.transpose(1, 2)                                             # (B,N,D)
    bidx = torch.arange(B).view(B, 1, 1)
    g_xyz = xyz[bidx, li]                                                   # (B,S,K,3)
    g_f = f_pm[bidx, li]                                                    # (B,S,K,D)
    c_xyz = xyz[:, :S].unsqueeze(2)
    c_f = f_pm[:, :S].unsqueeze(2).expand(-1, -1, K, -1)
    x = torch.cat([g_xyz - c_xyz, c_f, g_f - c_f], dim=-1).permute(0, 3, 1, 2)   # (B, 3 + 2D, S, K)
    for conv, bn in zip(convs, bns):
        x = F.relu(bn(conv(x)))
    return x.max(dim=-1)[0]


def test_the_query_is_shape_only():
    lib = L.load()
    bf = engine.PRECISIONS["bf16x3"]
    assert lib.pcr_sa_tables_take_xyz(0, 32, 64, 64, 64, 48, bf) == 1
    assert lib.pcr_sa_tables_take_xyz(0, 64, 128, 128, 128, 48, bf) == 1
    assert lib.pcr_sa_tables_take_xyz(0, 32, 32, 32, 32, 16, engine.PRECISIONS["bf16"]) == 1
    assert lib.pcr_sa_tables_take_xyz(0, 0, 32, 32, 32, 32, bf) == 0          # no features: no table at all
    assert lib.pcr_sa_tables_take_xyz(1, 32, 64, 64, 64, 48, bf) == 0         # query-and-group layers keep their row tables
    assert lib.pcr_sa_tables_take_xyz(0, 32, 64, 64, 128, 48, bf) == 0        # c3 = 2 c2
    assert lib.pcr_sa_tables_take_xyz(0, 32, 64, 64, 64, 40, bf) == 0         # K not in whole 16-row groups
    assert lib.pcr_sa_tables_take_xyz(0, 32, 64, 64, 64, 48, 0) == 0          # f32: the reference's arithmetic, exact dxyz


@pytest.mark.parametrize("D,c,K,B,N,S", [(32, 64, 48, 9, 1024, 512), (64, 128, 48, 6, 512, 256), (32, 32, 32, 5, 256, 256),
                                         (64, 128, 16, 3, 300, 77), (32, 64, 48, 4, 2048, 1000)])
def test_tables_with_the_coordinate_term_against_the_in_launch_form_and_torch(D, c, K, B, N, S, monkeypatch):
    g = torch.Generator().manual_seed(N + S + K + c)
    xyz = torch.randn(B, N, 3, generator=g).cuda()
    feat = torch.randn(B, D, N, generator=g).cuda()
    convs, bns, plan = _layer(D, c, 11)
    seen = []
    lib = L.load()
    real = lib.pcr_dense_pm_xyz_f32

    class Spy:   # (the path under test must be the one that ran)
        def __getattr__(self, name):
            if name == "pcr_dense_pm_xyz_f32":
                def f(*a):
                    seen.append(1)
                    return real(*a)
                return f
            return getattr(lib, name)
    with engine.precision("bf16x3"), torch.no_grad():
        idx = engine.knn_prefix(xyz, S, K)
        monkeypatch.setattr(engine, "SA_XYZ_TABLES", False)
        a = plan.run(xyz, feat, idx)
        monkeypatch.setattr(engine, "SA_XYZ_TABLES", True)
        monkeypatch.setattr(L, "load", lambda: Spy())
        b = plan.run(xyz, feat, idx)
        b2 = plan.run(xyz, feat, idx)
        monkeypatch.undo()
    assert seen, "the launch did not take the tables with the coordinate term"
    assert torch.isfinite(b).all() and torch.equal(b, b2)
    want = _torch_reference(convs, bns, xyz, feat, idx).detach()
    scale = float(want.abs().max())
    # the two forms: the rounding of Wa x_i - Wa x_c against Wa (x_i - x_c) (~1e-7) as two split-bf16 layers pass it on
    # (measured 3.5-4.5e-6 of the scale over these shapes): half of what either form is allowed against torch
    assert float((a - b).abs().max()) <= 1e-5 * scale
    # each against plain torch fp32 at the bound tests/test_gpu_precision.py holds the split-bf16 SA layer to
    assert float((a.cpu() - want).abs().max()) <= 2e-5 * scale
    assert float((b.cpu() - want).abs().max()) <= 2e-5 * scale


def test_the_table_builder_against_torch():
    lib = L.load()
    g = torch.Generator().manual_seed(5)
    B, D, N, cout = 3, 32, 200, 128
    feat = torch.randn(B, D, N, generator=g).cuda()
    xyz = torch.randn(B, N, 3, generator=g).cuda() * 3.0
    w = torch.randn(cout, D, generator=g) * 0.2
    wxyz = torch.randn(cout, 4, generator=g).cuda()
    y = torch.empty(B, N, cout, device="cuda")
    wp = engine.pack_weight_bf(w, torch.device("cuda"))
    for pm in (0, 1):
        x = feat.transpose(1, 2).contiguous() if pm else feat
        want = (torch.einsum("od,bdn->bno", w.double(), feat.cpu().double()) +
                torch.einsum("oc,bnc->bno", wxyz[:, :3].cpu().double(), xyz.cpu().double()) + wxyz[:, 3].cpu().double())
        for q_rows, q_off in ((N, cout), (128, 64), (64, 32), (0, 96)):
            y.fill_(-7.0)
            L.check(lib.pcr_dense_pm_xyz_f32(L.ptr(x), L.ptr(wp), L.ptr(xyz), L.ptr(wxyz), L.ptr(y), B, D, cout, N, pm,
                                             engine.PRECISIONS["bf16x3"], q_rows, q_off, L.stream_ptr()), "pcr_dense_pm_xyz_f32")
            got = y.cpu().double()
            tol = 1e-5 * float(want.abs().max())
            if q_rows:
                assert float((got[:, :q_rows] - want[:, :q_rows]).abs().max()) <= tol
            if q_rows < N:
                assert float((got[:, q_rows:, :q_off] - want[:, q_rows:, :q_off]).abs().max()) <= tol
                # the couts [q_off, cout) of the tokens behind q_rows: left alone
                assert bool((got[:, q_rows:, q_off:] == -7.0).all())
    # cout not a multiple of four, or only one of the two coordinate operands: refused
    bf = engine.PRECISIONS["bf16x3"]
    assert lib.pcr_dense_pm_xyz_f32(L.ptr(feat), L.ptr(wp), L.ptr(xyz), L.ptr(wxyz), L.ptr(y), B, D, 126, N, 0, bf, N, 124,
                                    L.stream_ptr()) != 0
    assert lib.pcr_dense_pm_xyz_f32(L.ptr(feat), L.ptr(wp), None, L.ptr(wxyz), L.ptr(y), B, D, cout, N, 0, bf, N, cout,
                                    L.stream_ptr()) != 0
    assert lib.pcr_dense_pm_xyz_f32(L.ptr(feat), L.ptr(wp), L.ptr(xyz), L.ptr(wxyz), L.ptr(y), B, D, cout, N, 0, 0, N, cout,
                                    L.stream_ptr()) != 0
    # q_rows not a whole number of 64-token tiles, q_off not a multiple of four
    assert lib.pcr_dense_pm_xyz_f32(L.ptr(feat), L.ptr(wp), L.ptr(xyz), L.ptr(wxyz), L.ptr(y), B, D, cout, N, 0, bf, 100, 64,
                                    L.stream_ptr()) != 0
    assert lib.pcr_dense_pm_xyz_f32(L.ptr(feat), L.ptr(wp), L.ptr(xyz), L.ptr(wxyz), L.ptr(y), B, D, cout, N, 0, bf, 128, 62,
                                    L.stream_ptr()) != 0


@pytest.mark.parametrize("D,cout,B,N", [(32, 128, 5, 256), (64, 256, 3, 512), (32, 256, 2, 64), (64, 128, 70, 128),
                                        (32, 128, 600, 256), (64, 256, 520, 256)])   # (the last two: more tiles than resident workgroups)
def test_the_persistent_form_gives_the_bits_of_the_one_shot_kernel(D, cout, B, N):
    """channel-major input in whole 64-token tiles per cloud runs dense_pm_xyz_res_kernel (persistent workgroups, weight rows
    and the next tile in registers); point-major input of the same values runs the one-shot dense_pm_kernel with the term in
    its store phase.  Same per-tile arithmetic in the same order: the same bits, for every q_rows / q_off, and nothing
    written outside them."""
    lib = L.load()
    g = torch.Generator().manual_seed(D + cout + N)
    feat = torch.randn(B, D, N, generator=g).cuda()
    feat_pm = feat.transpose(1, 2).contiguous()
    xyz = (torch.randn(B, N, 3, generator=g) * 2.0).cuda()
    w = torch.randn(cout, D, generator=g) * 0.2
    wxyz = torch.randn(cout, 4, generator=g).cuda()
    wp = engine.pack_weight_bf(w, torch.device("cuda"))
    for prec in ("bf16x3", "bf16"):
        for q_rows, q_off in ((N, cout), (64 if N > 64 else 0, cout // 2), (0, 32)):
            ys = []
            for x, pm in ((feat, 0), (feat_pm, 1)):
                y = torch.full((B, N, cout), float("nan"), device="cuda")          # (a value no launch can compute)
                L.check(lib.pcr_dense_pm_xyz_f32(L.ptr(x), L.ptr(wp), L.ptr(xyz), L.ptr(wxyz), L.ptr(y), B, D, cout, N, pm,
                                                 engine.PRECISIONS[prec], q_rows, q_off, L.stream_ptr()), "pcr_dense_pm_xyz_f32")
                ys.append(y)
            same = (ys[0] == ys[1]) | (torch.isnan(ys[0]) & torch.isnan(ys[1]))
            assert bool(same.all()), (prec, q_rows, q_off, int((~same).sum()))
            if q_rows < N:
                assert bool(torch.isnan(ys[0][:, q_rows:, q_off:]).all())
            assert not bool(torch.isnan(ys[0][:, :q_rows]).any()) and not bool(torch.isnan(ys[0][:, q_rows:, :q_off]).any())


def test_such_tables_are_refused_by_every_launch_that_would_not_read_them():
    """pq_has_xyz on a launch whose dispatch is not the wave-autonomous K-row kernel must fail, not evaluate Wa dxyz twice"""
    g = torch.Generator().manual_seed(9)
    B, D, c, N, S, K = 2, 32, 64, 256, 128, 48
    xyz = torch.randn(B, N, 3, generator=g).cuda()
    feat = torch.randn(B, D, N, generator=g).cuda()
    _, _, plan = _layer(D, c, 3)
    lib = L.load()
    calls = []
    real = lib.pcr_sa_mlp_f32

    class Forcing:   # sets the flag on a launch the engine would not set it on (f32 precision)
        def __getattr__(self, name):
            if name == "pcr_sa_mlp_f32":
                def f(pref, stream):
                    pref._obj.pq_has_xyz = 1
                    rc = real(pref, stream)
                    calls.append(rc)
                    return 0
                return f
            return getattr(lib, name)
    with engine.precision("f32"), torch.no_grad():
        idx = engine.knn_prefix(xyz, S, K)
        orig = L.load
        try:
            L.load = lambda: Forcing()
            plan.run(xyz, feat, idx)
        finally:
            L.load = orig
    assert calls and calls[0] != 0, "an f32 launch accepted tables with the coordinate term"
    torch.cuda.synchronize()
