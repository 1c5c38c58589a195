"""GPU parity of the PointNet ReIDNet (BASELINE config 1 shape: N=256, batch of pairs) against the golden
vectors recorded from the imported reference, and of stand-alone LinearRes rows."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, load_golden
from pcr_amd import testing as T

pytestmark = pytest.mark.gpu
TOL = 1e-4

PN_MODEL = dict(
    type="ReIDNet", hidden_size=128, pool_type="both", combine="point-cat", match_type="xcorr_eff",
    output_sequence_size=64, use_dgcnn=True, backbone_list=[128, 64, 32],
    backbone=dict(type="PointNet", k=40, normal_channel=False),
    match_head=[dict(type="LinearRes", n_in=128, n_out=128, norm="GN", ng=8),
                dict(type="Linear", in_features=128, out_features=1)],
    cls_head=None, fp_head=None, shape_head=None,
    downsample=[dict(type="LinearRes", n_in=1024, n_out=512, norm="GN", ng=64),
                dict(type="LinearRes", n_in=512, n_out=128, norm="GN", ng=16),
                dict(type="Linear", in_features=128, out_features=64)],
    cross_stage1=dict(type="corss_attention", d_model=64, nhead=2, attention="linear"),
    cross_stage2=dict(type="corss_attention", d_model=64, nhead=2, attention="linear"),
    local_stage1=dict(), local_stage2=dict(),
    losses_to_use=dict(kl=False, match=True, cls=False, shape=False, fp=False, triplet=False))


def build():
    import copy
    from mmdet3d.models import build_model
    m = build_model(copy.deepcopy(PN_MODEL))
    man = T.load_manifest(os.path.join(GOLDEN, "pointnet_manifest.json"))
    assert T.manifest_of(m) == man
    sd = T.seeded_state_dict(man, 0)
    m.load_state_dict(sd, strict=True)
    return m.cuda().eval(), sd


def test_pointnet_pairs_match_reference_golden():
    g = load_golden("pointnet_n256_randn")
    meta = g["meta"]
    m, _ = build()
    s1, s2 = T.synthetic_pairs(meta["pairs"], meta["n"], meta["input_seed"], meta["kind"])
    with torch.no_grad():
        xyz, feat = m.backbone(torch.cat([s1, s2], 0).permute(0, 2, 1).contiguous().cuda(), m.backbone_list)
        xyz1, xyz2, h1, h2 = m.siamese_forward(s1.cuda(), s2.cuda())
        logits = m.match_forward_inference(h1, h2, xyz1, xyz2)
    scale = np.abs(g["enc_max"]).max()
    assert np.abs(feat.max(dim=2)[0].cpu().numpy() - g["enc_max"]).max() < TOL * max(1.0, scale)
    # (the 1024-wide encoder features reach |x| ~ 20: like the max above, their mean is held to 1e-4 OF THAT SCALE -- in the
    # default split-bf16 arithmetic the absolute difference is ~2e-4 there, 1e-5 relative; the embeddings and the logits
    # below are held to the north star's absolute 1e-4)
    assert np.abs(feat.mean(dim=2).cpu().numpy() - g["enc_mean"]).max() < TOL * max(1.0, scale)
    assert np.abs(h1.cpu().numpy() - g["h1"]).max() < TOL and np.abs(h2.cpu().numpy() - g["h2"]).max() < TOL
    assert np.abs(logits.cpu().numpy() - g["logits"]).max() < TOL


def test_linear_res_rows_standalone():
    import model_oracle as MO
    from mmdet3d.models import LinearRes
    for n_in, n_out, ng in ((96, 64, 16), (64, 64, 8)):
        m = LinearRes(n_in, n_out, norm="GN", ng=ng)
        sd = T.seeded_state_dict(T.manifest_of(m), 7)
        m.load_state_dict(sd)
        x = torch.randn(37, n_in)
        p = dict(sd)
        p["__groups__"] = m.norm1.num_groups
        with torch.no_grad():
            want = MO.linear_res(p, x)
        from pcr_amd import engine
        mc = m.cuda().eval()
        with engine.precision("f32"):
            assert float((mc(x.cuda()).cpu() - want).abs().max()) < 2e-5
        with engine.precision("bf16x3"):      # (64 -> 64 rows run on the bf16 matrix core: inside the parity bound)
            assert float((mc(x.cuda()).cpu() - want).abs().max()) < 1e-4


@pytest.mark.parametrize("cin,cout,L,B", [(512, 256, 100, 3), (1024, 512, 77, 2), (768, 256, 64, 1), (512, 512, 33, 2),
                                          (1024, 256, 200, 1)])
def test_chunked_dense_ragged_token_counts(cin, cout, L, B):
    """the chunked-contraction dense kernel (cin >= 512, multiples of 256) on token counts that are not multiples of
    its 64-token tile, plain and with the fused GroupNorm (+ residual, ReLU) epilogue, against torch fp32"""
    import torch.nn.functional as F
    from pcr_amd import engine as E, rows
    g = torch.Generator().manual_seed(cin + L)
    x = torch.randn(B, cin, L, generator=g)
    w = torch.randn(cout, cin, generator=g) / cin ** 0.5
    sc, sh = torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g)
    wp = E.pack_weight(w, "cuda")
    for act in (0, 1, 2):
        want = torch.einsum("oc,bcl->bol", w, x) * sc.view(1, -1, 1) + sh.view(1, -1, 1)
        want = F.relu(want) if act == 1 else F.leaky_relu(want, 0.2) if act == 2 else want
        got = E.dense(x.cuda(), wp, cout, sc.cuda(), sh.cuda(), act=act).cpu()
        assert float((got - want).abs().max()) < 2e-4 * max(1.0, float(want.abs().max())), (act,)
    gn = torch.nn.GroupNorm(cout // 8, cout)
    with torch.no_grad():
        gn.weight.copy_(torch.randn(cout, generator=g))
        gn.bias.copy_(torch.randn(cout, generator=g))
    res = torch.randn(B, cout, L, generator=g)
    with torch.no_grad():
        y = torch.einsum("oc,bcl->bol", w, x)
        want = F.relu(gn(y.permute(0, 2, 1).reshape(-1, cout)).reshape(B, L, cout).permute(0, 2, 1) + res)
    got = rows.dense_gn(x.cuda(), wp, cout, gn, res=res.cuda(), relu=True).cpu()
    assert float((got - want).abs().max()) < 2e-4
    for groups in (cout // 4, cout // 16, cout // 32, 1):           # group sizes 4 / 16 / 32 fused, 1 group unfused
        gn2 = torch.nn.GroupNorm(groups, cout)
        with torch.no_grad():
            want = gn2(y.permute(0, 2, 1).reshape(-1, cout)).reshape(B, L, cout).permute(0, 2, 1)
        got = rows.dense_gn(x.cuda(), wp, cout, gn2).cpu()
        assert float((got - want).abs().max()) < 2e-4, groups


@pytest.mark.parametrize("B,cin,cout,Ln", [(5, 128, 1024, 256), (3, 128, 1024, 100), (2, 64, 256, 37), (70, 128, 512, 64)])
def test_dense_with_fused_max_over_points_equals_the_two_launches(B, cin, cout, Ln):
    """pcr_dense_max_f32 (round 5: STN conv3 + BN + ReLU + max over the points without the (B,cout,L) tensor) against
    pcr_dense_f32 + pcr_max_over_l_f32: bit-equal (a maximum is order-independent), ragged point counts included"""
    import ctypes
    from pcr_amd import _lib as L
    from pcr_amd import engine as E
    lib = L.load()
    assert lib.pcr_dense_max_ok(cin, cout, Ln) == 1 and lib.pcr_dense_max_ok(cin, 96, Ln) == 0
    g = torch.Generator().manual_seed(cin + Ln)
    x = torch.randn(B, cin, Ln, generator=g).cuda()
    w = torch.randn(cout, cin, generator=g) / cin ** 0.5
    sc = (1.0 + 0.2 * torch.randn(cout, generator=g)).cuda()
    sh = (0.3 * torch.randn(cout, generator=g)).cuda()
    wp = E.pack_weight(w, x.device)
    with E.precision("f32"):
        y = E.dense(x, wp, cout, sc, sh, act=1)
    want = torch.empty((1, cout, B), dtype=torch.float32, device="cuda")
    L.check(lib.pcr_max_over_l_f32(L.ptr(y), L.ptr(want), B, cout, Ln, L.stream_ptr()), "pcr_max_over_l_f32")
    got = torch.full((1, cout, B), float("nan"), dtype=torch.float32, device="cuda")
    L.check(lib.pcr_dense_max_f32(L.ptr(x), L.ptr(wp), L.ptr(sc), L.ptr(sh), L.ptr(got), B, cin, cout, Ln, 1,
                                  L.stream_ptr()), "pcr_dense_max_f32")
    assert torch.equal(got, want)
    assert torch.equal(want[0].t(), y.max(dim=2)[0])


@pytest.mark.parametrize("B,cin,cout,Ln", [(3, 1024, 512, 256), (2, 256, 256, 128), (5, 384, 160, 384), (1, 512, 1024, 256)])
@pytest.mark.parametrize("prec", ["bf16x3", "bf16"])
def test_two_role_dense_kernel_equals_the_one_role_kernel_bit_for_bit(B, cin, cout, Ln, prec):
    """dense_bf_pc_kernel (round 5: 128-token tiles, producer waves fetch + split the next 128-channel chunk while the
    consumer waves multiply the current one; token counts that are multiples of 128) against dense_bf_kernel (64-token
    tiles, one role), which the same tokens take when the cloud is cut to a length that is no multiple of 128: every output
    is a function of its own token, both kernels walk the 16-channel steps in the same order -> the same bits.  Plain
    (scale / shift / activation) and GroupNorm + residual + ReLU epilogues, cout windows of 256 and a partial window; and
    both stay on the torch result."""
    import torch.nn.functional as F
    from pcr_amd import engine as E, rows
    g = torch.Generator().manual_seed(cin + cout + Ln)
    x = torch.randn(B, cin, Ln, generator=g).cuda()
    w = torch.randn(cout, cin, generator=g) / cin ** 0.5
    sc, sh = (torch.rand(cout, generator=g) + 0.5).cuda(), torch.randn(cout, generator=g).cuda()
    res = torch.randn(B, cout, Ln, generator=g).cuda()
    gn = torch.nn.GroupNorm(cout // 8, cout)
    with torch.no_grad():
        gn.weight.copy_(torch.randn(cout, generator=g))
        gn.bias.copy_(torch.randn(cout, generator=g))
    cut = Ln - 64                      # not a multiple of 128: the one-role kernel
    with E.precision(prec):
        wp = E.pack_weight_dual(w, "cuda")
        assert getattr(wp, "_pcr_bf", None) is not None
        for act in (0, 1, 2):
            full = E.dense(x, wp, cout, sc, sh, act=act)
            part = E.dense(x[:, :, :cut].contiguous(), wp, cout, sc, sh, act=act)
            assert torch.equal(full[:, :, :cut], part), act
        full = rows.dense_gn(x, wp, cout, gn, res=res, relu=True)
        part = rows.dense_gn(x[:, :, :cut].contiguous(), wp, cout, gn, res=res[:, :, :cut].contiguous(), relu=True)
        assert torch.equal(full[:, :, :cut], part)
    with torch.no_grad():
        y = torch.einsum("oc,bcl->bol", w, x.cpu())
        want = F.relu(gn(y.permute(0, 2, 1).reshape(-1, cout)).reshape(B, Ln, cout).permute(0, 2, 1) + res.cpu())
    tol = 2e-4 if prec == "bf16x3" else 6e-2
    assert float((full.cpu() - want).abs().max()) < tol


@pytest.mark.parametrize("B,cin,cout,Ln", [(5, 128, 1024, 256), (3, 64, 128, 200), (2, 128, 160, 131), (4, 24, 256, 128)])
def test_resident_weight_dense_kernel_equals_the_streaming_kernels_bit_for_bit(B, cin, cout, Ln):
    """dense_rw_kernel (round 5: f32 layers with cin <= 128 keep their weights in registers for all tiles of a cloud and
    fetch the next tile during the MFMAs; clouds of >= 128 tokens) against the kernels that stream the weights per tile,
    which the same tokens take in a cloud cut below 128 tokens: same k order -> same bits, ragged last tiles, cout
    windows that are not multiples of 128, every activation; the fused max over the points equals the max of the stored
    layer; and both stay on torch."""
    import ctypes
    import torch.nn.functional as F
    from pcr_amd import _lib as L
    from pcr_amd import engine as E
    g = torch.Generator().manual_seed(cin + cout + Ln)
    x = torch.randn(B, cin, Ln, generator=g).cuda()
    w = torch.randn(cout, cin, generator=g) / cin ** 0.5
    sc, sh = (torch.rand(cout, generator=g) + 0.5).cuda(), torch.randn(cout, generator=g).cuda()
    wp = E.pack_weight(w, "cuda")
    cut = 96
    with E.precision("f32"):
        for act in (0, 1, 2):
            full = E.dense(x, wp, cout, sc, sh, act=act)
            part = E.dense(x[:, :, :cut].contiguous(), wp, cout, sc, sh, act=act)
            assert torch.equal(full[:, :, :cut], part), act
            want = torch.einsum("oc,bcl->bol", w, x.cpu()) * sc.cpu().view(1, -1, 1) + sh.cpu().view(1, -1, 1)
            want = F.relu(want) if act == 1 else F.leaky_relu(want, 0.2) if act == 2 else want
            assert float((full.cpu() - want).abs().max()) < 2e-5 * max(1.0, float(want.abs().max()))
        lib = L.load()
        if lib.pcr_dense_max_ok(cin, cout, Ln):
            full = E.dense(x, wp, cout, sc, sh, act=1)
            got = torch.full((1, cout, B), float("nan"), dtype=torch.float32, device="cuda")
            L.check(lib.pcr_dense_max_f32(L.ptr(x), L.ptr(wp), L.ptr(sc), L.ptr(sh), L.ptr(got), B, cin, cout, Ln, 1,
                                          L.stream_ptr()), "pcr_dense_max_f32")
            assert torch.equal(got[0].t(), full.max(dim=2)[0])
