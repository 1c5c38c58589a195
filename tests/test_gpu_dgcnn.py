"""GPU parity of the DGCNN path: feature-space kNN bit-exact against the C oracle, the
EdgeConv tail against the reference's materialised-edge formulation, and the whole DGCNN ReIDNet against the golden
vectors recorded from the imported reference (oracle/make_golden.py:gen_dgcnn)."""
import copy
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import GOLDEN, load_golden
from pcr_amd import testing as T

pytestmark = pytest.mark.gpu
TOL = 1e-4

DG_MODEL = dict(
    type="ReIDNet", hidden_size=128, pool_type="both", combine="point-cat", match_type="xcorr_eff",
    output_sequence_size=64, use_dgcnn=True, backbone_list=[128, 64, 32],
    backbone=dict(type="dgcnn", dropout=0.5, emb_dims=1024, k=20, output_channels=40),
    match_head=[dict(type="LinearRes", n_in=128, n_out=128, norm="GN", ng=16),
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
    from mmdet3d.models import build_model
    m = build_model(copy.deepcopy(DG_MODEL))
    man = T.load_manifest(os.path.join(GOLDEN, "dgcnn_manifest.json"))
    assert T.manifest_of(m) == man          # same state_dict names (incl. the doubly registered BatchNorms)
    sd = T.seeded_state_dict(man, 0)
    m.load_state_dict(sd, strict=True)
    return m.cuda().eval(), sd


@pytest.mark.parametrize("B,C,N,K", [(3, 3, 256, 20), (2, 64, 256, 20), (2, 128, 300, 20), (2, 64, 1024, 20),
                                     (2, 3, 64, 20), (3, 5, 37, 37), (1, 64, 2048, 20), (2, 128, 1024, 48),
                                     (2, 7, 1000, 1)])
def test_knn_feat_bit_exact(B, C, N, K):
    import point_ops as PO
    from pcr_amd import dgcnn_engine as DE
    g = torch.Generator().manual_seed(100 + N + C)
    x = torch.randn(B, C, N, generator=g)
    want = PO.knn_feat(x.numpy(), K)
    got = DE.knn_feat(x.cuda(), K).cpu().numpy()
    assert (got == want).all(), "rows differing: %d of %d" % ((got != want).any(-1).sum(), B * N)


def test_knn_feat_ties_and_duplicates():
    """duplicated points give exact distance ties: the lower index wins, in the oracle and on the GPU"""
    import point_ops as PO
    from pcr_amd import dgcnn_engine as DE
    g = torch.Generator().manual_seed(5)
    base = torch.randn(2, 16, 40, generator=g)
    x = base[:, :, torch.randint(0, 40, (256,), generator=g)].contiguous()     # every point ~6 times
    want = PO.knn_feat(x.numpy(), 20)
    got = DE.knn_feat(x.cuda(), 20).cpu().numpy()
    assert (got == want).all()
    same = torch.ones(1, 8, 512)                                                # every distance equal
    got = DE.knn_feat(same.cuda(), 20).cpu().numpy()
    assert (got == np.arange(20)[None, None, :]).all()


def test_knn_feat_channel_slice_with_batch_stride():
    import point_ops as PO
    from pcr_amd import dgcnn_engine as DE
    big = torch.randn(3, 96, 128, generator=torch.Generator().manual_seed(9)).cuda()
    sl = big[:, 32:64]
    from pcr_amd import _lib as L
    import ctypes
    xx = torch.empty((3, 128), device="cuda")
    idx = torch.empty((3, 128, 20), dtype=torch.int32, device="cuda")
    L.check(L.load().pcr_knn_feat_f32(L.ptr(sl), L.ptr(xx), L.ptr(idx), 3, 32, 128, 20, ctypes.c_long(96 * 128),
                                      L.stream_ptr()), "pcr_knn_feat_f32")
    assert (idx.cpu().numpy() == PO.knn_feat(sl.cpu().contiguous().numpy(), 20)).all()
    assert (DE.knn_feat(sl.contiguous(), 20) == idx).all()


def test_knn_feat_rejects_bad_arguments():
    from pcr_amd import dgcnn_engine as DE, _lib as L
    with pytest.raises(L.PcrError):
        DE.knn_feat(torch.randn(1, 3, 16).cuda(), 20)          # K > N
    with pytest.raises(L.PcrError):
        DE.knn_feat(torch.randn(1, 3, 4096).cuda(), 20)        # N beyond the LDS key tile
    with pytest.raises(L.PcrError):
        DE.knn_feat(torch.randn(1, 3, 64), 20)                 # host tensor: no CPU fallback


@pytest.mark.parametrize("n,seed", [(128, 3), (200, 4)])
def test_dgcnn_backbone_matches_oracle(n, seed):
    """whole backbone against the torch-eager oracle (materialised edge tensor) on other sizes than the golden"""
    import model_oracle as MO
    m, sd = build()
    x = T.synthetic_clouds(3, n, seed, "randn").permute(0, 2, 1).contiguous()
    st = {}
    with torch.no_grad():
        _, want = MO.dgcnn_backbone(MO._sub(sd, "backbone."), x, 20, st)
        _, got = m.backbone(x.cuda(), None)
    got = got.cpu()
    err = (got - want).abs()
    # a neighbour swap at a near-tie (features differ in the last bits between the two paths) would show up as a
    # few isolated large errors; none happens on these inputs
    assert float(err.max()) < TOL * max(1.0, float(want.abs().max())), float(err.max())


def test_dgcnn_pairs_match_reference_golden():
    g = load_golden("dgcnn_n256_randn")
    meta = g["meta"]
    m, _ = build()
    s1, s2 = T.synthetic_pairs(meta["pairs"], meta["n"], meta["input_seed"], meta["kind"])
    with torch.no_grad():
        xyz, feat = m.backbone(torch.cat([s1, s2], 0).permute(0, 2, 1).contiguous().cuda(), m.backbone_list)
        xyz1, xyz2, h1, h2 = m.siamese_forward(s1.cuda(), s2.cuda())
        logits = m.match_forward_inference(h1, h2, xyz1, xyz2)
    scale = np.abs(g["enc_max"]).max()
    assert np.abs(feat.max(dim=2)[0].cpu().numpy() - g["enc_max"]).max() < TOL * max(1.0, scale)
    assert np.abs(feat.mean(dim=2).cpu().numpy() - g["enc_mean"]).max() < TOL
    assert np.abs(h1.cpu().numpy() - g["h1"]).max() < TOL and np.abs(h2.cpu().numpy() - g["h2"]).max() < TOL
    assert np.abs(logits.cpu().numpy() - g["logits"]).max() < TOL


def test_edge_layer_matches_materialised_edge_formulation():
    """one EdgeConv layer from given neighbour indices: decomposed tables + gather-max == the reference's
    cat[f_j - f_i, f_i] -> conv -> BN -> LeakyReLU -> max (also with negative BatchNorm scales)"""
    import ctypes
    import model_oracle as MO
    from pcr_amd import dgcnn_engine as DE, engine as E, _lib as L
    g = torch.Generator().manual_seed(11)
    B, C, N, Co, K = 2, 24, 150, 96, 9
    f = torch.randn(B, C, N, generator=g)
    idx = torch.randint(0, N, (B, N, K), generator=g)
    w = torch.randn(Co, 2 * C, 1, 1, generator=g) * 0.2
    bn = torch.nn.BatchNorm2d(Co)
    with torch.no_grad():
        bn.weight.copy_(torch.randn(Co, generator=g))           # both signs
        bn.bias.copy_(torch.randn(Co, generator=g))
        bn.running_mean.copy_(torch.randn(Co, generator=g) * 0.1)
        bn.running_var.copy_(torch.rand(Co, generator=g) + 0.5)
    bn.eval()
    with torch.no_grad():
        want = F.leaky_relu(bn(F.conv2d(MO.graph_feature(f, idx), w)), 0.2).max(dim=-1)[0]
    sc, sh = E.fold_bn(bn, None, "cuda")
    s = sc.double().cpu().unsqueeze(1)
    w2d = w.double().reshape(Co, 2 * C)
    wa = E.pack_weight((s * w2d[:, :C]).float(), "cuda")
    wb = E.pack_weight((s * (w2d[:, C:] - w2d[:, :C])).float(), "cuda")
    fc = f.cuda()
    ta, tb = DE._table(fc, wa, Co), DE._table(fc, wb, Co)
    out = torch.empty((B, Co, N), device="cuda")
    L.check(L.load().pcr_edge_max_f32(L.ptr(ta), L.ptr(tb), L.ptr(idx.int().cuda()), L.ptr(sh), ctypes.c_float(0.2),
                                      L.ptr(out), ctypes.c_long(0), None, ctypes.c_long(0), B, N, Co, K,
                                      L.stream_ptr()), "pcr_edge_max_f32")
    assert float((out.cpu() - want).abs().max()) < 2e-5 * max(1.0, float(want.abs().max()))


# ---- local_self_attention / match_type='xcorr' (attention.py:221-296, ReIDNet.py:250-256) -------------------------
LOCAL = dict(type="local_self_attention", d_model=64, nhead=2, attention="linear", knum=48, pos_size=64)


def build_xcorr():
    import bench
    from mmdet3d.models import build_model
    cfg = copy.deepcopy(bench.PT_MODEL)
    cfg.update(match_type="xcorr", local_stage1=dict(LOCAL), local_stage2=dict(LOCAL), backbone_list=[128, 64, 32])
    m = build_model(cfg)
    man = T.load_manifest(os.path.join(GOLDEN, "pt_xcorr_manifest.json"))
    assert T.manifest_of(m) == man
    sd = T.seeded_state_dict(man, 0)
    m.load_state_dict(sd, strict=True)
    return m.cuda().eval(), sd


def test_xcorr_pairs_match_reference_golden():
    g = load_golden("pt_xcorr_n128_randn")
    meta = g["meta"]
    m, _ = build_xcorr()
    s1, s2 = T.synthetic_pairs(meta["pairs"], meta["n"], meta["input_seed"], meta["kind"])
    from pcr_amd import engine
    with torch.no_grad():
        # stage by stage in the reference's arithmetic (level 2 = the f32 kernels for the whole chain): local_self_attention
        # searches its neighbours in FEATURE space, so a last-bit difference in `a` can swap a neighbour and move `b` by 1e-2
        # without any stage being wrong (tools/fuzz_models.py forced_ptx) -- the stages are held to the golden in the arithmetic
        # the golden was recorded in, the logits through the guarded entry points in whatever level the guard chose
        with engine.guard_level(2):
            xyz1, xyz2, h1, h2 = m.siamese_forward(s1.cuda(), s2.cuda())
            a = m.cross_stage1(h1, xyz1, h2, xyz2)
            b = m.local_stage1(a, xyz1)
        xyz1, xyz2, h1, h2 = m.siamese_forward(s1.cuda(), s2.cuda())
        logits = m.match_forward_inference(h1, h2, xyz1, xyz2)
    assert np.abs(a.cpu().numpy() - g["xc_a"]).max() < TOL
    assert np.abs(b.cpu().numpy() - g["xc_b"]).max() < TOL
    assert np.abs(logits.cpu().numpy() - g["logits"]).max() < TOL


@pytest.mark.parametrize("n,knum", [(100, 48), (64, 32), (257, 20)])
def test_local_self_attention_matches_oracle(n, knum):
    import model_oracle as MO
    from mmdet3d.models.attention import local_self_attention
    m = local_self_attention(64, 2, knum=knum, pos_size=64)
    sd = T.seeded_state_dict(T.manifest_of(m), 13)
    m.load_state_dict(sd)
    g = torch.Generator().manual_seed(n)
    feat = torch.randn(3, 64, n, generator=g)
    xyz = torch.randn(3, n, 3, generator=g)
    with torch.no_grad():
        want = MO.local_self_attention(sd, feat, xyz, 2, knum)
        got = m.cuda().eval()(feat.cuda(), xyz.cuda()).cpu()
    assert float((got - want).abs().max()) < TOL
