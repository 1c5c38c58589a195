"""GPU parity of the mmdet3d-style PointNet++ modules (PointSAModule / PointFPModule / Points_Sampler /
QueryAndGroup) and of the PointNet2SSG siamese model (BASELINE config 2) against the oracles."""
import copy
import json

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from pcr_amd import testing as T
import point_ops as P

pytestmark = pytest.mark.gpu
TOL = 1e-4


def ssg_model():
    import bench
    return bench.build_model("ssg", None)


@pytest.mark.parametrize("pairs,n,kind", [(3, 1024, "box"), (2, 1024, "dup"), (2, 700, "randn")])
def test_ssg_pairs_match_oracle(pairs, n, kind):
    import model_oracle as MO
    model, sd = ssg_model()
    s1, s2 = T.synthetic_pairs(pairs, n, seed=5, kind=kind)
    ref = {}
    with torch.no_grad():
        want = MO.ssg_pairs(sd, s1, s2, stages=ref)
        xyz1, xyz2, h1, h2 = model.siamese_forward(s1.cuda(), s2.cuda())
        logits = model.match_forward_inference(h1, h2, xyz1, xyz2)
    assert xyz1.shape == (pairs, 128, 3) and h1.shape == (pairs, 64, 128)
    worst = dict(h1=float((h1.cpu() - ref["h1"]).abs().max()), h2=float((h2.cpu() - ref["h2"]).abs().max()),
                 logits=float((logits.cpu() - want).abs().max()))
    print(json.dumps(worst))
    assert max(worst.values()) < TOL, worst


def test_point_sa_module_stages_bit_exact_indices():
    from mmdet3d.ops import PointSAModule
    import model_oracle as MO
    torch.manual_seed(0)
    sa = PointSAModule(mlp_channels=[6, 32, 32, 64], num_point=96, radius=0.5, num_sample=16)
    sd = T.seeded_state_dict(T.manifest_of(sa), 3)
    sa.load_state_dict(sd)
    sa = sa.cuda().eval()
    xyz = T.synthetic_clouds(2, 300, seed=2, kind="dup")
    feats = torch.randn(2, 6, 300)
    new_xyz, out, idx = sa(xyz.cuda(), feats.cuda())
    st = {}
    with torch.no_grad():
        rx, ro = MO.ssg_sa_layer(sd, xyz, feats, 96, 0.5, 16, st, "s")
    assert (idx.cpu().numpy() == st["s_fps"].numpy()).all()
    assert torch.equal(new_xyz.cpu(), rx)
    # (default arithmetic bf16x3: tables and layers 2 / 3 on the bf16 matrix core; the f32 mode of the same layers is held
    # to 2e-6 of the output's scale in tests/test_gpu_precision.py)
    assert float((out.cpu() - ro).abs().max()) < 5e-5
    # stand-alone grouper = same indices as the C oracle
    from mmdet3d.ops import QueryAndGroup
    grouped, gidx = QueryAndGroup(0.5, 16, return_grouped_idx=True)(xyz.cuda(), new_xyz, feats.cuda())
    assert (gidx.cpu().numpy() == st["s_ball"].numpy()).all()
    assert grouped.shape == (2, 9, 96, 16)
    with pytest.raises(RuntimeError):
        sa.train()
        sa(xyz.cuda(), feats.cuda())


def test_point_fp_module_matches_torch():
    from mmdet3d.ops import PointFPModule
    fp = PointFPModule(mlp_channels=[24 + 10, 48, 32])
    sd = T.seeded_state_dict(T.manifest_of(fp), 4)
    fp.load_state_dict(sd)
    fp = fp.cuda().eval()
    target = T.synthetic_clouds(2, 257, seed=3, kind="box")
    source = target[:, :40].contiguous()
    tf, sf = torch.randn(2, 10, 257), torch.randn(2, 24, 40)
    out = fp(target.cuda(), source.cuda(), tf.cuda(), sf.cuda()).cpu()
    d2, i3 = P.three_nn(target.numpy(), source.numpy())
    dist = torch.from_numpy(np.sqrt(d2))
    w = 1.0 / (dist + 1e-8)
    w = w / w.sum(dim=2, keepdim=True)
    interp = torch.from_numpy(P.three_interp_fwd(sf.numpy(), i3, w.numpy()))
    x = torch.cat([interp, tf], dim=1).unsqueeze(-1)
    with torch.no_grad():
        for i in range(2):
            x = F.conv2d(x, sd[f"mlps.layer{i}.conv.weight"])
            x = F.relu(F.batch_norm(x, sd[f"mlps.layer{i}.bn.running_mean"], sd[f"mlps.layer{i}.bn.running_var"],
                                    sd[f"mlps.layer{i}.bn.weight"], sd[f"mlps.layer{i}.bn.bias"], False, 0.0, 1e-5))
    assert float((out - x.squeeze(-1)).abs().max()) < 2e-5


def test_points_sampler_modes_match_oracle():
    """D-FPS, F-FPS (feature-space FPS over the calc_square_dist matrix) and FS (both, concatenated), and a two-range
    sampler list (points_sampler.py:66-157): every index against the C oracle"""
    from mmdet3d.ops import Points_Sampler, calc_square_dist
    xyz = T.synthetic_clouds(2, 256, seed=9, kind="randn")
    feats = torch.randn(2, 8, 256, generator=torch.Generator().manual_seed(3))
    f = np.concatenate([xyz.numpy(), feats.numpy().transpose(0, 2, 1)], axis=2)
    dmat = P.pairwise_sqdist(f, f)
    got = calc_square_dist(torch.from_numpy(f).cuda(), torch.from_numpy(f).cuda(), norm=False).cpu().numpy()
    assert np.array_equal(got, dmat)
    gotn = calc_square_dist(torch.from_numpy(f).cuda(), torch.from_numpy(f[:, :40]).cuda(), norm=True).cpu().numpy()
    assert np.array_equal(gotn, P.pairwise_sqdist(f, f[:, :40], norm=True), equal_nan=True)
    idx = Points_Sampler([64], ["D-FPS"], [-1])(xyz.cuda(), feats.cuda()).cpu().numpy()
    assert (idx == P.fps(xyz.numpy(), 64)).all()
    idx = Points_Sampler([32], ["F-FPS"], [-1])(xyz.cuda(), feats.cuda())
    assert idx.dtype == torch.int32 and (idx.cpu().numpy() == P.fps_dist(dmat, 32)).all()
    idx = Points_Sampler([16], ["FS"], [-1])(xyz.cuda(), feats.cuda()).cpu().numpy()
    assert (idx == np.concatenate([P.fps_dist(dmat, 16), P.fps(xyz.numpy(), 16)], axis=1)).all()
    # two ranges: F-FPS on points [0,100), D-FPS on the rest; indices of the second range are offset by 100
    idx = Points_Sampler([8, 24], ["F-FPS", "D-FPS"], [100, -1])(xyz.cuda(), feats.cuda()).cpu().numpy()
    f0 = f[:, :100]
    want = np.concatenate([P.fps_dist(P.pairwise_sqdist(f0, f0), 8), P.fps(xyz.numpy()[:, 100:], 24) + 100], axis=1)
    assert (idx == want).all()


def test_point_sa_module_msg_two_scales_match_oracle():
    """PointSAModuleMSG (point_sa_module.py:219-299): one FPS, one ball query + shared MLP per radius (the second
    one dilated: min_radius = the first radius), outputs concatenated along the channels"""
    from mmdet3d.ops import PointSAModuleMSG
    import model_oracle as MO
    for dilated in (False, True):
        sa = PointSAModuleMSG(num_point=64, radii=[0.3, 0.6], sample_nums=[16, 32],
                              mlp_channels=[[6, 16, 16, 32], [6, 32, 32, 64]], dilated_group=dilated)
        sd = T.seeded_state_dict(T.manifest_of(sa), 9)
        sa.load_state_dict(sd)
        sa = sa.cuda().eval()
        xyz = T.synthetic_clouds(2, 400, seed=12, kind="dup")
        feats = torch.randn(2, 6, 400, generator=torch.Generator().manual_seed(2))
        new_xyz, out, idx = sa(xyz.cuda(), feats.cuda())
        x = xyz.numpy()
        fps = P.fps(x, 64)
        assert (idx.cpu().numpy() == fps).all()
        cx = np.take_along_axis(x, fps[..., None].astype(np.int64).repeat(3, -1), 1)
        assert np.array_equal(new_xyz.cpu().numpy(), cx)
        outs = []
        with torch.no_grad():
            for i, (r, k) in enumerate(((0.3, 16), (0.6, 32))):
                bq = torch.from_numpy(P.ball_query(0.3 if (dilated and i) else 0.0, r, k, x, cx)).long()
                g = MO._gather_rows(xyz, bq) - torch.from_numpy(cx).unsqueeze(2)
                h = torch.cat([g, MO._gather_rows(feats.permute(0, 2, 1), bq)], dim=-1).permute(0, 3, 1, 2)
                for l in range(3):
                    h = F.conv2d(h, sd[f"mlps.{i}.layer{l}.conv.weight"])
                    h = F.relu(MO._bn(h, sd, f"mlps.{i}.layer{l}.bn", 4))
                outs.append(h.max(dim=3)[0])
        want = torch.cat(outs, dim=1)
        assert out.shape == (2, 96, 64)
        assert float((out.cpu() - want).abs().max()) < 5e-5, dilated


def test_group_all_and_base_module_contract():
    """GroupAll (group_points.py:132-166): [xyz ; features] as one group of all N points; a BasePointSAModule cannot
    be built with num_point=None (point_sa_module.py:74-79 raises) in the reference either"""
    from mmdet3d.ops import GroupAll, PointSAModuleMSG
    xyz = T.synthetic_clouds(2, 50, seed=1, kind="box").cuda()
    feats = torch.randn(2, 5, 50, device="cuda")
    g = GroupAll(use_xyz=True)(xyz, None, feats)
    assert g.shape == (2, 8, 1, 50)
    assert torch.equal(g[:, :3, 0], xyz.transpose(1, 2)) and torch.equal(g[:, 3:, 0], feats)
    assert torch.equal(GroupAll(use_xyz=False)(xyz, None, feats), feats.unsqueeze(2))
    assert torch.equal(GroupAll()(xyz, None, None), xyz.transpose(1, 2).unsqueeze(2))
    with pytest.raises(NotImplementedError):
        PointSAModuleMSG(num_point=None, radii=[0.3], sample_nums=[8], mlp_channels=[[3, 8, 8, 8]])


@pytest.mark.parametrize("n,npoint,radius,k,chans,kind", [(1024, 512, 0.2, 32, [0, 64, 64, 128], "box"),
                                                          (512, 128, 0.4, 64, [16, 128, 128, 256], "dup"),
                                                          (300, 77, 5.0, 16, [6, 32, 32, 64], "randn"),
                                                          (200, 50, 0.05, 24, [3, 32, 64, 32], "box")])
def test_repeat_skipping_is_exact(n, npoint, radius, k, chans, kind):
    """the duplicate-free evaluation (ball-query hit counts) gives bit-identical features to the K-row one,
    from groups that are all padding (tiny radius) to groups that are completely full (huge radius)"""
    from mmdet3d.ops import PointSAModule, ball_query_cnt
    sa = PointSAModule(mlp_channels=list(chans), num_point=npoint, radius=radius, num_sample=k)
    sa.load_state_dict(T.seeded_state_dict(T.manifest_of(sa), 5))
    sa = sa.cuda().eval()
    xyz = T.synthetic_clouds(3, n, seed=6, kind=kind).cuda()
    feats = torch.randn(3, chans[0], n).cuda() if chans[0] else None
    sa.skip_repeats = False
    _, dense, idx = sa(xyz, feats)
    sa.skip_repeats = True
    new_xyz, ragged, idx2 = sa(xyz, feats)
    assert torch.equal(idx, idx2)
    assert torch.equal(dense, ragged)
    bq, cnt = ball_query_cnt(0.0, radius, k, xyz, new_xyz)
    want = P.ball_query(0.0, radius, k, xyz.cpu().numpy(), new_xyz.cpu().numpy())
    assert (bq.cpu().numpy() == want).all()
    uniq = np.array([[len(np.unique(r)) for r in b] for b in want])
    c = cnt.cpu().numpy()
    assert ((c == uniq) | ((c == 0) & (uniq == 1))).all()      # cnt = 0: no hit at all (row of zeros)


def test_point_major_layouts_change_nothing():
    """features may travel between SA layers as the (B,C,S) view of a point-major buffer: every consumer reads
    both layouts and gives identical results; the channel-major output form of the ragged kernel agrees too"""
    from mmdet3d.ops import PointSAModule
    from mmdet3d.ops.point_ops import ball_query_cnt, furthest_point_sample, gather_points
    from pcr_amd import engine
    sa1 = PointSAModule(mlp_channels=[0, 32, 32, 64], num_point=128, radius=0.3, num_sample=16)
    sa2 = PointSAModule(mlp_channels=[64, 64, 64, 128], num_point=32, radius=0.6, num_sample=32)
    for i, m in enumerate((sa1, sa2)):
        m.load_state_dict(T.seeded_state_dict(T.manifest_of(m), 11 + i))
        m.cuda().eval()
    xyz = T.synthetic_clouds(3, 500, seed=8, kind="box").cuda()
    x1, f1, _ = sa1(xyz, None)
    assert f1.shape == (3, 64, 128) and not f1.is_contiguous() and f1.transpose(1, 2).is_contiguous()
    _, f2_pm, _ = sa2(x1, f1)                      # point-major view in
    _, f2_cm, _ = sa2(x1, f1.contiguous())         # channel-major copy in
    assert torch.equal(f2_pm, f2_cm)
    # the plan itself, channel-major output against point-major output
    idx1 = furthest_point_sample(x1, 32)
    c = gather_points(x1.transpose(1, 2).contiguous(), idx1).transpose(1, 2).contiguous()
    bq, cnt = ball_query_cnt(0.0, 0.6, 32, x1, c)
    plan = sa2._plan(0, xyz.device)
    o_cm = plan.run(x1, f1, bq, centre_idx=idx1, cnt=cnt, out_point_major=False)
    o_pm = plan.run(x1, f1, bq, centre_idx=idx1, cnt=cnt, out_point_major=True)
    o_k = plan.run(x1, f1.contiguous(), bq, centre_idx=idx1, cnt=None, out_point_major=True)   # K-row kernel
    assert o_cm.is_contiguous() and torch.equal(o_cm, o_pm) and torch.equal(o_cm, o_k) and torch.equal(o_cm, f2_pm)
    # generic dense layer on either layout
    w = torch.randn(48, 128, device="cuda")
    wp = engine.pack_weight(w, xyz.device)
    assert torch.equal(engine.dense(f2_pm, wp, 48), engine.dense(f2_pm.contiguous(), wp, 48))


@pytest.mark.parametrize("B,n,npoint,k,radius", [(1, 64, 1, 8, 0.5), (2, 70, 33, 4, 0.2), (5, 1024, 512, 32, 0.05)])
def test_ragged_path_small_and_degenerate_shapes(B, n, npoint, k, radius):
    """one centre, odd sizes, (almost) empty balls: the tile lists and row tables of the ragged path hold up"""
    from mmdet3d.ops import PointSAModule
    sa = PointSAModule(mlp_channels=[0, 16, 32, 64], num_point=npoint, radius=radius, num_sample=k)
    sa.load_state_dict(T.seeded_state_dict(T.manifest_of(sa), 3))
    sa = sa.cuda().eval()
    xyz = T.synthetic_clouds(B, n, seed=2, kind="box").cuda()
    sa.skip_repeats = False
    _, dense, _ = sa(xyz, None)
    sa.skip_repeats = True
    _, ragged, _ = sa(xyz, None)
    assert torch.equal(dense, ragged)
