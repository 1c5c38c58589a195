"""Training mode beyond the default Point-Transformer config: the reference's other training
configs run the same `ReIDNet.forward_train` (mmdet3d/models/ReIDNet.py:586-634).  Families with a HIP training graph
are pinned to loss / accuracy / gradients recorded from the REFERENCE's own train_step
(oracle/make_golden.py gen_train_variants -> tests/golden/train_step_<tag>_n128.npz); families without one must fail
with a clean PcrError (never a crash, never a silent eval-mode forward)."""
import copy
import json
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, load_golden
from pcr_amd import testing as T
from test_gpu_config_variants import BASELINE, _pt_mul

pytestmark = pytest.mark.gpu

STNET = copy.deepcopy(_pt_mul(1, 64, 8))
STNET.update(match_type="xcorr-baseline", hidden_size=128)
# reid_pts_point-transformer_baseline_orig.py (reid_waymo_pts/pts_point-transformer_baseline-orig_waymo_det_4x256_400e.py)
_LOCAL = dict(type="local_self_attention", d_model=64, nhead=2, attention="linear", knum=48, pos_size=64)
ORIG = copy.deepcopy(_pt_mul(1, 64, 8))
ORIG.update(match_type="xcorr", hidden_size=128, local_stage1=dict(_LOCAL), local_stage2=dict(_LOCAL))


def _train_data(pairs, n, dev="cuda"):
    s1, s2 = T.synthetic_pairs(pairs, n, seed=2, kind="randn")
    ids1 = torch.arange(pairs)
    ids2 = torch.where(torch.arange(pairs) < pairs // 2, ids1, ids1 + 100)
    zero = torch.zeros(1, dtype=torch.long, device=dev)
    return dict(sparse_1=list(s1.to(dev)), sparse_2=list(s2.to(dev)), dense_1=list(s1.to(dev)), dense_2=list(s2.to(dev)),
                label_1=[zero] * pairs, label_2=[zero] * pairs,
                id_1=[i.view(1).to(dev) for i in ids1], id_2=[i.view(1).to(dev) for i in ids2])


LOSSES = dict(kl=False, match=True, cls=False, shape=False, fp=False, triplet=False)     # every ReID config's setting


def _build(cfg, manifest):
    from mmdet3d.models import build_model
    cfg = copy.deepcopy(cfg)
    cfg["losses_to_use"] = dict(LOSSES)
    m = build_model(cfg)
    man = T.load_manifest(os.path.join(GOLDEN, manifest + "_manifest.json"))
    assert T.manifest_of(m) == man
    m.load_state_dict(T.seeded_state_dict(man, 0), strict=True)
    return m.cuda()


def _pointnet_cfg():
    import bench
    return copy.deepcopy(bench.PN_MODEL)


def _dgcnn_cfg():
    import bench
    return copy.deepcopy(bench.DG_MODEL)


@pytest.mark.parametrize("tag,cfg,manifest", [("stnet", STNET, "pt"), ("orig", ORIG, "pt_xcorr"), ("baseline", BASELINE, "pt_baseline"),
                                              ("pt15m", _pt_mul(2, 64, 8), "pt15m"), ("pointnet", None, "pointnet"),
                                              ("dgcnn", None, "dgcnn")])
def test_train_step_of_other_configs_matches_the_reference(tag, cfg, manifest, grad_floor):
    cfg = cfg if cfg is not None else (_pointnet_cfg() if tag == "pointnet" else _dgcnn_cfg())
    g = load_golden("train_step_%s_n128" % tag)
    meta = g["meta"]
    m = _build(cfg, manifest)
    m.train()
    out = m.train_step(_train_data(meta["pairs"], meta["n"]), None)
    loss = float(out["loss"])
    out["loss"].backward()
    params = dict(m.named_parameters())
    worst, ref_own, vs64 = {}, {}, {}
    for k in g:
        if k.startswith("grad:"):
            got = params[k[5:]].grad.cpu().numpy()
            scale = max(1e-3, float(np.abs(g[k]).max()))
            worst[k[5:]] = float(np.abs(got - g[k]).max()) / scale
            g64 = g["grad64:" + k[5:]]
            ref_own[k[5:]] = float(np.abs(g[k] - g64).max()) / scale          # the reference's float32 vs its float64
            vs64[k[5:]] = float(np.abs(got - g64).max()) / scale              # the HIP gradient vs the float64 truth
    gn = float(torch.sqrt(sum((p.grad.double() ** 2).sum() for p in params.values() if p.grad is not None)))
    bufs = dict(m.named_buffers())
    bw = {k[4:]: float(np.abs(bufs[k[4:]].cpu().numpy() - g[k]).max()) for k in g if k.startswith("buf:")}
    print(json.dumps(dict(tag=tag, loss=loss, ref_loss=float(g["loss"]), grad_norm=gn, ref_grad_norm=float(g["grad_norm"]),
                          vs_ref32=worst, ref32_vs_ref64=ref_own, vs_ref64=vs64, running_mean=bw)))
    assert abs(loss - float(g["loss"])) < 1e-4
    assert abs(out["log_vars"]["match_acc"] - float(g["match_acc"])) < 1e-6
    # Everything behind the encoder's last max-over-K (SA3's attention, FP, cov_final, matching, head) agrees with the
    # reference's float32 gradients to ~1e-6 of the tensor's scale (bound 1e-5).  In FRONT of a max-over-K / BatchNorm
    # batch statistics a near-tie resolved the other way re-routes a whole gradient row, and float32 runs of the SAME
    # graph disagree by 1e-4 .. 1e-2 -- the reference's own float32 gradients are that far from its float64 ones
    # (ref32_vs_ref64, recorded by oracle/make_golden.py).  So those tensors are held to the float64 gradients, with the
    # reference's own float32 error as the yardstick: the HIP gradient must be no further from the exact one than
    # twice what the reference's float32 backward is.
    # PointNet: the input / feature transform nets normalise their fully connected outputs over the BATCH of clouds
    # (BatchNorm1d on 8 rows here, models/pointnet.py:38-40): channels whose variance over eight samples is near eps
    # amplify rounding a hundred-fold, and the 1024-term f32 MFMA chains of fc1 / the downsample layers (7e-7 of the
    # output's scale, sequential in k) are ~5x coarser than torch's blocked CPU matmul.  Observed: 1.4e-5 on the loss,
    # <= 2.7e-4 on the gradients behind the transforms (the reference's own float32: 1e-5 .. 2e-5), while the encoder's
    # own tensors sit at or below the reference's float32 error.  Floor for this family: 5e-4.
    floor = 5e-4 if tag == "pointnet" else grad_floor
    for k in worst:
        assert worst[k] < grad_floor or vs64[k] < max(floor, 2.0 * ref_own[k]), (k, worst[k], vs64[k], ref_own[k])
    assert gn == pytest.approx(float(g["grad_norm"]), rel=2e-3)
    assert all(v < 1e-5 for v in bw.values()), bw
    no_grad = sorted(k for k, p in params.items() if p.grad is None)
    assert no_grad == sorted(json.loads(str(g["no_grad_params"])))


def test_other_configs_train_through_the_trainer():
    """eight Trainer steps (HIP AdamW) on the `concat` baseline, the mul = 2 model and PointNet: finite, decreasing on a
    fixed batch"""
    from pcr_amd import train
    for cfg, manifest, pairs in ((BASELINE, "pt_baseline", 8), (_pt_mul(2, 64, 8), "pt15m", 4),
                                 (_pointnet_cfg(), "pointnet", 4), (_dgcnn_cfg(), "dgcnn", 4)):
        m = _build(cfg, manifest)
        m.train()
        data = _train_data(pairs, 128)
        tr = train.Trainer(m, max_iters=40, lr=3e-4, grad_clip=1.0)
        losses = [float(tr.step(data)["loss"].detach()) for _ in range(12)]
        # (a fixed batch of 4-8 pairs under a rising cyclic lr: the trajectory is noisy; it must stay finite and get below
        # its starting point)
        assert all(np.isfinite(losses)) and min(losses[1:]) < losses[0], losses


def _expect_clean_refusal(m, pairs=2, n=128):
    from pcr_amd import _lib as L
    m.train()
    with pytest.raises(L.PcrError):
        out = m.train_step(_train_data(pairs, n), None)
        out["loss"].backward()


def test_families_without_a_training_graph_fail_cleanly():
    import bench
    m, _ = bench.build_model("ssg", None)                     # BASELINE config 2's own composition: inference only
    _expect_clean_refusal(m, n=1024)
    _expect_clean_refusal(_build(_pt_mul(4, 128, 16), "pt7m"))   # mul = 4: 256-wide attention heads
    # the models still evaluate after the refusal
    m, sd = bench.build_model("pointnet", None)
    m.eval()
    s1, s2 = T.synthetic_pairs(2, 128, seed=3)
    with torch.no_grad():
        assert torch.isfinite(bench.hot_path(m, s1.cuda(), s2.cuda())).all()
