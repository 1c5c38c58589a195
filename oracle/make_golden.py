"""TEST INFRASTRUCTURE ONLY -- golden-vector generator (dev container only).

Imports the reference from /root/reference (oracle/ref_loader.py), loads the seeded synthetic
state_dict (pcr_amd/testing.py) into the reference's own ReIDNet and records tensors at every
stage boundary of the hot path.  Only tensors (npz) and name/shape manifests
(json) are written to tests/golden/; no reference source or bytecode is copied.

Run:  python oracle/make_golden.py            (writes tests/golden/*)
"""
import contextlib
import copy
import io
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(ROOT, "point-cloud-reid_amd"))

import ref_loader  # noqa: E402
from pcr_amd import testing as T  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
PT_CFG = "configs_reid/_base_/reidentifiers/reid_pts_point-transformer_point-cat.py"
PN_CFG = "configs_reid/_base_/reidentifiers/reid_pts_pointnet_point-cat.py"


def _np(t):
    return t.detach().cpu().numpy()


def build(cfg, seed=0, **over):
    m = ref_loader.build_ref_reidnet(cfg, **over)
    man = T.manifest_of(m)
    m.load_state_dict(T.seeded_state_dict(man, seed), strict=True)
    m.eval()
    return m, man


def record_pt(model, s1, s2):
    """Run reference siamese_forward + match_forward_inference, capturing stage tensors."""
    ref = ref_loader.load_reference()
    p2u = ref.pointnet2_utils
    rec = {}
    knn_calls = []
    orig_knn = p2u.knn_point

    def knn_spy(nsample, xyz, new_xyz):
        idx = orig_knn(nsample, xyz, new_xyz)
        knn_calls.append(idx)
        return idx

    hooks = []
    bb = model.backbone
    for i, sa in enumerate(bb.SA_modules):
        hooks.append(sa.self_attention.register_forward_pre_hook(
            lambda mod, args, i=i: rec.__setitem__(f"sa{i}_mlp", _np(args[0]))))
        hooks.append(sa.register_forward_hook(
            lambda mod, args, out, i=i: rec.__setitem__(f"sa{i}_out", _np(out[1]))))
    for j, fp in enumerate(bb.FP_modules):
        hooks.append(fp.register_forward_hook(
            lambda mod, args, out, j=j: rec.__setitem__(f"fp{j}_out", _np(out))))
    cross = {"cross_stage1": [], "cross_stage2": []}
    for name in cross:
        hooks.append(getattr(model, name).register_forward_hook(
            lambda mod, args, out, name=name: cross[name].append(_np(out))))
    p2u.knn_point = knn_spy
    try:
        with torch.no_grad(), contextlib.redirect_stdout(io.StringIO()):
            xyz1, xyz2, h1, h2 = model.siamese_forward(s1, s2)
            match_in, o1, o2 = model.xcorr_eff(h1, xyz1, h2, xyz2, model.combine)
            pooled = model.get_pooled_feats(match_in)
            logits = model.match_forward_inference(h1, h2, xyz1, xyz2)
    finally:
        p2u.knn_point = orig_knn
        for h in hooks:
            h.remove()
    for i, idx in enumerate(knn_calls[:3]):
        # order inside a K-set is not part of the contract (hard part 1)
        rec[f"sa{i}_knn_sorted"] = np.sort(_np(idx), axis=-1).astype(np.int16)
    rec["h1"], rec["h2"] = _np(h1), _np(h2)
    # xcorr_eff was called twice (explicitly and inside match_forward_inference): keep the first
    rec["x1_o1"], rec["x1_o2"] = cross["cross_stage1"][0], cross["cross_stage1"][1]
    rec["x2_o1"], rec["x2_o2"] = cross["cross_stage2"][0], cross["cross_stage2"][1]
    rec["pooled"] = _np(pooled)
    rec["logits"] = _np(logits)
    return rec


def gen_pt():
    cases = [  # name, pairs, N, backbone_list, kind, keep-all-stages
        ("pt_n128_randn", 2, 128, [128, 64, 32], "randn", True),
        ("pt_n128_dup", 2, 128, [128, 64, 32], "dup", True),
        ("pt_n256_box", 2, 256, [256, 128, 64], "box", False),
        ("pt_n1024_randn", 1, 1024, [1024, 512, 256], "randn", False),
    ]
    manifest = None
    for name, b, n, bl, kind, full in cases:
        model, manifest = build(PT_CFG, seed=0, backbone_list=bl)
        s1, s2 = T.synthetic_pairs(b, n, seed=1, kind=kind)
        rec = record_pt(model, s1, s2)
        if not full:
            keep = ("sa0_knn_sorted", "sa1_knn_sorted", "sa2_knn_sorted", "sa2_out", "fp0_out",
                    "h1", "h2", "x2_o1", "pooled", "logits")
            rec = {k: v for k, v in rec.items() if k in keep}
        rec["meta"] = np.array(json.dumps(dict(pairs=b, n=n, backbone_list=bl, kind=kind,
                                               input_seed=1, weight_seed=0)))
        np.savez_compressed(os.path.join(GOLD, name + ".npz"), **rec)
        print(name, {k: getattr(v, "shape", None) for k, v in rec.items()}, "logits", rec["logits"])
    with open(os.path.join(GOLD, "pt_manifest.json"), "w") as f:
        json.dump(manifest, f)


BASE_CFG = "configs_reid/_base_/reidentifiers/reid_pts_point-transformer_baseline.py"
MUL_CFGS = (("pt15m", "configs_reid/_base_/reidentifiers/reid_pts_point-transformer-1.5M_point-cat.py"),
            ("pt7m", "configs_reid/_base_/reidentifiers/reid_pts_point-transformer-7M_point-cat.py"))


def gen_baseline():
    """reid_pts_point-transformer_baseline.py: match_type='concat', pool_type='max' (a max over the 64 CHANNELS of
    every point, ReIDNet.py:145,526-528), LinearRes(256) head.  The config's shape_head (105 M parameters, never
    evaluated: losses_to_use.shape is off) is left out of the fixture model."""
    model, manifest = build(BASE_CFG, seed=0, shape_head=None)
    s1, s2 = T.synthetic_pairs(4, 128, seed=1, kind="randn")
    with torch.no_grad(), contextlib.redirect_stdout(io.StringIO()):
        xyz1, xyz2, h1, h2 = model.siamese_forward(s1, s2)
        pooled1 = model.get_pooled_feats(h1)
        logits = model.match_forward_inference(h1, h2, xyz1, xyz2)
        # the training / forward_test entry point pools through get_pooled_feats (:412-416): same numbers here
        preds, _, _ = model.match_forward(h1, h2, xyz1, xyz2, torch.zeros(4), None, "cpu")
    assert torch.equal(preds, logits)
    rec = dict(h1=_np(h1), h2=_np(h2), pooled1=_np(pooled1), logits=_np(logits),
               meta=np.array(json.dumps(dict(pairs=4, n=128, kind="randn", input_seed=1, weight_seed=0,
                                             backbone_list=[128, 64, 32]))))
    np.savez_compressed(os.path.join(GOLD, "pt_baseline_n128_randn.npz"), **rec)
    with open(os.path.join(GOLD, "pt_baseline_manifest.json"), "w") as f:
        json.dump(manifest, f)
    print("baseline", rec["logits"], rec["pooled1"].shape)


def gen_pt_mul():
    """the 1.5M (mul=2) and 7M (mul=4) Point-Transformer configs (backbone_net.py:43-46,84-86: SA widths 64/128/256
    and 128/256/512, attention d_model up to 512), 2 pairs of 128 points"""
    for tag, cfg in MUL_CFGS:
        model, manifest = build(cfg, seed=0)
        s1, s2 = T.synthetic_pairs(2, 128, seed=1, kind="randn")
        rec = record_pt(model, s1, s2)
        keep = ("sa0_knn_sorted", "sa1_knn_sorted", "sa2_knn_sorted", "sa0_mlp", "sa0_out", "sa1_out", "sa2_mlp",
                "sa2_out", "fp2_out", "fp1_out", "fp0_out", "h1", "h2", "x2_o1", "pooled", "logits")
        rec = {k: v for k, v in rec.items() if k in keep}
        rec["meta"] = np.array(json.dumps(dict(pairs=2, n=128, backbone_list=[128, 64, 32], kind="randn",
                                               input_seed=1, weight_seed=0)))
        np.savez_compressed(os.path.join(GOLD, tag + "_n128_randn.npz"), **rec)
        with open(os.path.join(GOLD, tag + "_manifest.json"), "w") as f:
            json.dump(manifest, f)
        print(tag, "params", sum(int(np.prod(m[1])) for m in manifest), "logits", rec["logits"])


def gen_pointnet():
    model, manifest = build(PN_CFG, seed=0)
    s1, s2 = T.synthetic_pairs(2, 256, seed=1, kind="randn")
    with torch.no_grad(), contextlib.redirect_stdout(io.StringIO()):
        xyz1, xyz2, h1, h2 = model.siamese_forward(s1, s2)
        feat_xyz, feat = model.backbone(torch.cat([s1, s2], 0).permute(0, 2, 1), model.backbone_list)
        logits = model.match_forward_inference(h1, h2, xyz1, xyz2)
    rec = dict(h1=_np(h1), h2=_np(h2), logits=_np(logits),
               enc_max=_np(feat.max(dim=2)[0]), enc_mean=_np(feat.mean(dim=2)),
               meta=np.array(json.dumps(dict(pairs=2, n=256, kind="randn", input_seed=1, weight_seed=0))))
    np.savez_compressed(os.path.join(GOLD, "pointnet_n256_randn.npz"), **rec)
    with open(os.path.join(GOLD, "pointnet_manifest.json"), "w") as f:
        json.dump(manifest, f)
    print("pointnet", rec["logits"])


DG_CFG = "configs_reid/_base_/reidentifiers/reid_pts_dgcnn_point-cat.py"


class _CpuTorch:
    """dgcnn_orig.get_graph_feature hard-codes torch.device('cuda') (dgcnn_orig.py:38); there is no GPU in the
    development container, so for fixture generation that module sees a torch whose device() answers 'cpu'."""

    def __getattr__(self, name):
        return getattr(torch, name)

    @staticmethod
    def device(*a, **k):
        return torch.device("cpu")


def gen_dgcnn():
    ref = ref_loader.load_reference()
    ref.dgcnn_orig.torch = _CpuTorch()
    model, manifest = build(DG_CFG, seed=0)
    s1, s2 = T.synthetic_pairs(2, 256, seed=1, kind="randn")
    knns = []
    orig_knn = ref.dgcnn_orig.knn

    def spy(x, k):
        idx = orig_knn(x, k)
        knns.append(_np(idx).astype(np.int16))
        return idx

    ref.dgcnn_orig.knn = spy
    try:
        with torch.no_grad(), contextlib.redirect_stdout(io.StringIO()):
            xyz1, xyz2, h1, h2 = model.siamese_forward(s1, s2)
            logits = model.match_forward_inference(h1, h2, xyz1, xyz2)
            knns_fwd = list(knns)
            _, feat = model.backbone(torch.cat([s1, s2], 0).permute(0, 2, 1), model.backbone_list)
    finally:
        ref.dgcnn_orig.knn = orig_knn
    rec = dict(h1=_np(h1), h2=_np(h2), logits=_np(logits),
               enc_max=_np(feat.max(dim=2)[0]), enc_mean=_np(feat.mean(dim=2)),
               meta=np.array(json.dumps(dict(pairs=2, n=256, kind="randn", input_seed=1, weight_seed=0, k=20))))
    for i, kk in enumerate(knns_fwd[:4]):
        rec["knn%d" % (i + 1)] = kk
    np.savez_compressed(os.path.join(GOLD, "dgcnn_n256_randn.npz"), **rec)
    with open(os.path.join(GOLD, "dgcnn_manifest.json"), "w") as f:
        json.dump(manifest, f)
    print("dgcnn", rec["logits"])


LOCAL = dict(type="local_self_attention", d_model=64, nhead=2, attention="linear", knum=48, pos_size=64)


def gen_xcorr():
    """Point-Transformer with the 'baseline-orig' matching (cross -> local_self_attention -> cross -> local;
    configs_reid/_base_/reidentifiers/reid_pts_point-transformer_baseline_orig.py), 4 pairs of 128 points"""
    ref = ref_loader.load_reference()
    ref.attention.torch = _CpuTorch()        # attention.get_graph_feature hard-codes torch.device('cuda') (:114)
    model, manifest = build(PT_CFG, seed=0, match_type="xcorr", local_stage1=dict(LOCAL), local_stage2=dict(LOCAL),
                            backbone_list=[128, 64, 32])
    s1, s2 = T.synthetic_pairs(4, 128, seed=1, kind="randn")
    knns = []
    orig_knn = ref.attention.knn

    def spy(x, k):
        idx = orig_knn(x, k)
        knns.append(_np(idx).astype(np.int16))
        return idx

    ref.attention.knn = spy
    try:
        with torch.no_grad(), contextlib.redirect_stdout(io.StringIO()):
            xyz1, xyz2, h1, h2 = model.siamese_forward(s1, s2)
            a = model.cross_stage1(h1, xyz1, h2, xyz2)
            b = model.local_stage1(a, xyz1)
            knn_b = knns[-1]
            logits = model.match_forward_inference(h1, h2, xyz1, xyz2)
    finally:
        ref.attention.knn = orig_knn
    rec = dict(h1=_np(h1), h2=_np(h2), xc_a=_np(a), xc_b=_np(b), knn_local1=knn_b, logits=_np(logits),
               meta=np.array(json.dumps(dict(pairs=4, n=128, kind="randn", input_seed=1, weight_seed=0, knum=48,
                                             backbone_list=[128, 64, 32]))))
    np.savez_compressed(os.path.join(GOLD, "pt_xcorr_n128_randn.npz"), **rec)
    with open(os.path.join(GOLD, "pt_xcorr_manifest.json"), "w") as f:
        json.dump(manifest, f)
    print("xcorr", rec["logits"])


def gen_train_step():
    """reference train_step loss + a few gradients (training rows are 'next')."""
    ref_loader.load_reference()
    model, _ = build(PT_CFG, seed=0, backbone_list=[128, 64, 32],
                     losses_to_use=dict(kl=False, match=True, cls=False, shape=False, fp=False, triplet=False))
    model.train()
    s1, s2 = T.synthetic_pairs(8, 128, seed=2, kind="randn")
    ids1 = torch.arange(8)
    ids2 = torch.tensor([0, 1, 2, 3, 9, 9, 9, 9])
    data = dict(sparse_1=list(s1), sparse_2=list(s2), dense_1=list(s1), dense_2=list(s2),
                label_1=[torch.zeros(1, dtype=torch.long)] * 8, label_2=[torch.zeros(1, dtype=torch.long)] * 8,
                id_1=[i.view(1) for i in ids1], id_2=[i.view(1) for i in ids2])
    with contextlib.redirect_stdout(io.StringIO()):
        out = model.train_step(data, None)
    out["loss"].backward()
    rec = dict(loss=np.float32(out["loss"].item()),
               match_acc=np.float32(out["log_vars"]["match_acc"]))
    no_grad = []
    for k, p in model.named_parameters():
        if p.grad is None:
            no_grad.append(k)
    for k in ("match_head.1.weight", "cross_stage2.merge.weight", "backbone.cov_final.weight",
              "backbone.SA_modules.0.mlp_convs.0.weight", "backbone.SA_modules.2.self_attention.q_proj.weight"):
        rec["grad:" + k] = _np(dict(model.named_parameters())[k].grad)
    rec["no_grad_params"] = np.array(json.dumps(no_grad))
    # the same step in float64 (see gen_train_variants): the yardstick for the gradients in front of a max-over-K
    model64, _ = build(PT_CFG, seed=0, backbone_list=[128, 64, 32],
                       losses_to_use=dict(kl=False, match=True, cls=False, shape=False, fp=False, triplet=False))
    model64 = model64.double().train()
    data64 = {k: [t.double() if t.is_floating_point() else t for t in v] for k, v in data.items()}
    with contextlib.redirect_stdout(io.StringIO()):
        out64 = model64.train_step(data64, None)
    out64["loss"].backward()
    p64 = dict(model64.named_parameters())
    for k in list(rec):
        if k.startswith("grad:"):
            rec["grad64:" + k[5:]] = p64[k[5:]].grad.numpy().astype(np.float32)
    np.savez_compressed(os.path.join(GOLD, "pt_train_step_n128.npz"), **rec)
    print("train_step loss", rec["loss"], "params without grad:", len(no_grad))

STNET_CFG = "configs_reid/_base_/reidentifiers/reid_pts_point-transformer_baseline_stnet.py"


def _train_data(pairs, n, seed=2):
    s1, s2 = T.synthetic_pairs(pairs, n, seed=seed, kind="randn")
    ids1 = torch.arange(pairs)
    ids2 = torch.where(torch.arange(pairs) < pairs // 2, ids1, ids1 + 100)      # first half of the pairs match
    zero = torch.zeros(1, dtype=torch.long)
    return dict(sparse_1=list(s1), sparse_2=list(s2), dense_1=list(s1), dense_2=list(s2),
                label_1=[zero] * pairs, label_2=[zero] * pairs,
                id_1=[i.view(1) for i in ids1], id_2=[i.view(1) for i in ids2])


def gen_train_variants(only=None):
    """the reference's own train_step (loss, accuracy, gradients of a spread of parameters, which parameters get none)
    for the OTHER training configs it ships: the mul = 2 Point-Transformer
    (reid_waymo_pts/pts_point-transformer_point-cat_waymo_det_4x256_400e_512pts_2.py:25 -> the 1.5M reidentifier), the
    `concat` + channel-max baseline, the `xcorr-baseline` matching, PointNet and DGCNN; 8 pairs (4 for the wide
    encoders) of 128 points, seeded weights, BatchNorm in batch-statistics mode"""
    ref = ref_loader.load_reference()
    ref.dgcnn_orig.torch = _CpuTorch()
    ref.attention.torch = _CpuTorch()        # attention.get_graph_feature hard-codes torch.device('cuda') (:114)
    losses = dict(kl=False, match=True, cls=False, shape=False, fp=False, triplet=False)
    cases = (("pt15m", MUL_CFGS[0][1], dict(), 4, 128),
             ("baseline", BASE_CFG, dict(shape_head=None), 8, 128),
             # (reid_pts_point-transformer_baseline_stnet.py = the point-cat file + match_type 'xcorr-baseline'; the loader
             # executes plain config files only, so the `_base_` merge is spelled out)
             ("stnet", PT_CFG, dict(match_type="xcorr-baseline"), 8, 128),
             ("pointnet", PN_CFG, dict(), 4, 128),
             ("dgcnn", DG_CFG, dict(), 4, 128),
             # (reid_pts_point-transformer_baseline_orig.py = the point-cat file + match_type 'xcorr' + the two
             # local_self_attention stages: reid_waymo_pts/pts_point-transformer_baseline-orig_waymo_det_4x256_400e.py)
             ("orig", PT_CFG, dict(match_type="xcorr", local_stage1=dict(LOCAL), local_stage2=dict(LOCAL)), 8, 128))
    for tag, cfg, over, pairs, n in cases:
        if only and tag not in only:
            continue
        # (deep copies: the reference's build_module deletes cfg['type'] from the dict it is given)
        model, manifest = build(cfg, seed=0, losses_to_use=losses, **copy.deepcopy(over))
        if "backbone_list" in over or tag in ("pt15m", "baseline", "stnet", "orig"):
            model.backbone_list = [128, 64, 32]
        model.train()
        data = _train_data(pairs, n)
        with contextlib.redirect_stdout(io.StringIO()):
            out = model.train_step(data, None)
        out["loss"].backward()
        params = dict(model.named_parameters())
        rec = dict(loss=np.float32(out["loss"].item()), match_acc=np.float32(out["log_vars"]["match_acc"]))
        with_grad = [k for k, p in params.items() if p.grad is not None]
        # a spread of gradients: first / last tensors with a gradient, every 7th in between, capped in size
        pick = sorted(set(with_grad[:3] + with_grad[-3:] + with_grad[::7]))
        for k in pick:
            g = params[k].grad
            if g.numel() <= 40000:
                rec["grad:" + k] = _np(g)
        rec["no_grad_params"] = np.array(json.dumps([k for k, p in params.items() if p.grad is None]))
        rec["grad_norm"] = np.float32(float(torch.sqrt(sum((params[k].grad.double() ** 2).sum() for k in with_grad))))
        # the same step of the same reference model in float64 ("grad64:"): how far the reference's OWN float32
        # gradients are from the exact ones.  Tensors in front of a max-over-K / BatchNorm batch statistics move by
        # 1e-3 .. 1e-2 of their scale between float32 and float64 (a near-tie resolved the other way re-routes a whole
        # gradient row), tensors behind the last max by ~1e-6: the yardstick the GPU test holds the HIP gradients to
        model64, _ = build(cfg, seed=0, losses_to_use=losses, **copy.deepcopy(over))
        if tag in ("pt15m", "baseline", "stnet", "orig"):
            model64.backbone_list = [128, 64, 32]
        model64 = model64.double().train()
        data64 = {k: [t.double() if t.is_floating_point() else t for t in v] for k, v in data.items()}
        with contextlib.redirect_stdout(io.StringIO()):
            out64 = model64.train_step(data64, None)
        out64["loss"].backward()
        p64 = dict(model64.named_parameters())
        rec["loss64"] = np.float64(out64["loss"].item())
        for k in list(rec):
            if k.startswith("grad:"):
                rec["grad64:" + k[5:]] = p64[k[5:]].grad.numpy().astype(np.float32)   # (exact value rounded once)
        bn = [(k, b) for k, b in model.named_buffers() if k.endswith("running_mean")]
        for k, b in bn[:2] + bn[-1:]:
            rec["buf:" + k] = _np(b)
        rec["meta"] = np.array(json.dumps(dict(pairs=pairs, n=n, input_seed=2, weight_seed=0, cfg=cfg)))
        np.savez_compressed(os.path.join(GOLD, "train_step_%s_n%d.npz" % (tag, n)), **rec)
        # (state_dict manifests: the inference fixtures' -- pt15m / pt_baseline / pt / pointnet / dgcnn _manifest.json)
        print("train variant", tag, "loss", rec["loss"], "grads", len(with_grad), "recorded",
              sum(1 for k in rec if k.startswith("grad:")), "no grad", len(params) - len(with_grad))


def _train_loop_run(dtype, iters=5, lr0=1e-3, clip=1.0):
    from pcr_amd import train as TR
    model, _ = build(PT_CFG, seed=0, backbone_list=[128, 64, 32],
                     losses_to_use=dict(kl=False, match=True, cls=False, shape=False, fp=False, triplet=False))
    model = model.to(dtype).train()
    s1, s2 = T.synthetic_pairs(8, 128, seed=2, kind="randn")
    s1, s2 = s1.to(dtype), s2.to(dtype)
    ids1 = torch.arange(8)
    ids2 = torch.tensor([0, 1, 2, 3, 9, 9, 9, 9])
    data = dict(sparse_1=list(s1), sparse_2=list(s2), dense_1=list(s1), dense_2=list(s2),
                label_1=[torch.zeros(1, dtype=torch.long)] * 8, label_2=[torch.zeros(1, dtype=torch.long)] * 8,
                id_1=[i.view(1) for i in ids1], id_2=[i.view(1) for i in ids2])
    opt = torch.optim.AdamW(model.parameters(), lr=lr0, weight_decay=0.01, betas=(0.9, 0.999))
    losses, norms = [], []
    for it in range(iters):
        lr = TR.cyclic_value(lr0, it, 10)
        b1 = TR.cyclic_value(0.9, it, 10, target_ratio=(0.85 / 0.95, 1.0))
        for g in opt.param_groups:
            g["lr"], g["betas"] = lr, (b1, 0.999)
        opt.zero_grad(set_to_none=True)
        with contextlib.redirect_stdout(io.StringIO()):
            out = model.train_step(data, None)
        out["loss"].backward()
        norms.append(float(torch.nn.utils.clip_grad_norm_([p for p in model.parameters() if p.grad is not None], clip)))
        opt.step()
        losses.append(float(out["loss"].item()))
    model.eval()
    with torch.no_grad(), contextlib.redirect_stdout(io.StringIO()):
        xyz1, xyz2, h1, h2 = model.siamese_forward(s1, s2)
        logits = model.match_forward_inference(h1, h2, xyz1, xyz2)
    return model, losses, norms, logits


def gen_train_loop():
    """five iterations of the REFERENCE model (its own train_step: forward with BatchNorm batch statistics, autograd
    backward, running statistics) under torch's clip_grad_norm_ + AdamW with the cyclic lr / beta1 values of the
    reference's schedule config: the loss trajectory, the eval-mode logits after training and the BatchNorms' running
    statistics.  Pins the multi-step behaviour (update applied, statistics tracked, caches refreshed), which the
    one-step fixture cannot see.  The SAME loop is recorded in float64 too (keys ending in 64): this training problem
    amplifies rounding differences from step to step -- the reference's own float32 run is 5 % (step 4) and 7 % (step 5)
    away from its float64 run in the loss -- and that divergence, not a guess, is the yardstick the GPU test uses.
    (Measured while writing this: freezing the conv biases in front of a BatchNorm, whose true gradient is zero, changes
    neither trajectory: their walk under AdamW is not what separates two float32 runs.)"""
    ref_loader.load_reference()
    iters, lr0, clip = 5, 1e-3, 1.0
    model, losses, norms, logits = _train_loop_run(torch.float32, iters, lr0, clip)
    model64, losses64, norms64, logits64 = _train_loop_run(torch.float64, iters, lr0, clip)
    rec = dict(losses=np.array(losses, dtype=np.float64), grad_norms=np.array(norms, dtype=np.float64),
               logits=_np(logits), head_weight=_np(model.match_head[1].weight), iters=np.int32(iters), lr=np.float64(lr0),
               clip=np.float64(clip), max_iters=np.int32(10),
               losses64=np.array(losses64, dtype=np.float64), grad_norms64=np.array(norms64, dtype=np.float64),
               logits64=_np(logits64).astype(np.float32))
    for i, (sa, sa64) in enumerate(zip(model.backbone.SA_modules, model64.backbone.SA_modules)):
        for j, (bn, bn64) in enumerate(zip(sa.mlp_bns, sa64.mlp_bns)):
            rec["bn%d%d_mean" % (i, j)] = _np(bn.running_mean)
            rec["bn%d%d_var" % (i, j)] = _np(bn.running_var)
            rec["bn%d%d_n" % (i, j)] = np.int64(int(bn.num_batches_tracked))
            rec["bn%d%d_mean64" % (i, j)] = _np(bn64.running_mean).astype(np.float32)
            rec["bn%d%d_var64" % (i, j)] = _np(bn64.running_var).astype(np.float32)
    np.savez_compressed(os.path.join(GOLD, "pt_train_loop_n128.npz"), **rec)
    print("train loop losses", losses, "float64", losses64)


def gen_python_twins():
    """Python twins of the dormant CUDA ops (pointnet2_utils.py:116-240) where semantics
    coincide with the .cu kernels: FPS with the start index forced to 0 (power-of-two N <= 1024,
    no exact ties), ball query on strictly-inside radii, kNN as sorted sets."""
    ref = ref_loader.load_reference()
    p2u = ref.pointnet2_utils
    xyz = T.synthetic_clouds(4, 256, seed=7, kind="randn")
    orig = torch.randint
    torch.randint = lambda lo, hi, size, **kw: torch.zeros(size, dtype=torch.long)
    try:
        fps = p2u.farthest_point_sample(xyz, 64)
    finally:
        torch.randint = orig
    centres = p2u.index_points(xyz, fps)
    ball = p2u.query_ball_point(0.6, 16, xyz, centres)
    knn = p2u.knn_point(16, xyz, centres)
    np.savez_compressed(os.path.join(GOLD, "ops_python_twins.npz"),
                        fps=_np(fps).astype(np.int32), ball=_np(ball).astype(np.int32),
                        knn_sorted=np.sort(_np(knn), -1).astype(np.int32),
                        meta=np.array(json.dumps(dict(clouds=4, n=256, seed=7, kind="randn", m=64,
                                                      radius=0.6, nsample=16))))
    print("twins", fps.shape, ball.shape, knn.shape)



def cuda_semantics_cases():
    """the hand-built tie / boundary inputs of tests/golden/ops_cuda_semantics.npz: integer or dyadic coordinates, so
    that every distance is exact in binary32 and the ties are real (no dependence on fma contraction)"""
    rng = np.random.RandomState(20261005)
    cases = {}
    # ---- FPS (furthest_point_sample_cuda.cu:17-23, 56-71): non-power-of-two N, N > 1024, exact ties, duplicates ----
    a = np.zeros((1, 12, 3), np.float32)
    a[0, 1:, 0] = 1.0                                             # test_oracle_ops' [0, 8, 0] case
    b = rng.randint(0, 3, size=(2, 12, 3)).astype(np.float32)     # block 8, heavy duplication
    lat = np.stack(np.meshgrid(np.arange(5), np.arange(5), np.arange(4), indexing="ij"), -1).reshape(-1, 3)
    c = np.stack([lat[rng.permutation(100)], lat[rng.permutation(100)]]).astype(np.float32)      # N = 100: block 64
    d = rng.randint(0, 14, size=(1, 3000, 3)).astype(np.float32)  # N = 3000: block 1024, three strided rounds per thread
    e = rng.randint(0, 6, size=(1, 1024, 3)).astype(np.float32)   # power of two, duplicates everywhere
    for name, xyz, m in (("fps_n12", a, 3), ("fps_n12_dup", b, 9), ("fps_n100", c, 24), ("fps_n3000", d, 40),
                         ("fps_n1024_dup", e, 32)):
        cases[name] = dict(kind="fps", xyz=xyz, m=m)
    dl = c[0][:, None, :] - c[0][None, :, :]
    cases["fpsd_n100"] = dict(kind="fps_dist", dist=(dl * dl).sum(-1)[None].astype(np.float32), m=16)
    cases["fpsd_n40_asym"] = dict(kind="fps_dist", dist=rng.randint(0, 9, size=(2, 40, 40)).astype(np.float32), m=12)
    # ---- ball query (ball_query_cuda.cu:38-52): d2 == max_r^2 excluded, d2 == min_r^2 included, d2 == 0 always
    # included (also below min_r), first nsample by index, padded with the first hit, zeros when nothing is in range
    # (min_r < max_r always: BallQuery.forward asserts it, ball_query.py:31) ----
    g = (np.stack(np.meshgrid(np.arange(4), np.arange(4), np.arange(4), indexing="ij"), -1).reshape(-1, 3) * 0.25)
    g = g[rng.permutation(64)].astype(np.float32)
    pts = np.concatenate([g, g[:8]], 0)[None]                     # 72 points, the first 8 lattice points twice
    ctr = np.concatenate([g[:10], np.array([[9, 9, 9], [0.125, 0.125, 0.125]], np.float32)], 0)[None]
    for name, lo, hi, k in (("bq_r050", 0.0, 0.5, 5), ("bq_r100_k16", 0.0, 1.0, 16), ("bq_min050_r075", 0.5, 0.75, 6),
                            ("bq_min025_r030", 0.25, 0.3, 8)):
        cases[name] = dict(kind="ball", xyz=pts, centres=ctr, min_r=lo, max_r=hi, k=k)
    # ---- heap kNN (knn_cuda.cu:27-94): equal distances leave the heap in the order its sift sequence produces ----
    for name, k, nq in (("knn_k8", 8, 12), ("knn_k1", 1, 12), ("knn_k64_all", 64, 4), ("knn_k100", 100, 3)):
        cases[name] = dict(kind="knn", xyz=pts if k <= 72 else np.concatenate([pts, pts], 1), centres=ctr[:, :nq], k=k)
    # ---- three_nn (three_nn_cuda.cu:11-65): strict '<' cascade on equal distances, float d against double bests ----
    cases["nn3_lattice"] = dict(kind="three_nn", unknown=ctr, known=pts)
    cases["nn3_three_known"] = dict(kind="three_nn", unknown=ctr, known=pts[:, :3])
    cases["nn3_two_known"] = dict(kind="three_nn", unknown=ctr, known=pts[:, :2])     # the third best stays 1e40 -> +inf, index 0
    return cases


def gen_cuda_semantics():
    """tests/golden/ops_cuda_semantics.npz: expected outputs of the reference's dormant CUDA ops on tie / boundary inputs
    the Python twins cannot cover, produced by oracle/cuda_sim.py -- a thread-faithful simulation of the .cu kernels'
    execution (block / tid loops, shared arrays, barrier phases), independent of the C oracle.  Needs no reference import:
    the kernels are CUDA and cannot run anywhere in this build; the simulator IS the second transliteration."""
    import cuda_sim as S
    out = {}
    meta = {}
    for name, c in cuda_semantics_cases().items():
        kind = c["kind"]
        if kind == "fps":
            out[name + "_xyz"], out[name + "_idx"] = c["xyz"], S.fps(c["xyz"], c["m"])
            meta[name] = dict(kind=kind, m=c["m"], block=S.launch_block_size(c["xyz"].shape[1]))
        elif kind == "fps_dist":
            out[name + "_dist"], out[name + "_idx"] = c["dist"], S.fps_with_dist(c["dist"], c["m"])
            meta[name] = dict(kind=kind, m=c["m"], block=S.launch_block_size(c["dist"].shape[1]))
        elif kind == "ball":
            out[name + "_xyz"], out[name + "_centres"] = c["xyz"], c["centres"]
            out[name + "_idx"] = S.ball_query(c["min_r"], c["max_r"], c["k"], c["xyz"], c["centres"])
            meta[name] = dict(kind=kind, min_r=c["min_r"], max_r=c["max_r"], k=c["k"])
        elif kind == "knn":
            out[name + "_xyz"], out[name + "_centres"] = c["xyz"], c["centres"]
            out[name + "_idx"], out[name + "_d2"] = S.knn(c["k"], c["xyz"], c["centres"])
            meta[name] = dict(kind=kind, k=c["k"])
        else:
            out[name + "_unknown"], out[name + "_known"] = c["unknown"], c["known"]
            out[name + "_d2"], out[name + "_idx"] = S.three_nn(c["unknown"], c["known"])
            meta[name] = dict(kind=kind)
        print("cuda semantics", name, meta[name])
    np.savez_compressed(os.path.join(GOLD, "ops_cuda_semantics.npz"), meta=np.array(json.dumps(meta)), **out)


def gen_eval_metric():
    """val_match_acc (reidentification_base.py:104) and the reference's own MatchingEval.f1_precision_recall
    (datasets/utils.py:254-277) on a seeded (logits, gt) sample"""
    g = np.random.default_rng(5)
    logits = g.standard_normal(64).astype(np.float32)
    gt = (g.uniform(size=64) > 0.5).astype(np.float32)
    tl, tg = torch.from_numpy(logits), torch.from_numpy(gt)
    acc = (torch.sigmoid(tl) > 0.5).float().eq(tg).float().mean()
    me = ref_loader.load_dataset_utils().MatchingEval()
    f1 = me.f1_precision_recall((torch.sigmoid(tl) > 0.5).float(), tg)
    np.savez_compressed(os.path.join(GOLD, "eval_metric.npz"), logits=logits, gt=gt, val_match_acc=np.float32(acc),
                        **{k: np.float32(v) for k, v in f1.items()})
    print("eval metric", float(acc), f1)


def gen_eval_tables():
    """MatchingEval.evaluate_points / evaluate_distance / eval_per_visibility (datasets/utils.py:280-533) of the imported
    reference on a seeded sample: every table entry, flattened to 'family/key/metric' -> value"""
    g = np.random.default_rng(9)
    n = 200
    logits = torch.from_numpy(g.standard_normal(n).astype(np.float32))
    gt = torch.from_numpy((g.uniform(size=n) > 0.5).astype(np.float32))
    num_points = torch.from_numpy(g.integers(1, 700, (n, 2)))
    vis = torch.from_numpy(g.integers(0, 4, (n, 2)))
    dist = torch.from_numpy(g.integers(0, 60, (n, 2)))
    gt_fp = gt.clone()
    gt_fp[::17] = -1            # false-positive pairs are excluded from the visibility tables
    me = ref_loader.load_dataset_utils().MatchingEval()

    def flat(t):
        out = {}
        for fam, rows in t.items():
            for key, entry in rows.items():
                for m, v in entry.items():
                    out["%s/%s/%s" % (fam, key, m)] = float(v)
        return out
    rec = dict(logits=logits.numpy(), gt=gt.numpy(), gt_fp=gt_fp.numpy(), num_points=num_points.numpy(), vis=vis.numpy(),
               dist=dist.numpy())
    for name, t in (("points", me.evaluate_points(logits, gt, num_points)), ("distance", me.evaluate_distance(logits, gt, dist)),
                    ("visibility", me.eval_per_visibility(logits, gt_fp, vis))):
        f = flat(t)
        rec[name + "_keys"] = np.array(json.dumps(sorted(f)))
        rec[name + "_vals"] = np.array([f[k] for k in sorted(f)], dtype=np.float64)
        print("tables", name, len(f))
    np.savez_compressed(os.path.join(GOLD, "eval_tables.npz"), **rec)


def gen_pairs():
    """the reference's dataset classes (reidentification_nuscenes.py:16-72 train, :78-145 FPVal, :150-249 FPValEven over
    reidentification_base.py and object_loader_base.py; oracle/ref_datasets.py) on a toy object table whose crops are
    rebuilt from seeds: the order of `idx` after the constructor's shuffle, three passes of training items (every tensor
    of the returned dict), and both validation pair sets -- what pcr_amd/loader.py TrainPairs and pcr_amd/pairs.py must
    reproduce call for call under the same numpy seed."""
    import tempfile
    import ref_datasets as RD
    NS, ND, SEED, MAXC = 32, 16, 5, 3
    objs = RD.toy_objects(seed=1, n_true=26, n_fp=12)
    tmp = tempfile.mkdtemp()
    root = os.path.join(tmp, "crops")
    RD.write_crops(root, objs)
    rec = dict(meta=np.array(json.dumps(dict(ns=NS, nd=ND, seed=SEED, item_seed=11, max_combinations=MAXC, toy_seed=1,
                                             n_true=26, n_fp=12, passes=3))),
               objects=np.array(json.dumps([dict(token=o["token"], class_name=o["class_name"], fp=o["fp"],
                                                 frames={str(k): v for k, v in o["frames"].items()},
                                                 visibility={str(k): v for k, v in o["visibility"].items()}) for o in objs])))
    ds = RD.build_reference_dataset("train", objs, root, tmp, NS, ND, seed=SEED)
    rec["train_idx"] = np.asarray(ds.idx, dtype=np.int64)
    rec["train_classes"] = np.asarray(ds.classes, dtype=np.int64)
    np.random.seed(11)
    items = {k: [] for k in ("sparse_1", "sparse_2", "dense_1", "dense_2", "label_1", "label_2", "id_1", "id_2")}
    for _ in range(3):
        for i in range(len(ds)):
            it = ds[i]
            for k in items:
                items[k].append(it[k].data.numpy())
    for k, v in items.items():
        rec["train_" + k] = np.stack(v)
    rec["train_rng_after"] = np.random.randint(0, 2 ** 31 - 1, size=4)       # the generator's state after the passes
    print("train items", len(items["id_1"]), "negatives", int((rec["train_id_1"] != rec["train_id_2"]).sum()),
          "fp", int((rec["train_id_2"] == -1).sum()))
    toks = [o["token"] for o in objs]

    def pairs(ps, ns_):
        pos = np.array([[toks.index(p["tok"]), p["o1"], p["o2"], p["cls"]] for p in ps], dtype=np.int64)
        neg = np.array([[toks.index(n["tok1"]), n["o1"], toks.index(n["tok2"]), n["o2"], n["cls1"], n["cls2"]]
                        for n in ns_], dtype=np.int64)
        return pos, neg
    for kind in ("val", "val_even"):
        dv = RD.build_reference_dataset(kind, objs, root, tmp, NS, ND, seed=SEED, max_combinations=MAXC)
        pos, neg = pairs(dv.val_positives, dv.val_negatives)
        rec[kind + "_idx"] = np.asarray(dv.idx, dtype=np.int64)
        rec[kind + "_pos"], rec[kind + "_neg"] = pos, neg
        print(kind, "positives", len(pos), "negatives", len(neg), "fp negatives", int((neg[:, 5] >= 2).sum()),
              "self-paired negatives", int((neg[:, 0] == neg[:, 2]).sum()))
        # two items of the validation set (a positive and a negative) with the size / visibility keys
        np.random.seed(3)
        for name, j in (("p", 0), ("n", len(pos))):
            it = dv[j]
            for k, v in it.items():
                rec["%s_item_%s_%s" % (kind, name, k)] = v.data.numpy()
    np.savez_compressed(os.path.join(GOLD, "pairs_toy.npz"), **rec)


if __name__ == "__main__":
    os.makedirs(GOLD, exist_ok=True)
    if "--only-pairs" in sys.argv:
        gen_pairs()
        sys.exit(0)
    if "--only-tables" in sys.argv:
        gen_eval_tables()
        sys.exit(0)
    torch.set_num_threads(8)
    if "--only-metric" in sys.argv:
        gen_eval_metric()
        sys.exit(0)
    if "--only-xcorr" in sys.argv:
        gen_xcorr()
        sys.exit(0)
    if "--only-dgcnn" in sys.argv:
        gen_dgcnn()
        sys.exit(0)
    if "--only-baseline" in sys.argv:
        gen_baseline()
        sys.exit(0)
    if "--only-train-loop" in sys.argv:
        gen_train_loop()
        sys.exit(0)
    if "--only-cuda-semantics" in sys.argv:
        gen_cuda_semantics()
        sys.exit(0)
    if "--only-mul" in sys.argv:
        gen_pt_mul()
    if "--only-train-step" in sys.argv:
        gen_train_step()
        sys.exit(0)
    if "--only-train-variants" in sys.argv:
        gen_train_variants([a for a in sys.argv[1:] if not a.startswith("--")] or None)
        sys.exit(0)
        sys.exit(0)
    if "--only-small" not in sys.argv:
        gen_baseline()
        gen_pt_mul()
        gen_dgcnn()
        gen_xcorr()
        gen_pt()
        gen_pointnet()
    gen_train_step()
    gen_train_variants()
    gen_train_loop()
    gen_python_twins()
    gen_cuda_semantics()
    gen_eval_metric()
    gen_eval_tables()
    gen_pairs()
