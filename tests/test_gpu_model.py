"""GPU parity of the Point-Transformer ReIDNet (HIP path through mmdet3d.models -> engine ->
C ABI) against (a) the golden vectors recorded from the imported reference and (b) the torch
CPU restatement (oracle/model_oracle.py) at larger seeded sizes.

Tolerance: north_star asks for embeddings/logits within 1e-4 (fp32) on identical inputs."""
import copy
import json
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, load_golden
from pcr_amd import testing as T

pytestmark = pytest.mark.gpu
TOL = 1e-4

PT_MODEL = dict(
    type="ReIDNet", hidden_size=128, combine="point-cat", match_type="xcorr_eff", pool_type="both",
    backbone_list=[128, 64, 32], output_sequence_size=64,
    backbone=dict(type="Pointnet_Backbone", input_channels=0, use_xyz=True, conv_out=64),
    match_head=[dict(type="LinearRes", n_in=128, n_out=128, norm="GN", ng=8),
                dict(type="Linear", in_features=128, out_features=1)],
    downsample=None, cls_head=None, fp_head=None, shape_head=None,
    cross_stage1=dict(type="corss_attention", d_model=64, nhead=2, attention="linear"),
    cross_stage2=dict(type="corss_attention", d_model=64, nhead=2, attention="linear"),
    local_stage1=dict(), local_stage2=dict(),
    losses_to_use=dict(kl=False, match=True, cls=False, shape=False, fp=False, triplet=False))


def build_pt(backbone_list):
    from mmdet3d.models import build_model
    cfg = copy.deepcopy(PT_MODEL)
    cfg["backbone_list"] = list(backbone_list)
    m = build_model(cfg)
    sd = T.seeded_state_dict(T.load_manifest(os.path.join(GOLDEN, "pt_manifest.json")), 0)
    m.load_state_dict(sd, strict=True)
    return m.cuda().eval(), sd


def run_stages(m, s1, s2, fused_final=True):
    st = {}
    hooks = []
    bb = m.backbone
    if not fused_final:
        bb.FP_modules[0].interpolation.fuse_final_conv(None)
    for i, sa in enumerate(bb.SA_modules):
        hooks.append(sa.self_attention.register_forward_pre_hook(
            lambda mod, args, i=i: st.__setitem__(f"sa{i}_mlp", args[0].cpu().numpy())))
        hooks.append(sa.register_forward_hook(
            lambda mod, args, out, i=i: st.__setitem__(f"sa{i}_out", out[1].cpu().numpy())))
    for j, fp in enumerate(bb.FP_modules):
        hooks.append(fp.register_forward_hook(
            lambda mod, args, out, j=j: st.__setitem__(f"fp{j}_out", out.cpu().numpy())))
    with torch.no_grad():
        xyz1, xyz2, h1, h2 = m.siamese_forward(s1.cuda(), s2.cuda())
        out, o1, o2 = m.xcorr_eff(h1, xyz1, h2, xyz2)
        pooled = m.get_pooled_feats(out)
        logits = m.match_forward_inference(h1, h2, xyz1, xyz2)
    torch.cuda.synchronize()
    for h in hooks:
        h.remove()
    if not fused_final:
        bb.FP_modules[0].interpolation.fuse_final_conv(bb.cov_final)
    st.update(h1=h1.cpu().numpy(), h2=h2.cpu().numpy(), x2_o1=o1.cpu().numpy(), x2_o2=o2.cpu().numpy(),
              pooled=pooled.cpu().numpy(), logits=logits.cpu().numpy())
    return st


def _report(st, ref, keys):
    worst = {}
    for k in keys:
        if k in st and k in ref:
            worst[k] = float(np.abs(st[k] - np.asarray(ref[k])).max())
    return worst


@pytest.mark.parametrize("case", ["pt_n128_randn", "pt_n128_dup", "pt_n256_box", "pt_n1024_randn"])
def test_pt_matches_reference_golden(case):
    g = load_golden(case)
    meta = g["meta"]
    m, _ = build_pt(meta["backbone_list"])
    s1, s2 = T.synthetic_pairs(meta["pairs"], meta["n"], meta["input_seed"], meta["kind"])
    st = run_stages(m, s1, s2, fused_final=False)
    keys = [k for k in g if k not in ("meta",) and not k.endswith("knn_sorted")]
    worst = _report(st, g, keys)
    print(case, json.dumps(worst))
    assert worst, "no stage compared"
    bad = {k: v for k, v in worst.items() if not v < TOL}
    assert not bad, bad
    # fused cov_final must give the same h
    st2 = run_stages(m, s1, s2, fused_final=True)
    assert np.abs(st2["h1"] - g["h1"]).max() < TOL and np.abs(st2["logits"] - g["logits"]).max() < TOL


def test_pt_knn_sets_match_reference_golden():
    from pcr_amd import engine
    for case in ("pt_n128_randn", "pt_n256_box", "pt_n1024_randn"):
        g = load_golden(case)
        meta = g["meta"]
        s1, s2 = T.synthetic_pairs(meta["pairs"], meta["n"], meta["input_seed"], meta["kind"])
        xyz = torch.cat([s1, s2], 0).cuda()
        bl = meta["backbone_list"]
        idx = engine.knn_prefix(xyz, bl[0], 32).cpu().numpy()
        assert (np.sort(idx, -1) == g["sa0_knn_sorted"]).all()
        idx = engine.knn_prefix(xyz[:, :bl[0]].contiguous(), bl[1], 48).cpu().numpy()
        assert (np.sort(idx, -1) == g["sa1_knn_sorted"]).all()
        idx = engine.knn_prefix(xyz[:, :bl[1]].contiguous(), bl[2], 48).cpu().numpy()
        assert (np.sort(idx, -1) == g["sa2_knn_sorted"]).all()


@pytest.mark.parametrize("pairs,n,bl,kind", [(5, 200, [200, 100, 50], "box"), (3, 512, [512, 256, 128], "dup"),
                                             (2, 1000, [1000, 500, 250], "randn")])
def test_pt_matches_cpu_oracle(pairs, n, bl, kind):
    """sizes with ragged tiles (N not a multiple of 32/64) against the torch restatement"""
    import model_oracle as MO
    m, sd = build_pt(bl)
    s1, s2 = T.synthetic_pairs(pairs, n, seed=77, kind=kind)
    ref = {}
    with torch.no_grad():
        MO.pt_pairs(sd, s1, s2, bl, stages=ref)
    ref = {k: v.numpy() for k, v in ref.items()}
    st = run_stages(m, s1, s2)
    worst = _report(st, ref, ["sa0_mlp", "sa0_out", "sa1_mlp", "sa1_out", "sa2_mlp", "sa2_out", "fp2_out", "fp1_out",
                              "h1", "h2", "x2_o1", "x2_o2", "pooled", "logits"])
    print(json.dumps(worst))
    bad = {k: v for k, v in worst.items() if not v < TOL}
    assert not bad, bad


def test_forward_test_api_and_decisions():
    """mmdet-style call: lists of per-sample tensors in, [dict] out (ReIDNet.forward_test)"""
    g = load_golden("pt_n128_randn")
    m, _ = build_pt([128, 64, 32])
    s1, s2 = T.synthetic_pairs(2, 128, 1, "randn")
    dev = "cuda"
    data = dict(sparse_1=list(s1.to(dev)), sparse_2=list(s2.to(dev)), dense_1=list(s1.to(dev)), dense_2=list(s2.to(dev)),
                label_1=[torch.zeros(1, dtype=torch.long, device=dev)] * 2, label_2=[torch.zeros(1, dtype=torch.long, device=dev)] * 2,
                id_1=[torch.tensor([3], device=dev), torch.tensor([4], device=dev)],
                id_2=[torch.tensor([3], device=dev), torch.tensor([5], device=dev)],
                size_1=[torch.tensor([128], device=dev)] * 2, size_2=[torch.tensor([128], device=dev)] * 2,
                vis_1=[torch.tensor([1], device=dev)] * 2, vis_2=[torch.tensor([1], device=dev)] * 2)
    with torch.no_grad():
        res = m(return_loss=False, rescale=True, **data)
    assert isinstance(res, list) and len(res) == 1
    r = res[0]
    assert np.abs(r["val_match_preds"].cpu().numpy() - g["logits"]).max() < TOL
    assert r["val_match_gt"].cpu().tolist() == [1.0, 0.0]
    assert r["match_classes"].shape == (2, 2) and r["num_points"].shape == (2, 2)
    assert ((torch.sigmoid(r["val_match_preds"]) > 0.5).cpu().numpy() == (1 / (1 + np.exp(-g["logits"])) > 0.5)).all()


def test_train_step_matches_reference_loss_and_grads(grad_floor):
    """ReIDNet.train_step in training mode (every forward / backward node a HIP launch: pcr_amd/train_graph.py,
    train_ops.py) against loss and gradients recorded from the reference's own train_step"""
    g = load_golden("pt_train_step_n128")
    m, _ = build_pt([128, 64, 32])
    m.train()
    s1, s2 = T.synthetic_pairs(8, 128, seed=2, kind="randn")
    dev = "cuda"
    ids1 = torch.arange(8)
    ids2 = torch.tensor([0, 1, 2, 3, 9, 9, 9, 9])
    data = dict(sparse_1=list(s1.to(dev)), sparse_2=list(s2.to(dev)), dense_1=list(s1.to(dev)), dense_2=list(s2.to(dev)),
                label_1=[torch.zeros(1, dtype=torch.long, device=dev)] * 8,
                label_2=[torch.zeros(1, dtype=torch.long, device=dev)] * 8,
                id_1=[i.view(1).to(dev) for i in ids1], id_2=[i.view(1).to(dev) for i in ids2])
    out = m.train_step(data, None)
    assert set(out) == {"loss", "log_vars", "num_samples"} and out["num_samples"] == 8
    assert abs(float(out["loss"]) - float(g["loss"])) < 1e-4
    assert abs(out["log_vars"]["match_acc"] - float(g["match_acc"])) < 1e-6
    out["loss"].backward()
    params = dict(m.named_parameters())
    worst, ref_own, vs64 = {}, {}, {}
    for k in g:
        if k.startswith("grad:"):
            got = params[k[5:]].grad.cpu().numpy()
            scale = max(1e-3, float(np.abs(g[k]).max()))
            worst[k[5:]] = float(np.abs(got - g[k]).max()) / scale
            ref_own[k[5:]] = float(np.abs(g[k] - g["grad64:" + k[5:]]).max()) / scale
            vs64[k[5:]] = float(np.abs(got - g["grad64:" + k[5:]]).max()) / scale
    print(json.dumps(dict(vs_ref32=worst, ref32_vs_ref64=ref_own, vs_ref64=vs64)))
    # Against the REFERENCE's own backward (CPU, float32): match head, cross attention, cov_final and the SA attention
    # projection agree to ~6e-7 of the tensor's scale (bound 1e-5).  The first SA conv's gradient passes through three
    # BatchNorm layers in batch-statistics mode and the max-over-K routing, where a near-tie between two rows of a group
    # resolved differently by two summation orders moves a whole gradient row: the reference's OWN float32 gradient of
    # that tensor is ~4e-3 of its scale away from the reference's float64 gradient (`grad64:` in the fixture, recorded
    # by oracle/make_golden.py gen_train_step).  That is the yardstick: the HIP gradient must be no further from the
    # float64 gradient than twice what the reference's float32 backward is.  (tests/test_gpu_train_ops.py pins the same
    # layer to 2e-4 against torch autograd on identical indices.)
    for k in worst:
        assert worst[k] < grad_floor or vs64[k] < max(grad_floor, 2.0 * ref_own[k]), (k, worst[k], vs64[k], ref_own[k])
    no_grad = sorted(k for k, p in params.items() if p.grad is None)
    assert no_grad == sorted(json.loads(str(g["no_grad_params"])))      # the 24 never-used FP tensors


def test_trainer_overfits_a_fixed_batch():
    """pcr_amd.train.Trainer (AdamW, cyclic lr / beta1, clipping, one gradient bucket) drives ReIDNet.train_step:
    the loss on a fixed batch goes down, only the live parameters get gradients, the bucket is the survey's 2.3 MB"""
    from pcr_amd import train
    m, _ = build_pt([128, 64, 32])
    m.train()
    s1, s2 = T.synthetic_pairs(8, 128, seed=2, kind="randn")
    dev = "cuda"
    ids1 = torch.arange(8)
    ids2 = torch.tensor([0, 1, 2, 3, 9, 9, 9, 9])
    data = dict(sparse_1=list(s1.to(dev)), sparse_2=list(s2.to(dev)), dense_1=list(s1.to(dev)), dense_2=list(s2.to(dev)),
                label_1=[torch.zeros(1, dtype=torch.long, device=dev)] * 8,
                label_2=[torch.zeros(1, dtype=torch.long, device=dev)] * 8,
                id_1=[i.view(1).to(dev) for i in ids1], id_2=[i.view(1).to(dev) for i in ids2])
    tr = train.Trainer(m, max_iters=12, lr=1e-3, grad_clip=1.0)
    losses = [float(tr.step(data)["loss"].detach()) for _ in range(12)]
    assert all(np.isfinite(losses)) and losses[-1] < 0.7 * losses[0], losses
    tr.bucket._layout()
    assert tr.bucket.nbytes() == 4 * 579425          # live gradient payload 2.32 MB


@pytest.mark.parametrize("fused", [False, True])
def test_training_loop_follows_the_reference_over_five_iterations(fused):
    """pcr_amd.train.Trainer on the HIP graph + HIP optimizer against five iterations of the REFERENCE model under
    torch's clip_grad_norm_ + AdamW (tests/golden/pt_train_loop_n128.npz, oracle/make_golden.py gen_train_loop): loss and
    gradient-norm trajectories, BatchNorm running statistics, and the eval-mode logits of the trained weights.

    This training problem amplifies rounding differences from step to step: the REFERENCE's own float32 trajectory is
    4e-5 (step 2), 7e-5 (step 3), 5 % (step 4) and 7 % (step 5) away from the same loop in float64 (losses64 in the
    fixture).  Two yardsticks, both built on that measured divergence:
    * fused = False (one launch per layer, the graph of rounds 2-4): the HIP run stays within HALF of it of the
      reference's float32 run (plus 1e-4 where the two references still agree) -- the pin of rounds 3-4, unchanged;
    * fused = True (the default since round 5: the attention blocks' per-token chains as one launch each way,
      csrc/train_chain_kernels.hip).  Its gradients are as close to float64 autograd as the unfused launches' and closer
      than torch's own float32 (measured per tensor: 2-4e-7 of scale, torch float32 1e-6 on the weights), yet a different
      summation order is a different sample of the same chaos, and the float32 reference is only ONE sample of it: the
      fused run is held to the FLOAT64 trajectory instead -- no further from it than 4x the reference's float32 run is
      (plus the same floors).  Measured: step 3 2.9e-4 from float64 (reference float32 7e-5), steps 4 / 5 0.013 / 0.040
      (reference float32 0.091 / 0.070: the fused run is the closer one where the divergence is large)."""
    from pcr_amd import train, train_ops
    g = load_golden("pt_train_loop_n128")
    prev, train_ops.FUSED_CHAINS = train_ops.FUSED_CHAINS, fused
    try:
        m, _ = build_pt([128, 64, 32])
        m.train()
        s1, s2 = T.synthetic_pairs(8, 128, seed=2, kind="randn")
        dev = "cuda"
        ids1 = torch.arange(8)
        ids2 = torch.tensor([0, 1, 2, 3, 9, 9, 9, 9])
        data = dict(sparse_1=list(s1.to(dev)), sparse_2=list(s2.to(dev)), dense_1=list(s1.to(dev)), dense_2=list(s2.to(dev)),
                    label_1=[torch.zeros(1, dtype=torch.long, device=dev)] * 8,
                    label_2=[torch.zeros(1, dtype=torch.long, device=dev)] * 8,
                    id_1=[i.view(1).to(dev) for i in ids1], id_2=[i.view(1).to(dev) for i in ids2])
        tr = train.Trainer(m, max_iters=int(g["max_iters"]), lr=float(g["lr"]), grad_clip=float(g["clip"]))
        losses, norms = [], []
        for _ in range(int(g["iters"])):
            out = tr.step(data)
            losses.append(float(out["loss"].detach()))
            norms.append(float(out["grad_norm"]))
    finally:
        train_ops.FUSED_CHAINS = prev
    ref, ref64 = g["losses"], g["losses64"]
    print(json.dumps(dict(fused=fused, losses=losses, ref=ref.tolist(), ref64=ref64.tolist(), norms=norms,
                          ref_norms=g["grad_norms"].tolist(), ref_norms64=g["grad_norms64"].tolist())))

    def held(mine, r32, r64, floor):
        """the yardstick of the docstring for one quantity (arrays or scalars; max-norm)"""
        own = float(np.abs(np.asarray(r32, dtype=np.float64) - np.asarray(r64, dtype=np.float64)).max())
        if fused:
            d64 = float(np.abs(np.asarray(mine, dtype=np.float64) - np.asarray(r64, dtype=np.float64)).max())
            return d64 <= floor + 4.0 * own, (d64, own)
        d32 = float(np.abs(np.asarray(mine, dtype=np.float64) - np.asarray(r32, dtype=np.float64)).max())
        return d32 <= floor + 0.5 * own, (d32, own)
    for i in range(len(losses)):
        ok, why = held(losses[i], float(ref[i]), float(ref64[i]), 1e-4)
        assert ok, ("loss", i, losses[i], float(ref[i]), float(ref64[i]), why)
        ok, why = held(norms[i], float(g["grad_norms"][i]), float(g["grad_norms64"][i]), 1e-3 * norms[i])
        assert ok, ("grad norm", i, norms[i], why)
    worst, own_bn = {}, {}
    for i, sa in enumerate(m.backbone.SA_modules):
        for j, bn in enumerate(sa.mlp_bns):
            for nm, t in (("mean", bn.running_mean), ("var", bn.running_var)):
                r32, r64 = g["bn%d%d_%s" % (i, j, nm)], g["bn%d%d_%s64" % (i, j, nm)]
                sc = max(1e-3, np.abs(r32).max())
                key = "bn%d%d_%s" % (i, j, nm)
                worst[key] = float(np.abs(t.cpu().numpy() - (r64 if fused else r32)).max() / sc)
                own_bn[key] = float(np.abs(r32 - r64).max() / sc)
            assert int(bn.num_batches_tracked) == int(g["bn%d%d_n" % (i, j)])
    print(json.dumps(dict(worst=max(worst.values()), ref32_vs_ref64=max(own_bn.values()))))
    # running statistics: every tensor within 1e-4 + the yardstick's share of the reference's own float32-vs-float64
    # distance (2.3e-2 at most)
    for k in worst:
        assert worst[k] <= 1e-4 + (4.0 if fused else 0.5) * max(own_bn.values()), (k, worst[k], own_bn[k])
    m.eval()
    with torch.no_grad():
        logits = m.match_forward_inference(*_hx(m, s1.to(dev), s2.to(dev))).cpu().numpy()
    # eval-mode logits of the trained weights: the two references are 0.42 apart (the untrained model's logits are ~1
    # away); unfused: a quarter of that from the float32 reference; fused: no further from the float64 reference than the
    # float32 reference is
    own_l = float(np.abs(g["logits"] - g["logits64"]).max())
    dl = float(np.abs(logits - (g["logits64"] if fused else g["logits"])).max())
    print(json.dumps(dict(dlogits=dl, ref32_vs_ref64=own_l)))
    assert dl <= (1.0 if fused else 0.25) * own_l, (logits, g["logits"], g["logits64"])


def test_trainer_checkpoint_resumes_bit_for_bit(tmp_path):
    """mmcv-layout checkpoint through the HIP optimizer: two iterations, save, load into a FRESH model + Trainer, two
    more -- the losses and gradient norms of iterations 3 and 4 equal those of an uninterrupted run exactly (weights,
    BatchNorm statistics, AdamW moments and step counts, the schedule position all round-trip; every kernel is
    deterministic)"""
    from pcr_amd import train
    s1, s2 = T.synthetic_pairs(8, 128, seed=2, kind="randn")
    dev = "cuda"
    ids1 = torch.arange(8)
    ids2 = torch.tensor([0, 1, 2, 3, 9, 9, 9, 9])
    data = dict(sparse_1=list(s1.to(dev)), sparse_2=list(s2.to(dev)), dense_1=list(s1.to(dev)), dense_2=list(s2.to(dev)),
                label_1=[torch.zeros(1, dtype=torch.long, device=dev)] * 8,
                label_2=[torch.zeros(1, dtype=torch.long, device=dev)] * 8,
                id_1=[i.view(1).to(dev) for i in ids1], id_2=[i.view(1).to(dev) for i in ids2])

    def fresh():
        m, _ = build_pt([128, 64, 32])
        m.train()
        return m, train.Trainer(m, max_iters=8, lr=1e-3, grad_clip=1.0)

    def run(tr, n):
        out = []
        for _ in range(n):
            o = tr.step(data)
            out.append((float(o["loss"].detach()), float(o["grad_norm"])))
        return out
    _, tr = fresh()
    whole = run(tr, 4)
    _, tr1 = fresh()
    first = run(tr1, 2)
    path = str(tmp_path / "ck.pth")
    tr1.save(path)
    m2, tr2 = fresh()
    with torch.no_grad():
        for p in m2.parameters():
            p.add_(0.1)                       # the load must really overwrite
    tr2.load(path)
    assert tr2.iter == 2
    second = run(tr2, 2)
    assert first == whole[:2] and second == whole[2:], (whole, first, second)


def _hx(m, a, b):
    xyz1, xyz2, h1, h2 = m.siamese_forward(a, b)
    return h1, h2, xyz1, xyz2


def test_training_steps_agree_between_the_hip_and_torch_optimizers(grad_floor):
    """four Trainer iterations on the same HIP forward / backward graph, once with the HIP norm + clip + AdamW launches
    and once with clip_grad_norm_ + torch.optim.AdamW: losses and parameters must agree.  (Guards the caches keyed by
    Tensor._version -- padded biases, inference launch plans -- against an update that does not bump it: with stale
    biases the two runs part after the first step.)  Then eval mode must see the trained weights."""
    from pcr_amd import train
    s1, s2 = T.synthetic_pairs(8, 128, seed=2, kind="randn")
    dev = "cuda"
    ids1 = torch.arange(8)
    ids2 = torch.tensor([0, 1, 2, 3, 9, 9, 9, 9])
    data = dict(sparse_1=list(s1.to(dev)), sparse_2=list(s2.to(dev)), dense_1=list(s1.to(dev)), dense_2=list(s2.to(dev)),
                label_1=[torch.zeros(1, dtype=torch.long, device=dev)] * 8,
                label_2=[torch.zeros(1, dtype=torch.long, device=dev)] * 8,
                id_1=[i.view(1).to(dev) for i in ids1], id_2=[i.view(1).to(dev) for i in ids2])
    runs = []
    for fused in (True, False):
        m, _ = build_pt([128, 64, 32])
        m.train()
        tr = train.Trainer(m, max_iters=8, lr=1e-3, grad_clip=1.0, fused=fused)
        assert tr.fused == fused
        losses = [float(tr.step(data)["loss"].detach()) for _ in range(4)]
        runs.append((m, losses))
    (ma, la), (mb, lb) = runs
    # This 8-pair trajectory at lr 1e-3 is touchy: scaling every weight by 1 + 1e-7 before the first step moves the
    # step-3 / step-4 losses by 7e-4 / 4e-3 and the evaluation logits afterwards by 2.6e-2 (f32 arithmetic; the same with
    # split-bf16 gradient products: tools/train_sensitivity.py).  The two optimizers see bit-identical
    # gradients at step 1 (tests/test_gpu_train_ops.py) and round their updates differently by ~1e-7: with f32 gradient
    # products the runs happen to stay within 2e-4 / 2e-3, with split bf16 (whose rounding re-draws itself when a weight
    # moves by an ulp) they spread to the trajectory's own sensitivity, 1.6e-3 / 2.1e-2.  What this test guards -- a cache
    # keyed by Tensor._version that misses an update -- parts the losses by per cent at step TWO: that is held to 1e-5
    # in both arithmetics.
    strict = grad_floor == 1e-5
    assert la[:2] == pytest.approx(lb[:2], rel=1e-5, abs=2e-6), (la, lb)
    assert la == pytest.approx(lb, rel=2e-4 if strict else 1e-2, abs=2e-5), (la, lb)
    # (parameters are not compared one by one: conv biases in front of a BatchNorm have a zero true gradient, and AdamW
    # turns their rounding noise into +-lr steps -- the FUNCTION is what must agree: losses above, eval logits below)
    # eval after training: the launch plans are rebuilt from the updated weights
    ma.eval(); mb.eval()
    with torch.no_grad():
        ea = ma.match_forward_inference(*_hx(ma, s1.to(dev), s2.to(dev)))
        eb = mb.match_forward_inference(*_hx(mb, s1.to(dev), s2.to(dev)))
    assert float((ea - eb).abs().max()) < (2e-3 if strict else 6e-2) * max(1.0, float(eb.abs().max()))


def test_submodules_refuse_training_mode_on_the_fused_path():
    from pcr_amd._lib import PcrError
    m, _ = build_pt([128, 64, 32])
    m.train()
    x = torch.randn(2, 64, 128).cuda()
    xyz = torch.randn(2, 128, 3).cuda()
    with pytest.raises(PcrError):
        m.cross_stage1(x, xyz, x, xyz)          # stand-alone fused modules are eval-only


@pytest.mark.parametrize("mode,D,K,S,N,widths", [(0, 0, 32, 100, 128, (32, 32, 32)), (0, 32, 48, 70, 150, (64, 64, 64)),
                                                 (0, 64, 48, 33, 96, (128, 128, 128)), (1, 16, 64, 40, 300, (128, 128, 256)),
                                                 (1, 5, 20, 17, 64, (24, 40, 72)), (0, 8, 16, 50, 50, (32, 64, 128)),
                                                 # the wave-autonomous forms: two centres per block (K = 16), two / three
                                                 # blocks per item (K = 64 / 96), partial last items, c3 = 2 c2, no features
                                                 (0, 16, 16, 37, 80, (32, 32, 32)), (0, 24, 64, 21, 128, (64, 64, 64)),
                                                 (0, 8, 96, 11, 200, (32, 32, 64)), (1, 12, 32, 45, 256, (64, 64, 128)),
                                                 (0, 0, 16, 63, 64, (128, 128, 128)), (1, 0, 48, 29, 100, (32, 32, 32))])
def test_sa_fast_and_generic_paths_agree_with_torch(mode, D, K, S, N, widths):
    """fused SA kernel (decomposed layer 1, in-place LDS) vs the generic kernel vs plain torch fp32"""
    import torch.nn as nn
    import torch.nn.functional as F
    from pcr_amd import engine
    g = torch.Generator().manual_seed(5)
    cin = 3 + (2 * D if mode == 0 else D)
    convs, bns = [], []
    last = cin
    for w in widths:
        c = nn.Conv2d(last, w, 1)
        b = nn.BatchNorm2d(w)
        with torch.no_grad():
            b.running_mean.copy_(torch.randn(w, generator=g) * 0.1)
            b.running_var.copy_(torch.rand(w, generator=g) + 0.5)
            b.weight.copy_(1 + 0.1 * torch.randn(w, generator=g))
            b.bias.copy_(0.1 * torch.randn(w, generator=g))
        convs.append(c)
        bns.append(b.eval())
        last = w
    xyz = T.synthetic_clouds(3, N, seed=8, kind="box")
    feat = torch.randn(3, D, N, generator=g) if D else None
    idx = torch.randint(0, N, (3, S, K), generator=g, dtype=torch.int32)
    cidx = torch.randint(0, N, (3, S), generator=g, dtype=torch.int32) if mode == 1 else None
    # torch reference
    with torch.no_grad():
        ci = cidx.long() if cidx is not None else torch.arange(S).expand(3, S)
        gather = lambda t, ix: torch.gather(t, 1, ix.reshape(3, -1, 1).expand(-1, -1, t.shape[-1])).view(*ix.shape, t.shape[-1])
        centre_xyz = gather(xyz, ci)
        rel = gather(xyz, idx.long()) - centre_xyz.unsqueeze(2)
        rows = rel
        if D:
            pts = feat.permute(0, 2, 1)
            nb = gather(pts, idx.long())
            if mode == 0:
                cf = gather(pts, ci).unsqueeze(2)
                rows = torch.cat([rel, cf.expand(-1, -1, K, -1), nb - cf], -1)
            else:
                rows = torch.cat([rel, nb], -1)
        x = rows.permute(0, 3, 1, 2)
        for c, b in zip(convs, bns):
            x = F.relu(b(c(x)))
        want = x.max(dim=3)[0].numpy()
    outs = {}
    for fast in (True, False):
        plan = engine.SaPlan(convs, bns, "cuda", mode, fast=fast)
        if not fast:
            assert plan.fast is False
        out = plan.run(xyz.cuda(), None if feat is None else feat.cuda(), idx.cuda(),
                       None if cidx is None else cidx.cuda())
        outs[fast] = out.cpu().numpy()
        assert np.abs(outs[fast] - want).max() < 2e-5, (fast, np.abs(outs[fast] - want).max())


def test_backbone_shares_the_neighbour_search_of_its_first_two_levels():
    """the first Point-Transformer level keeps every point, so the second level queries the same cloud: one
    pcr_knn_prefix2_f32 launch ranks both levels' neighbours (round 5) -- same index tensors, hence bit-identical
    features; the profile shows one knn_prefix2 + one knn_prefix instead of three knn_prefix launches"""
    from pcr_amd import engine
    from mmdet3d.models import backbone_net as BN
    m, _ = build_pt([128, 64, 32])
    clouds = T.synthetic_clouds(5, 128, seed=3, kind="randn").cuda()
    with torch.no_grad():
        engine.PROFILE = []
        try:
            xyz_a, h_a = m.forward_inference(clouds)
            names = [r[0].split("[")[0] for r in engine.PROFILE]
        finally:
            engine.PROFILE = None
        prev, BN._NO_KNN2 = BN._NO_KNN2, True
        try:
            xyz_b, h_b = m.forward_inference(clouds)
        finally:
            BN._NO_KNN2 = prev
    assert names.count("knn_prefix2") == 1 and names.count("knn_prefix") == 1, names
    assert torch.equal(xyz_a, xyz_b) and torch.equal(h_a, h_b)


def test_match_gallery_equals_pairwise_matching():
    """encode once, score arbitrary (i, j) combinations"""
    import model_oracle as MO
    m, sd = build_pt([128, 64, 32])
    clouds = T.synthetic_clouds(6, 128, seed=11, kind="box")
    pairs = torch.tensor([[0, 1], [1, 0], [2, 5], [3, 3], [4, 0], [5, 2], [0, 4]])
    with torch.no_grad():
        xyz, h = m.forward_inference(clouds.cuda())
        got = m.match_gallery(h, xyz, pairs).cpu()
        direct = m.match_forward_inference(h[pairs[:, 0]].contiguous(), h[pairs[:, 1]].contiguous(),
                                           xyz[pairs[:, 0]].contiguous(), xyz[pairs[:, 1]].contiguous()).cpu()
        _, href = MO.pt_backbone(MO._sub(sd, "backbone."), clouds, [128, 64, 32])
        want = MO.match(sd, href[pairs[:, 0]], clouds[pairs[:, 0]], href[pairs[:, 1]], clouds[pairs[:, 1]])
    assert float((got - direct).abs().max()) < 1e-5
    assert float((got - want).abs().max()) < TOL
    # more combinations than one pass takes (the launches index clouds by a 16-bit grid dimension): chunked, same values
    many = torch.stack([torch.arange(70000) % 6, (torch.arange(70000) * 7 + 1) % 6], dim=1)
    with torch.no_grad():
        big = m.match_gallery(h, xyz, many).cpu()
        ref = m.match_gallery(h, xyz, many[:36]).cpu()
    assert m.match_gallery(h, xyz, many[:0]).shape == (0,)
    assert big.shape == (70000,) and torch.equal(big[:36], ref)
    assert torch.equal(big[66006:66042], big[:36])        # (the pattern of combinations repeats every 6 rows: second pass)


@pytest.mark.parametrize("match_type,combine", [("xcorr_eff", "add"), ("xcorr_eff", "minus"), ("xcorr-baseline", "cat"),
                                                ("concat", "cat")])
def test_match_variants_generic_path(match_type, combine):
    """the other match_type / combine variants of ReIDNet.match_forward (ReIDNet.py:387-462) run as a
    composition of the same launches + row kernels; checked against the torch restatement"""
    import model_oracle as MO
    import torch.nn.functional as F
    from mmdet3d.models import build_model
    cfg = copy.deepcopy(PT_MODEL)
    cfg.update(match_type=match_type, combine=combine)
    width = 64 * 2 if (match_type, combine) != ("concat", "cat") else 64 * 4
    cfg["match_head"] = [dict(type="LinearRes", n_in=width, n_out=width, norm="GN", ng=8),
                         dict(type="Linear", in_features=width, out_features=1)]
    m = build_model(cfg)
    sd = T.seeded_state_dict(T.manifest_of(m), 0)
    m.load_state_dict(sd)
    m = m.cuda().eval()
    s1, s2 = T.synthetic_pairs(3, 128, seed=4, kind="box")
    with torch.no_grad():
        xyz1, xyz2, h1, h2 = m.siamese_forward(s1.cuda(), s2.cuda())
        if match_type == "concat":
            # forward_test / forward_train pool through get_pooled_feats (reference :415-419); the reference's
            # match_forward_inference pools 'concat' inputs with self.maxpool whatever pool_type says (:455-457)
            got = m.match_forward(h1, h2, xyz1, xyz2, torch.zeros(3, device="cuda"), None, "cuda")[0].cpu()
        else:
            got = m.match_forward_inference(h1, h2, xyz1, xyz2).cpu()
        xr, hr = MO.pt_backbone(MO._sub(sd, "backbone."), torch.cat([s1, s2]), [128, 64, 32])
        a1, a2, x1, x2 = hr[:3], hr[3:], xr[:3], xr[3:]
        c1, c2 = MO._sub(sd, "cross_stage1."), MO._sub(sd, "cross_stage2.")
        if match_type == "xcorr_eff":
            p1, p2 = MO.cross_attention(c1, a1, x1, a2, x2), MO.cross_attention(c1, a2, x2, a1, x1)
            o1, o2 = MO.cross_attention(c2, p1, x1, p2, x2), MO.cross_attention(c2, p2, x2, p1, x1)
            pooled = MO.pool_both(o1 + o2 if combine == "add" else o1 - o2)
        elif match_type == "xcorr-baseline":
            pooled = MO.pool_both(MO.cross_attention(c2, MO.cross_attention(c1, a1, x1, a2, x2), x1, a2, x2))
        else:
            pooled = torch.cat([MO.pool_both(a1), MO.pool_both(a2)], dim=1)
        x = MO.linear_res(MO._linres_params(sd, "match_head.0.", 8), pooled)
        want = F.linear(x, sd["match_head.1.weight"], sd["match_head.1.bias"]).squeeze(1)
    assert float((got - want).abs().max()) < TOL


@pytest.mark.parametrize("sampling,use_knn", [("FPS", True), ("FPS", False), ("RANDOM", False)])
def test_sa_layer_dormant_sampling_and_grouping_branches(sampling, use_knn):
    """PointNetSetAbstractionEdgeSA with FPS centres and / or ball-query groups (the branches sample_and_group_edge
    keeps behind hard-coded flags, pointnet2_utils.py:262-272): HIP twins + the fused SA launch with explicit centre
    indices against the oracle fed with the same indices"""
    import model_oracle as MO
    from mmdet3d.models import pointnet2_utils as U
    sa = U.PointNetSetAbstractionEdgeSA(npoint=None, radius=0.9, nsample=16, mlp=[32, 32, 32, 64], sampling=sampling,
                                        use_xyz=True, use_knn=use_knn)
    # edge features: 3 + 2 D input channels for D = 16 point features
    sa.mlp_convs[0] = torch.nn.Conv2d(3 + 2 * 16, 32, 1)
    sd = T.seeded_state_dict(T.manifest_of(sa), 7)
    sa.load_state_dict(sd)
    sa = sa.cuda().eval()
    xyz = T.synthetic_clouds(3, 200, seed=31, kind="randn")
    feats = torch.randn(3, 16, 200, generator=torch.Generator().manual_seed(1))
    S = 50
    torch.manual_seed(5)
    new_xyz, out = sa(xyz.cuda(), feats.cuda(), S)
    # recover the centres the module drew (FPS start is random) from new_xyz, then replay the oracle with them
    d = (new_xyz.cpu().unsqueeze(2) - xyz.unsqueeze(1)).abs().sum(-1)
    centre = d.argmin(dim=-1)
    assert float(d.min(dim=-1)[0].max()) == 0.0
    if sampling == "FPS":
        assert torch.equal(centre, MO.farthest_point_sample_py(xyz, S, centre[:, 0]))
    else:
        assert torch.equal(centre, torch.arange(S).repeat(3, 1))
    cx = torch.gather(xyz, 1, centre.unsqueeze(-1).expand(-1, -1, 3))
    if use_knn:
        gidx = torch.argsort(MO.square_distance(cx, xyz), dim=-1)[:, :, :16]
    else:
        gidx = MO.query_ball_point_py(0.9, 16, xyz, cx)
    with torch.no_grad():
        _, want = MO.sa_edge_layer(sd, xyz, feats, S, 16, centre_idx=centre, group_idx=gidx)
    assert float((out.cpu() - want).abs().max()) < TOL


def test_trainer_graph_mode_equals_eager_bit_for_bit():
    """Trainer(graph=True): forward + backward replayed from a HIP graph after three eager iterations -- same launches in the
    same order, so losses, gradient norms, parameters, BatchNorm statistics and log values equal the eager run's bit for
    bit, also when every iteration brings NEW batch tensors; evaluation afterwards sees the trained weights"""
    import bench
    from pcr_amd import train

    def batch(seed, pairs=8, n=128):
        s1, s2 = T.synthetic_pairs(pairs, n, seed=seed, kind="randn")
        ids1 = torch.arange(pairs)
        ids2 = torch.where(torch.arange(pairs) % 2 == 0, ids1, ids1 + 100)
        zero = torch.zeros(1, dtype=torch.long, device="cuda")
        return dict(sparse_1=list(s1.cuda()), sparse_2=list(s2.cuda()), dense_1=list(s1.cuda()), dense_2=list(s2.cuda()),
                    label_1=[zero] * pairs, label_2=[zero] * pairs,
                    id_1=[i.view(1).cuda() for i in ids1], id_2=[i.view(1).cuda() for i in ids2])

    runs = {}
    for mode in (False, True):
        m, _ = bench.build_pt_model([128, 64, 32])
        m.train()
        tr = train.Trainer(m, max_iters=12, lr=1e-3, grad_clip=1.0, graph=mode)
        rec, held = [], []
        for it in range(7):
            out = tr.step(batch(10 + it))
            held.append((out["loss"], out["log_vars"]))             # read only AFTER the loop (loader.run_epochs does that)
            rec.append((float(out["loss"]), float(out["grad_norm"]), out["log_vars"]["match_acc"], out["log_vars"]["loss"]))
        assert tr.graph == mode                                    # (the capture did not fall back)
        # an iteration's outputs stay that iteration's: a replay must not overwrite what an earlier replay handed out
        assert [float(l) for l, _ in held] == [r[0] for r in rec]
        assert [lv["loss"] for _, lv in held] == [r[3] for r in rec]
        assert len({l.data_ptr() for l, _ in held}) == len(held)
        m.eval()
        s1, s2 = T.synthetic_pairs(4, 128, seed=3)
        with torch.no_grad():
            logits = bench.hot_path(m, s1.cuda(), s2.cuda()).cpu()
        runs[mode] = (rec, {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}, logits)
    assert runs[False][0] == runs[True][0], (runs[False][0], runs[True][0])
    for k, v in runs[False][1].items():
        assert torch.equal(v, runs[True][1][k]), k
    assert torch.equal(runs[False][2], runs[True][2])


def test_eval_flip_equals_the_swapped_batch():
    """eval_flip (ReIDNet.py:144-147): the training path on a batch with eval_flip=True computes, bit for bit, what it
    computes on the batch with its two sides swapped"""
    import copy
    import bench
    from mmdet3d.models import build_model
    s1, s2 = T.synthetic_pairs(6, 128, seed=21, kind="randn")
    ids1 = torch.arange(6)
    ids2 = torch.where(torch.arange(6) % 2 == 0, ids1, ids1 + 100)
    zero = torch.zeros(1, dtype=torch.long, device="cuda")

    def batch(a, b, ia, ib):
        return dict(sparse_1=list(a.cuda()), sparse_2=list(b.cuda()), dense_1=list(a.cuda()), dense_2=list(b.cuda()),
                    label_1=[zero] * 6, label_2=[zero] * 6, id_1=[i.view(1).cuda() for i in ia], id_2=[i.view(1).cuda() for i in ib])
    man = T.load_manifest(os.path.join(bench.ROOT, "tests", "golden", "pt_manifest.json"))
    sd = T.seeded_state_dict(man, 0)
    loss = {}
    for flip in (False, True):
        cfg = copy.deepcopy(bench.PT_MODEL)
        cfg["eval_flip"] = flip
        m = build_model(cfg)
        m.load_state_dict(sd, strict=True)
        m = m.cuda().train()
        data = batch(s1, s2, ids1, ids2) if flip else batch(s2, s1, ids2, ids1)
        out = m.train_step(data, None)
        loss[flip] = (float(out["loss"]), out["log_vars"]["match_acc"])
    assert loss[False] == loss[True]
