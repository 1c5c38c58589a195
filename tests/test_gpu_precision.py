"""The arithmetic modes of the MFMA-bound layers (pcr_amd/engine.py PRECISION, include/pcr.h PCR_PREC_*) against the
oracles: "f32" (f32-input MFMA, exact fmaf chains), "bf16x3" (split bf16, the default) inside the 1e-4 parity bound
of the north star, and "bf16" (BASELINE config 2 as stated: bf16 activations / weights, f32 accumulate) inside the
tolerance SURVEY Appendix B measured for bf16 autocast (1e-2 on the logits)."""
import json

import pytest
import torch

from pcr_amd import testing as T

pytestmark = pytest.mark.gpu


def _logits(model, s1, s2):
    import bench
    with torch.no_grad():
        return bench.hot_path(model, s1.cuda(), s2.cuda()).cpu()


@pytest.mark.parametrize("kind,n,clouds", [("pt", 128, "randn"), ("pt", 1024, "box"), ("ssg", 1024, "box"),
                                           ("ssg", 1024, "crop")])
def test_precisions_against_the_oracle(kind, n, clouds):
    import bench
    import model_oracle as MO
    from pcr_amd import engine
    bl = {128: [128, 64, 32], 1024: [1024, 512, 256]}[n] if kind == "pt" else None
    model, sd = bench.build_model(kind, bl)
    s1, s2 = T.synthetic_pairs(3, n, seed=11, kind=clouds)
    with torch.no_grad():
        want = MO.pt_pairs(sd, s1, s2, bl) if kind == "pt" else MO.ssg_pairs(sd, s1, s2)
    err = {}
    for prec in ("f32", "bf16x3", "bf16"):
        with engine.precision(prec):
            err[prec] = float((_logits(model, s1, s2) - want).abs().max())
    print(json.dumps(err))
    assert err["f32"] < 1e-4 and err["bf16x3"] < 1e-4, err
    assert err["bf16"] < 1e-2, err
    # the three modes really are different kernels: plain bf16 is measurably coarser than the split form
    assert err["bf16"] > 4 * err["bf16x3"], err


@pytest.mark.parametrize("shape", [(0, 32, 32, 32, 32, 1), (32, 64, 64, 64, 48, 0), (64, 128, 128, 128, 48, 0),
                                   (0, 64, 64, 128, 32, 1), (128, 128, 128, 256, 64, 1), (128, 256, 256, 256, 16, 0),
                                   (64, 512, 512, 512, 16, 1)])
def test_sa_layer_in_every_precision_against_torch(shape):
    """one grouped SA layer (gather + 3 x conv/BN/ReLU + max over K) per precision against plain torch fp32 on the same
    kNN groups; (D, c1, c2, c3, K, mode); the 512-wide layer has no bf16 instantiation and must come back as f32"""
    import torch.nn as nn
    from pcr_amd import engine
    D, c1, c2, c3, K, mode = shape
    B, N, S = 3, 256, 96
    g = torch.Generator().manual_seed(5)
    xyz = torch.randn(B, N, 3, generator=g)
    feat = torch.randn(B, D, N, generator=g) if D else None
    cin = 3 + (2 * D if mode == 0 else D)
    convs = [nn.Conv2d(a, b, 1) for a, b in ((cin, c1), (c1, c2), (c2, c3))]
    bns = [nn.BatchNorm2d(c) for c in (c1, c2, c3)]
    for i, bn in enumerate(bns):
        bn.running_mean.copy_(torch.randn(bn.num_features, generator=g) * 0.1)
        bn.running_var.copy_(torch.rand(bn.num_features, generator=g) + 0.5)
        bn.weight.data.copy_(torch.rand(bn.num_features, generator=g) + 0.5)
        bn.bias.data.copy_(torch.randn(bn.num_features, generator=g) * 0.1)
        bn.eval()
    idx = engine.knn_prefix(xyz.cuda(), S, K)
    il = idx.long().cpu()
    with torch.no_grad():
        centre = xyz[:, :S]
        nb = torch.gather(xyz, 1, il.reshape(B, S * K, 1).expand(-1, -1, 3)).view(B, S, K, 3)
        rows = [nb - centre.unsqueeze(2)]
        if D:
            pts = feat.permute(0, 2, 1)
            fn = torch.gather(pts, 1, il.reshape(B, S * K, 1).expand(-1, -1, D)).view(B, S, K, D)
            if mode == 0:
                fc = pts[:, :S].unsqueeze(2).expand(-1, -1, K, -1)
                rows += [fc, fn - fc]
            else:
                rows += [fn]
        x = torch.cat(rows, dim=-1).permute(0, 3, 1, 2)
        for c, b in zip(convs, bns):
            x = torch.relu(b(c(x)))
        want = x.max(dim=3)[0]
    plan = engine.SaPlan(convs, bns, torch.device("cuda"), mode)
    out = {}
    for prec in ("f32", "bf16x3", "bf16"):
        with engine.precision(prec), torch.no_grad():
            out[prec] = plan.run(xyz.cuda(), None if feat is None else feat.cuda(), idx).cpu()
    scale = float(want.abs().max())
    err = {k: float((v - want).abs().max()) / scale for k, v in out.items()}
    print(json.dumps(dict(shape=shape, scale=scale, **err)))
    assert err["f32"] < 2e-6 and err["bf16x3"] < 2e-5, err
    if c1 == 512:       # no bf16 SA kernel for this width: layers 2 / 3 ran in f32 (only the layer-1 tables follow the mode)
        assert err["bf16x3"] < 5e-6 and err["bf16"] < 3e-3
    else:
        assert 1e-5 < err["bf16"] < 3e-2, err
        assert not torch.equal(out["bf16x3"], out["f32"])


@pytest.mark.parametrize("prec", ["f32", "bf16x3", "bf16"])
def test_ragged_and_k_row_evaluation_agree_bit_for_bit_in_every_precision(prec):
    """the duplicate-free (ragged) SA evaluation must equal the K-row evaluation bit for bit whatever the arithmetic:
    both forms run the same dense tile on the same rows"""
    import bench
    from pcr_amd import engine
    model, _ = bench.build_model("ssg", None)
    s1, s2 = T.synthetic_pairs(4, 1024, seed=9, kind="dup")
    with engine.precision(prec):
        a = _logits(model, s1, s2)
        for sa in model.backbone.SA_modules:
            sa.skip_repeats = False
        b = _logits(model, s1, s2)
    assert torch.equal(a, b)


def test_attention_blocks_in_split_bf16_against_the_f32_path():
    """the dense phases of the attention apply kernel (Q, message, feed-forward, cov_final) in "bf16x3" mode against
    the f32 kernels on the same inputs: a different kernel (bits differ) within 2e-5 of the output's scale; a key-side
    state built in one arithmetic cannot be applied in the other (the per-cloud matrix is stored in its layout)"""
    import bench
    from pcr_amd import _lib as L
    from pcr_amd import engine
    model, _ = bench.build_pt_model([128, 64, 32])
    g = torch.Generator().manual_seed(3)
    for d, n in ((64, 128), (64, 100)):
        h1 = torch.randn(5, d, n, generator=g).cuda()
        h2 = torch.randn(5, d, n, generator=g).cuda()
        x1 = torch.randn(5, n, 3, generator=g).cuda()
        x2 = torch.randn(5, n, 3, generator=g).cuda()
        out = {}
        for prec in ("f32", "bf16x3"):
            with engine.precision(prec), torch.no_grad():
                out[prec] = model.cross_stage1(h1, x1, h2, x2).cpu()
        scale = float(out["f32"].abs().max())
        err = float((out["bf16x3"] - out["f32"]).abs().max()) / scale
        print(json.dumps(dict(d=d, n=n, err=err)))
        assert 0 < err < 2e-5, err
    for i, sa in enumerate(model.backbone.SA_modules):       # d_model 32 / 64 / 128 self-attention
        d = (32, 64, 128)[i]
        f = torch.randn(3, d, 96, generator=g).cuda()
        x = torch.randn(3, 96, 3, generator=g).cuda()
        out = {}
        for prec in ("f32", "bf16x3"):
            with engine.precision(prec), torch.no_grad():
                out[prec] = sa.self_attention(f, x).cpu()
        err = float((out["bf16x3"] - out["f32"]).abs().max()) / float(out["f32"].abs().max())
        print(json.dumps(dict(d=d, err=err)))
        assert 0 < err < 2e-5, err
    plan = model.cross_stage1.plan(torch.device("cuda"))
    with engine.precision("f32"):
        kv = plan.kv(h2, x2)
    with engine.precision("bf16x3"), pytest.raises(L.PcrError):
        plan.apply(h1, None, kv, h2.shape[2])


# ---- how much of the 1e-4 parity bound split bf16 spends, beyond the bench's two input distributions ------------------
# (the margin was only known on randn clouds and the 4 x 2 x 1.5 m box.)  The error of a split-bf16
# product scales with the operands' magnitudes, so the sweep moves everything that moves them: the input extent (x 0.1,
# x 1, x 10 of the box -- LiDAR crops of large vehicles reach ~10 m), the weights (three seeds), and BatchNorm statistics
# far from the seeded ones (running variance x 0.25 / x 4, means x 3, which rescales every folded layer).  The yardstick is
# the f32-input MFMA path on the same inputs (exact fmaf chains; pinned to the oracle and the reference goldens by the
# tests above), so the sweep can run at sizes the CPU oracle cannot.
def _rescale_bn(sd, var_scale, mean_scale):
    out = {}
    for k, v in sd.items():
        if k.endswith("running_var"):
            out[k] = v * var_scale
        elif k.endswith("running_mean"):
            out[k] = v * mean_scale
        else:
            out[k] = v
    return out


BN_CASES = (("seeded", (1.0, 1.0)), ("var/4,mean*3", (0.25, 3.0)), ("var*4", (4.0, 1.0)), ("var/16,mean*5", (1.0 / 16.0, 5.0)))


@pytest.mark.parametrize("kind,n", [("ssg", 1024), ("pt", 1024), ("pt", 128), ("pointnet", 256), ("dgcnn", 256)])
def test_split_bf16_margin_over_input_scale_weights_and_bn_statistics(kind, n):
    """With the split-bf16 guard (pcr_amd/engine.py, ReIDNet.calibrate_precision; round 5): every case of the sweep is
    CALIBRATED on one batch and MEASURED on another batch of the same distribution.  On the calibration batch the logits
    are within engine.GUARD_BOUND = 5e-5 of the f32 path by construction; the fresh batch must stay inside the 1e-4
    parity bound, and the worst of the whole sweep is printed (round 4, no guard: 9.7e-5 for the SSG model under shifted
    statistics; the fourth statistics case -- variance / 16, means x 5 -- put split bf16 at 6.7e-3 there)."""
    import bench
    from pcr_amd import engine
    bl = {128: [128, 64, 32], 1024: [1024, 512, 256]}.get(n) if kind == "pt" else None
    worst, levels = {}, {}
    for seed in (0, 1, 2):
        model, _ = bench.build_model(kind, bl)
        man = T.manifest_of(model)
        base = T.seeded_state_dict(man, seed)
        for bn_name, (vs, ms) in BN_CASES:
            if seed and bn_name != "seeded":
                continue                                  # (BN variants on the first seed only)
            model.load_state_dict(_rescale_bn(base, vs, ms), strict=True)
            model = model.cuda().eval()
            for scale in (0.1, 1.0, 10.0):
                if scale != 1.0 and bn_name != "seeded":
                    continue
                dist = "box" if kind in ("ssg", "pt") else "randn"
                c1, c2 = T.synthetic_pairs(6, n, seed=40 + seed, kind=dist)          # calibration batch
                s1, s2 = T.synthetic_pairs(6, n, seed=140 + seed, kind=dist)         # measured batch
                with engine.precision("bf16x3"):
                    st = model.calibrate_precision((c1 * scale).cuda(), (c2 * scale).cuda())
                    assert st["dlogit"][st["level"]] <= engine.GUARD_BOUND, st
                    got = _logits(model, s1 * scale, s2 * scale)
                with engine.precision("f32"):
                    ref = _logits(model, s1 * scale, s2 * scale)
                assert torch.isfinite(ref).all() and torch.isfinite(got).all()
                worst[(seed, bn_name, scale)] = float((got - ref).abs().max())
                levels[(seed, bn_name, scale)] = st["level"]
    print(json.dumps({"%d|%s|x%g" % k: (levels[k], v) for k, v in worst.items()}))
    bad = {k: v for k, v in worst.items() if not v <= 5e-5}     # (worst of the sweep <= 5e-5, guard on)
    assert not bad, bad
    # the seeded checkpoints (every bench line) keep every launch in split bf16
    assert all(lv == 0 for k, lv in levels.items() if k[1] == "seeded" and k[2] == 1.0) or kind == "pt", levels


@pytest.mark.parametrize("kind,n,pairs", [("ssg", 1024, 4), ("pt", 128, 6), ("pt", 256, 6), ("pointnet", 256, 6)])
@pytest.mark.parametrize("bn_name,vs,ms", [("var/4,mean*3", 0.25, 3.0), ("var*4", 4.0, 1.0)])
def test_guarded_split_bf16_against_the_oracle_under_shifted_statistics(kind, n, pairs, bn_name, vs, ms):
    """the same guard held to the REFERENCE restatement (oracle/model_oracle.py, pinned to the reference's goldens), not to
    the f32 HIP path: shifted BatchNorm statistics, calibrated on one batch, measured on another against the CPU oracle
    at sizes it finishes in seconds -- inside the north star's 1e-4"""
    import bench
    import model_oracle as MO
    from pcr_amd import engine
    bl = {128: [128, 64, 32], 256: [256, 128, 64]}.get(n) if kind == "pt" else None
    model, _ = bench.build_model(kind, bl)
    sd = _rescale_bn(T.seeded_state_dict(T.manifest_of(model), 0), vs, ms)
    model.load_state_dict(sd, strict=True)
    model = model.cuda().eval()
    dist = "box" if kind in ("ssg", "pt") else "randn"
    c1, c2 = T.synthetic_pairs(pairs, n, seed=41, kind=dist)
    s1, s2 = T.synthetic_pairs(pairs, n, seed=141, kind=dist)
    with engine.precision("bf16x3"):
        st = model.calibrate_precision(c1.cuda(), c2.cuda())
        got = _logits(model, s1, s2)
    with torch.no_grad():
        want = {"ssg": lambda: MO.ssg_pairs(sd, s1, s2), "pt": lambda: MO.pt_pairs(sd, s1, s2, bl),
                "pointnet": lambda: MO.pointnet_pairs(sd, s1, s2)}[kind]()
    err = float((got - want).abs().max())
    print(json.dumps(dict(kind=kind, n=n, bn=bn_name, level=st["level"], dlogit=st["dlogit"], err_vs_oracle=err)))
    assert err < 1e-4, (err, st)


def test_guard_levels_move_the_folded_batchnorm_launches_to_f32():
    """level 1 = the grouped SA launches in f32, everything else split bf16; level 2 = the f32 path bit for bit; an
    uncalibrated model, a model in another arithmetic and PCR_GUARD=0 stay at level 0"""
    import bench
    from pcr_amd import engine
    model, _ = bench.build_model("ssg", None)
    s1, s2 = T.synthetic_pairs(4, 1024, seed=9, kind="box")
    assert model.guard_state() is None and model.precision_level() == 0
    with engine.precision("f32"):
        ref = _logits(model, s1, s2)
    outs = {}
    for lv in (0, 1, 2):
        with engine.precision("bf16x3"), engine.guard_level(lv):
            outs[lv] = _logits(model, s1, s2)
    assert torch.equal(outs[2], ref) and not torch.equal(outs[0], ref) and not torch.equal(outs[1], outs[0])
    assert float((outs[1] - ref).abs().max()) < 1e-4
    with engine.precision("bf16x3"):
        st = model.calibrate_precision(s1.cuda(), s2.cuda(), bound=0.0)       # an impossible bound: ends at the f32 path
        assert st["level"] == 2 and model.precision_level() == 2
        assert torch.equal(_logits(model, s1, s2), ref)
        st = model.calibrate_precision(s1.cuda(), s2.cuda())
        assert st["level"] == 0 and st["dlogit"][0] <= engine.GUARD_BOUND and model.precision_level() == 0
        with torch.no_grad():
            next(model.parameters()).mul_(1.0)                                # new weight version: not calibrated any more
        assert model.guard_state() is None and model.precision_level() == 0


# ---- round 6: lazy calibration at every entry point + the run-time sentinel ---------
def _dev0(model, s1, s2):
    """max |logit(level 0) - logit(f32)| per pair, measured explicitly (no guard state touched)"""
    from pcr_amd import engine
    with torch.no_grad():
        with engine.precision("f32"):
            ref = model._hot(s1, s2)
        with engine.precision("bf16x3"), engine.guard_level(0):
            got = model._hot(s1, s2)
    return (got - ref).abs(), ref


def test_every_inference_entry_point_calibrates_on_its_first_batch():
    """forward_inference / siamese_forward / match_forward_inference / match_gallery only READ the level, so a
    tracker that never called calibrate_precision ran unguarded.  Now the first batch of a weight version calibrates,
    whichever entry point sees it; a replaced Parameter (not an in-place write) invalidates the state too."""
    import bench
    from pcr_amd import engine
    s1, s2 = [t.cuda() for t in T.synthetic_pairs(6, 128, seed=9, kind="randn")]
    with engine.precision("bf16x3"):
        for entry in ("siamese_forward", "forward_inference", "match_forward_inference", "match_gallery"):
            model, _ = bench.build_pt_model([128, 64, 32])
            assert model.guard_state() is None
            with torch.no_grad(), engine.guard_level(0):
                xyz1, xyz2, h1, h2 = model._siamese_forward(s1, s2)          # (features from "elsewhere": no entry point)
            assert model.guard_state() is None
            with torch.no_grad():
                if entry == "siamese_forward":
                    model.siamese_forward(s1, s2)
                elif entry == "forward_inference":
                    model.forward_inference(torch.cat([s1, s2]))
                elif entry == "match_forward_inference":
                    model.match_forward_inference(h1, h2, xyz1, xyz2)
                else:
                    pairs = torch.tensor([[0, 6], [1, 7]])
                    model.match_gallery(torch.cat([h1, h2]), torch.cat([xyz1, xyz2]), pairs)
            st = model.guard_state()
            assert st is not None and st["dlogit"][st["level"]] <= engine.GUARD_BOUND, (entry, st)
        # a REPLACED parameter (same shape, new object) is a new weight version (the cached tensor list)
        w = model.match_head[1].weight
        model.match_head[1].weight = torch.nn.Parameter(w.detach().clone())
        assert model.guard_state() is None
        model.load_state_dict({k: v.clone() for k, v in model.state_dict().items()}, assign=True)
        assert model.guard_state() is None
        with torch.no_grad():
            model.siamese_forward(s1, s2)
        assert model.guard_state() is not None
        # PointNet returns a transformed xyz: a match-only call on an uncalibrated model runs the matching in f32
        pn, _ = bench.build_model("pointnet", None)
        p1, p2 = [t.cuda() for t in T.synthetic_pairs(3, 256, seed=4, kind="randn")]
        with torch.no_grad(), engine.guard_level(0):
            xyz1, xyz2, h1, h2 = pn._siamese_forward(p1, p2)
        assert pn._match_level(xyz1, xyz2) == 2 and pn.guard_state() is None
    with engine.precision("f32"):
        model, _ = bench.build_pt_model([128, 64, 32])
        with torch.no_grad():
            model.siamese_forward(s1, s2)
        assert model.guard_state() is None          # (nothing to guard in the reference's arithmetic)


def test_sentinel_raises_the_level_before_a_hostile_batch_is_returned(monkeypatch):
    """Batch 1 is benign, batch k is not: the weights are the shifted-statistics checkpoint that measured 9.7e-5 unguarded
    (SSG, variance / 4, means x 3), the bound sits between the level-0 deviations of two live batches, the model is
    calibrated (lazily) on the benign one and keeps level 0; when the hostile batch arrives the sentinel re-checks it
    against the f32 path BEFORE computing it, raises the level of this weight version and the logits that come back are
    inside the bound -- at no point is a returned logit off by more than the bound."""
    import bench
    from pcr_amd import engine
    model, _ = bench.build_model("ssg", None)
    sd = _rescale_bn(T.seeded_state_dict(T.manifest_of(model), 0), 0.25, 3.0)
    model.load_state_dict(sd, strict=True)
    model = model.cuda().eval()
    cands = []
    for seed, scale in ((40, 0.1), (41, 0.5), (42, 1.0), (43, 2.0), (44, 10.0)):
        a, b = T.synthetic_pairs(6, 1024, seed=seed, kind="box")
        a, b = (a * scale).cuda(), (b * scale).cuda()
        d, _ = _dev0(model, a, b)
        cands.append((float(d.max()), a, b))
    cands.sort(key=lambda c: c[0])
    (d_lo, a1, a2), (d_hi, b1, b2) = cands[0], cands[-1]
    print(json.dumps(dict(level0_deviation=[c[0] for c in cands])))
    assert d_hi > 1.5 * d_lo > 0, (d_lo, d_hi)
    bound = (d_lo * d_hi) ** 0.5
    monkeypatch.setattr(engine, "GUARD_BOUND", bound)
    monkeypatch.setattr(engine, "GUARD_EVERY", 1)
    assert engine.GUARD_SENTINEL_PAIRS >= 6
    with engine.precision("bf16x3"), torch.no_grad():
        out = bench.hot_path(model, a1, a2)                          # batch 1: calibrates lazily, level 0
        st = model.guard_state()
        assert st["level"] == 0 and st["sentinel"]["checks"] == 0
        out = bench.hot_path(model, a1, a2)                          # batch 2: sentinel check, still benign
        st = model.guard_state()
        assert st["level"] == 0 and st["sentinel"]["checks"] == 1 and st["sentinel"]["worst"] <= bound
        with engine.precision("f32"):
            ref = model._hot(b1, b2)
        got = bench.hot_path(model, b1, b2)                          # batch k: hostile
        st = model.guard_state()
        assert st["level"] >= 1 and st["sentinel"]["raised"] and st["sentinel"]["raised"][0]["was"] == 0, st
        assert float((got - ref).abs().max()) <= bound, (float((got - ref).abs().max()), bound, st)
        # the level stays raised for this weight version (benign batches included) ...
        bench.hot_path(model, a1, a2)
        assert model.guard_state()["level"] == st["level"]
        # ... and the sentinel switched off (PCR_GUARD_EVERY=0) would have let the hostile batch through
        monkeypatch.setattr(engine, "GUARD_EVERY", 0)
        model.calibrate_precision(a1, a2)
        got0 = bench.hot_path(model, b1, b2)
        assert model.guard_state()["level"] == 0 and float((got0 - ref).abs().max()) > bound
    del out


def test_sentinel_is_silent_inside_a_graph_capture_and_cheap():
    """the bench's captured pass: nothing can be read back inside a capture, so the tick is a no-op there and the replay
    equals the eager pass bit for bit; the default period is 64 batches"""
    import bench
    from pcr_amd import engine
    assert engine.GUARD_EVERY == 64 and engine.GUARD_SENTINEL_PAIRS == 8
    model, _ = bench.build_pt_model([128, 64, 32])
    s1, s2 = [t.cuda() for t in T.synthetic_pairs(16, 128, seed=2, kind="randn")]
    with engine.precision("bf16x3"), torch.no_grad():
        ref = bench.hot_path(model, s1, s2)
        calls = model.guard_state()["calls"]
        step, mode = bench.graph_step(lambda: bench.hot_path(model, s1, s2))
        assert mode == "hipgraph", mode
        for _ in range(3):
            out = step()
        torch.cuda.synchronize()
        assert torch.equal(out, ref)
        assert model.guard_state()["calls"] > calls          # (the eager pilot calls ticked; the capture and replays did not)
