"""Randomised MODEL-level parity sweep on the GPU box (not part of the test suite): whole siamese forward passes with
random point counts / backbone lists / cloud kinds against the torch-eager oracle, tolerance 1e-4 on the logits.
python tools/fuzz_models.py [seconds] [seed]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "point-cloud-reid_amd"), os.path.join(ROOT, "oracle")]
import bench                       # noqa: E402
import model_oracle as MO          # noqa: E402
from pcr_amd import testing as T   # noqa: E402


def forced_ptx(model, sd, a, b):
    """stage-wise parity of the xcorr matching with the GPU's own stage inputs fed to the oracle (feature-space kNN
    is discontinuous, so END-TO-END agreement can break at a near-tie without any stage being wrong)"""
    from pcr_amd import engine
    # (round 6: every entry point applies the model's guard level -- an xcorr model calibrates to level 2, the f32 path --
    # so the stages called directly here must run at that level too, or the two chains differ by arithmetic and flip
    # neighbours between themselves)
    with torch.no_grad(), engine.guard_level(model.precision_level()):
        xyz1, xyz2, h1, h2 = model.siamese_forward(a.cuda(), b.cuda())
        ga = model.cross_stage1(h1, xyz1, h2, xyz2)
        gb = model.local_stage1(ga, xyz1)
        gc = model.cross_stage2(gb, xyz1, h2, xyz2)
        gd = model.local_stage2(gc, xyz1)
        got = model.match_forward_inference(h1, h2, xyz1, xyz2).cpu()
        c = lambda t: t.cpu()       # noqa: E731
        errs = [
            float((c(ga) - MO.cross_attention(MO._sub(sd, "cross_stage1."), c(h1), c(xyz1), c(h2), c(xyz2))).abs().max()),
            float((c(gb) - MO.local_self_attention(MO._sub(sd, "local_stage1."), c(ga), c(xyz1), 2, 48)).abs().max()),
            float((c(gc) - MO.cross_attention(MO._sub(sd, "cross_stage2."), c(gb), c(xyz1), c(h2), c(xyz2))).abs().max()),
            float((c(gd) - MO.local_self_attention(MO._sub(sd, "local_stage2."), c(gc), c(xyz1), 2, 48)).abs().max()),
        ]
        pooled = MO.pool_both(c(gd))
        x = MO.linear_res(MO._linres_params(sd, "match_head.0.", 8), pooled)
        head = torch.nn.functional.linear(x, sd["match_head.1.weight"], sd["match_head.1.bias"]).squeeze(1)
        errs.append(float((got - head).abs().max()))
    return max(errs)


def forced_dgcnn(model, sd, a, b):
    from pcr_amd import dgcnn_engine
    x = torch.cat([a, b], 0).permute(0, 2, 1).contiguous()
    st = {}
    with torch.no_grad():
        dgcnn_engine.forward(model.backbone, x.cuda(), st)
        p = MO._sub(sd, "backbone.")
        f, worst = x, 0.0
        for i in (1, 2, 3, 4):
            want = MO.dgcnn_edge_layer(p, f, i)
            got = st["x%d" % i].cpu()
            worst = max(worst, float((got - want).abs().max()) / max(1.0, float(want.abs().max())))
            f = got                                            # teacher forcing: the next layer sees the GPU's output
    return worst


def pt_near_tie(a, b, bl):
    """A Point-Transformer case off by >= 1e-4 in BOTH arithmetics: is it a near-tie of the K-th neighbour between the distance the
    kernels rank by (squared differences, pcr_sqdist3) and the expanded form -2 a.b + |a|^2 + |b|^2 of the reference's
    square_distance (models/pointnet2_utils.py:169-188), whose rounding at |x| ~ 1 is ~1e-7?  -> the differing (kept, dropped)
    pairs' relative distance gaps, or None if the neighbour sets agree / differ by more than a rounding."""
    from pcr_amd import engine
    gaps = []
    for cl in (a, b):
        pts = cl
        for S, K in zip(bl, (32, 48, 48)):
            idx = engine.knn_prefix(pts.cuda().contiguous(), S, K).cpu().long()
            src, dst = pts[:, :S], pts
            d_exp = -2 * torch.matmul(src, dst.transpose(1, 2)) + (src ** 2).sum(-1)[:, :, None] + (dst ** 2).sum(-1)[:, None, :]
            ref = d_exp.topk(K, dim=-1, largest=False)[1]
            d_dir = ((src[:, :, None, :] - dst[:, None, :, :]) ** 2).sum(-1)
            for bi in range(pts.shape[0]):
                for c in range(S):
                    g, r = set(idx[bi, c].tolist()), set(ref[bi, c].tolist())
                    if g != r:
                        dg = torch.tensor([float(d_dir[bi, c, i]) for i in sorted(g - r)])
                        dr = torch.tensor([float(d_dir[bi, c, i]) for i in sorted(r - g)])
                        gap = float((dr.max() - dg.min()).abs() / max(1e-12, float(dr.max())))
                        if gap > 2e-6:
                            return None
                        gaps.append(gap)
            pts = pts[:, :S]
    return gaps or None


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    ssg, sd_ssg = bench.build_model("ssg", None)
    pn, sd_pn = bench.build_model("pointnet", None)
    dg, sd_dg = bench.build_model("dgcnn", None)
    t0, n, worst, counts = time.time(), 0, {}, {}
    while time.time() - t0 < budget:
        case = ["pt", "ptx", "ssg", "pointnet", "dgcnn"][rng.integers(0, 5)]
        pairs = int(rng.integers(1, 5))
        seed = int(rng.integers(0, 1 << 30))
        if case in ("pt", "ptx"):
            n0 = int(rng.integers(48, 400))
            s0 = int(rng.integers(48, n0 + 1))
            s1 = int(rng.integers(48, s0 + 1))
            s2 = int(rng.integers(48, s1 + 1))           # nsample = 32/48/48 needs >= 48 points per level
            bl = [s0, s1, s2]
            kind = ["randn", "box", "dup"][rng.integers(0, 3)]
            model, sd = bench.build_model(case, bl)
            a, b = T.synthetic_pairs(pairs, n0, seed, kind)
            with torch.no_grad():
                want = MO.pt_pairs(sd, a, b, bl) if case == "pt" else MO.pt_pairs_xcorr(sd, a, b, bl)
        elif case == "ssg":
            n0 = int(rng.integers(512, 1025))
            kind = ["box", "dup"][rng.integers(0, 2)]
            model, sd = ssg, sd_ssg
            a, b = T.synthetic_pairs(pairs, n0, seed, kind)
            with torch.no_grad():
                want = MO.ssg_pairs(sd, a, b)
        elif case == "pointnet":
            n0 = int(rng.integers(33, 300))
            model, sd = pn, sd_pn
            a, b = T.synthetic_pairs(pairs, n0, seed, "randn")
            with torch.no_grad():
                want = MO.pointnet_pairs(sd, a, b)
        else:
            n0 = int(rng.integers(20, 300))
            model, sd = dg, sd_dg
            a, b = T.synthetic_pairs(pairs, n0, seed, "randn")
            with torch.no_grad():
                want = MO.dgcnn_pairs(sd, a, b)
        with torch.no_grad():
            got = bench.hot_path(model, a.cuda(), b.cuda()).cpu()
        err = float((got - want).abs().max())
        if err >= 1e-4 and case in ("ptx", "dgcnn"):
            # suspected neighbour flip at a near-tie: every stage must still agree when fed the GPU's inputs
            ferr = forced_ptx(model, sd, a, b) if case == "ptx" else forced_dgcnn(model, sd, a, b)
            assert ferr < 1e-4, (case, n0, pairs, seed, err, "stage-wise", ferr)
            counts[case + "_flip"] = counts.get(case + "_flip", 0) + 1
            worst[case + "_flip"] = max(worst.get(case + "_flip", 0.0), err)
            err = ferr
        if err >= 1e-4 and case == "pt":
            # the xyz-space search ranks by squared differences, the reference by the expanded form: a near-tie of the K-th
            # neighbour flips under the expanded form's own rounding (seed 1234 of round 6: 3.6e-7 between two candidates)
            gaps = pt_near_tie(a, b, bl)
            assert gaps, (case, n0, pairs, seed, err, "no near-tie explains it")
            counts["pt_tie"] = counts.get("pt_tie", 0) + 1
            worst["pt_tie"] = max(worst.get("pt_tie", 0.0), err)
            n += 1
            continue
        worst[case] = max(worst.get(case, 0.0), err)
        counts[case] = counts.get(case, 0) + 1
        assert err < 1e-4, (case, n0, pairs, seed, err)
        n += 1
    print("model fuzz ok: %d forward passes in %.0f s, by kind %s, worst |dlogit| %s"
          % (n, time.time() - t0, counts, {k: "%.1e" % v for k, v in worst.items()}))


if __name__ == "__main__":
    main()
