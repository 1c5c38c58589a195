"""calibration batch against the next batch: the guard level a model calibrates to, the deviation it measured there, and what a fresh
batch of the same distribution returns at levels 0 and 1 (GPU box; python tools/guard_cases.py)"""
import sys, json
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "point-cloud-reid_amd"), os.path.join(ROOT, "tests")]
import torch, bench
from pcr_amd import engine, testing as T
import test_gpu_precision as TP
for kind, n, seeds in (("pt", 128, (0, 1, 2)), ("pointnet", 256, (0, 1, 2)), ("pt", 1024, (0, 1))):
    bl = {128: [128, 64, 32], 1024: [1024, 512, 256]}.get(n) if kind == "pt" else None
    for seed in seeds:
        model, _ = bench.build_model(kind, bl)
        base = T.seeded_state_dict(T.manifest_of(model), seed)
        model.load_state_dict(TP._rescale_bn(base, 1.0, 1.0), strict=True)
        model = model.cuda().eval()
        for scale in (0.1, 1.0, 10.0):
            dist = "box" if kind in ("ssg", "pt") else "randn"
            c1, c2 = T.synthetic_pairs(6, n, seed=40 + seed, kind=dist)
            s1, s2 = T.synthetic_pairs(6, n, seed=140 + seed, kind=dist)
            with engine.precision("bf16x3"):
                st = model.calibrate_precision((c1 * scale).cuda(), (c2 * scale).cuda())
                fresh = {}
                with engine.precision("f32"):
                    ref = TP._logits(model, s1 * scale, s2 * scale)
                for lv in (0, 1):
                    model.__dict__["_pcr_guard"]["level"] = lv
                    fresh[lv] = float((TP._logits(model, s1 * scale, s2 * scale) - ref).abs().max())
            print(kind, n, "seed", seed, "scale", scale, "calibration", {k: "%.2e" % v for k, v in st["dlogit"].items()}, "level", st["level"],
                  "fresh batch at level 0 / 1: %.2e / %.2e" % (fresh[0], fresh[1]), "max|logit| %.2f" % float(ref.abs().max()))
