"""split-bf16 guard: per-plan calibrated deviation and the logit error it leaves, over the margin sweep's cases
(tests/test_gpu_precision.py): python tools/guard_sweep.py [kinds...]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "point-cloud-reid_amd"), os.path.join(ROOT, "tests")]
import torch
import bench
from pcr_amd import engine, testing as T
from test_gpu_precision import _rescale_bn, _logits

kinds = sys.argv[1:] or ["ssg", "pt128", "pt1024"]
for kind in kinds:
    k, n = ("pt", int(kind[2:])) if kind.startswith("pt") and kind != "pointnet" else (kind, {"ssg": 1024, "pointnet": 256, "dgcnn": 256}[kind])
    bl = {128: [128, 64, 32], 1024: [1024, 512, 256]}.get(n) if k == "pt" else None
    for seed in (0, 1, 2):
        model, _ = bench.build_model(k, bl)
        base = T.seeded_state_dict(T.manifest_of(model), seed)
        for bn_name, (vs, ms) in (("seeded", (1.0, 1.0)), ("var/4,mean*3", (0.25, 3.0)), ("var*4", (4.0, 1.0)), ("var/16,mean*5", (1 / 16.0, 5.0))):
            if seed and bn_name != "seeded":
                continue
            for scale in (1.0,) if bn_name != "seeded" else (0.1, 1.0, 10.0):
                s1, s2 = T.synthetic_pairs(6, n, seed=40 + seed, kind="box" if k in ("ssg", "pt") else "randn")
                s1, s2 = s1 * scale, s2 * scale
                rec = {}
                for tau in (1e9, float(os.environ.get("TAU", "6e-6"))):
                    engine.GUARD_TAU = tau
                    engine._GUARD_LOG.clear()
                    model.load_state_dict(_rescale_bn(base, vs, ms), strict=True)     # (new parameter versions: new plans)
                    model = model.cuda().eval()
                    with engine.precision("f32"):
                        ref = _logits(model, s1, s2)
                    with engine.precision("bf16x3"):
                        got = _logits(model, s1, s2)
                        got2 = _logits(model, s1, s2)
                    rep = engine.guard_report()
                    rec["tau=%g" % tau] = dict(dlogit=float((got - ref).abs().max()), second=float((got2 - ref).abs().max()),
                                               rel={kk: float("%.2e" % v["rel"]) for kk, v in rep.items()},
                                               f32=[kk for kk, v in rep.items() if v["f32"]])
                print(kind, seed, bn_name, "x%g" % scale, "|logit|max %.2f" % float(ref.abs().max()), json.dumps(rec))
