import sys, torch
sys.path.insert(0, "point-cloud-reid_amd"); sys.path.insert(0, ".")
import bench
from pcr_amd import engine, testing as T
model, sd = bench.build_model("ssg", None)
s1, s2 = T.synthetic_pairs(256, 1024, seed=1234, kind="box")
s1, s2 = s1.cuda(), s2.cuda()
res = {}
for prec in ("f32", "bf16x3"):
    with torch.no_grad(), engine.precision(prec):
        xyz = s1[..., :3].contiguous()
        feats = None
        outs = []
        bb = model.backbone
        for sa in bb.SA_modules:
            xyz, feats, _ = sa(xyz, feats)
            outs.append(feats.contiguous().clone())
        x2, h = bb(s1)
        outs.append(h.clone())
        lg = bench.hot_path(model, s1, s2)
        outs.append(lg.clone())
    res[prec] = outs
for i, (a, b) in enumerate(zip(res["f32"], res["bf16x3"])):
    d = (a - b).abs()
    print(i, tuple(a.shape), float(d.max()), float(a.abs().max()))
a, b = res["f32"][2], res["bf16x3"][2]
d = (a - b).abs().amax(dim=(1, 2)); print("bad clouds", (d > 1e-3).nonzero().flatten()[:20].tolist())
