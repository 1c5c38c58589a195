import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "point-cloud-reid_amd"), os.path.join(ROOT, "tests")]
import numpy as np, torch
import bench
from pcr_amd import engine, rows, testing as T
for kind, n in (("pointnet", 256), ("dgcnn", 256)):
    m, _ = bench.build_model(kind, None)
    s1, s2 = T.synthetic_pairs(4, n, 3, "randn")
    both = torch.cat([s1, s2], 0).permute(0, 2, 1).contiguous().cuda()
    out = {}
    with torch.no_grad():
        for enc in ("f32", "bf16x3"):
            with engine.precision(enc):
                xyz, f = m.backbone(both, m.backbone_list)
            for dn in ("f32", "bf16x3"):
                with engine.precision(dn):
                    h = rows.downsample_points(m.downsample, f)
                out[(enc, dn)] = (f.clone(), h.clone())
    f0, h0 = out[("f32", "f32")]
    print(kind, "feat scale %.2f h scale %.2f" % (float(f0.abs().max()), float(h0.abs().max())))
    for k, (f, h) in out.items():
        print("  enc %-7s down %-7s  |d feat| %.2e  |d h| %.2e" % (k[0], k[1], float((f - f0).abs().max()), float((h - h0).abs().max())))
    # per LinearRes stage error with f32 encoder
    with torch.no_grad():
        x = f0
        for i, layer in enumerate(m.downsample):
            outs = {}
            for p in ("f32", "bf16x3"):
                with engine.precision(p):
                    outs[p] = rows.downsample_points(torch.nn.Sequential(layer), x)
            print("  stage %d %s in-scale %.2f out-scale %.2f |d| %.2e" % (i, type(layer).__name__, float(x.abs().max()), float(outs["f32"].abs().max()), float((outs["bf16x3"] - outs["f32"]).abs().max())))
            x = outs["f32"]
