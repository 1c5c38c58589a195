import sys, torch
sys.path.insert(0, "point-cloud-reid_amd"); sys.path.insert(0, ".")
import bench
from pcr_amd import engine, testing as T
from mmdet3d.ops import pointnet_modules as PM
model, sd = bench.build_model("ssg", None)
s1, s2 = T.synthetic_pairs(256, 1024, seed=1234, kind="box")
s1 = s1.cuda()
sa = model.backbone.SA_modules[0]
xyz = s1[..., :3].contiguous()
with torch.no_grad():
    with engine.precision("f32"):
        _, fr, _ = sa(xyz, None)
    fr = fr.contiguous()
    with engine.precision("bf16x3"):
        for flag in (False, True):
            PM._NO_ROW_TABLE = flag
            outs = [sa(xyz, None)[1].contiguous().clone() for _ in range(4)]
            for o in outs:
                d = (o - fr).abs().amax(1)
                print("no_tab" if flag else "tab", "max vs f32", float(d.max()), "bad centres", int((d > 1e-3).sum()),
                      "equal to first run", torch.equal(o, outs[0]))
