import sys, torch
sys.path.insert(0, "point-cloud-reid_amd"); sys.path.insert(0, ".")
import bench
from pcr_amd import engine, testing as T
from mmdet3d.ops import pointnet_modules as PM
from mmdet3d.ops.point_ops import ball_query_cnt, ball_query_rows
model, sd = bench.build_model("ssg", None)
s1, s2 = T.synthetic_pairs(256, 1024, seed=1234, kind="box")
s1 = s1.cuda()
sa = model.backbone.SA_modules[0]
xyz = s1[..., :3].contiguous()
with torch.no_grad(), engine.precision("bf16x3"):
    PM._NO_ROW_TABLE = False
    x1, f1, i1 = sa(xyz, None)
    PM._NO_ROW_TABLE = True
    x2, f2, i2 = sa(xyz, None)
    print("idx equal", torch.equal(i1, i2))
    d = (f1 - f2).abs()     # (B, C, S)
    print("max diff", float(d.max()))
    bad = (d.amax(1) > 0).nonzero()
    print("bad (cloud, centre) count", len(bad), bad[:30].tolist())
    new_xyz = x1
    idx, cnt = ball_query_cnt(0.0, 0.2, 32, xyz, new_xyz)
    _, cnt2, rows = ball_query_rows(0.2, 32, xyz, new_xyz)
    print("cnt equal", torch.equal(cnt, cnt2), "cnt max", int(cnt.max()), "min", int(cnt.min()))
    if len(bad):
        b, c = bad[0].tolist()
        it = c // 16
        print("cloud", b, "centre", c, "item", it, "cnts of item", cnt[b, it*16:it*16+16].tolist())
        R = int(((cnt[b, it*16:it*16+16].clamp(1, 32) + 1) // 2 * 2).sum())
        print("R", R)
        bads_in_cloud = bad[bad[:, 0] == b][:, 1].tolist()
        print("bad centres in cloud", bads_in_cloud[:40])
    import numpy as np
    tab = rows.view(256, 32, 512, 4)
    nbad = 0
    for (b, c) in bad[:12].tolist():
        it = c // 16
        cn = cnt[b, it*16:it*16+16].clamp(1, 32)
        nr = ((cn + 1) // 2 * 2).tolist()
        start = sum(nr[:c % 16])
        ent = tab[b, it, start:start + nr[c % 16]].cpu()
        ii = ent[:, 0].view(torch.int32).tolist()
        want_i = idx[b, c, :nr[c % 16]].tolist()
        dx = (xyz[b, idx[b, c, :nr[c % 16]].long()] - new_xyz[b, c]).cpu()
        print((b, c), "cnt", int(cnt[b, c]), "tab idx", ii, "want", want_i, "dxyz equal", torch.equal(dx, ent[:, 1:]))
