import sys, torch
sys.path.insert(0, "point-cloud-reid_amd"); sys.path.insert(0, ".")
import bench
from pcr_amd import engine, testing as T
from mmdet3d.ops.point_ops import ball_query_cnt, ball_query_rows, furthest_point_sample, gather_points
model, sd = bench.build_model("ssg", None)
s1, s2 = T.synthetic_pairs(256, 1024, seed=1234, kind="box")
xyz = s1[..., :3].contiguous().cuda()
sa = model.backbone.SA_modules[0]
with torch.no_grad(), engine.precision("bf16x3"):
    new_xyz, indices = sa._sample_points(xyz, None, None, None)
    idx, cnt = ball_query_cnt(0.0, 0.2, 32, xyz, new_xyz)
    tabs = []
    for r in range(4):
        _, c2, rows = ball_query_rows(0.2, 32, xyz, new_xyz)
        tabs.append((c2.clone(), rows.clone()))
    # compare written entries of the tables
    cn = cnt.clamp(1, 32); nr = (cn + 1) // 2 * 2
    R = nr.view(256, 32, 16).sum(-1)                       # rows per item
    pos = torch.arange(512, device="cuda").view(1, 1, 512)
    mask = pos < ((R + 31) // 32 * 32).unsqueeze(-1)       # written entries
    for r in range(4):
        t = tabs[r][1].view(256, 32, 512, 4)
        t0 = tabs[0][1].view(256, 32, 512, 4)
        diff = ((t.view(torch.int32) != t0.view(torch.int32)).any(-1) & mask)
        print("table", r, "cnt equal", torch.equal(tabs[r][0], cnt), "entries differing from call 0:", int(diff.sum()))
    plan = sa._plan(0, xyz.device)
    outs = [plan.run(xyz, None, None, centre_idx=indices.contiguous(), cnt=tabs[0][0], rows=tabs[0][1], K=32, out_point_major=True).contiguous().clone() for _ in range(4)]
    ref = plan.run(xyz, None, idx, centre_idx=indices.contiguous(), cnt=cnt, out_point_major=True).contiguous()
    for o in outs:
        print("SA from one table: equal to first", torch.equal(o, outs[0]), "bad centres vs indexed", int(((o - ref).abs().amax(1) > 0).sum()))
    t1 = tabs[1][1].view(256, 32, 512, 4); t0 = tabs[0][1].view(256, 32, 512, 4)
    diff = ((t1.view(torch.int32) != t0.view(torch.int32)).any(-1) & mask).nonzero()
    starts = (nr.view(256, 32, 16).cumsum(-1) - nr.view(256, 32, 16))
    for (b, it, r) in diff[:16].tolist():
        st = starts[b, it].tolist(); nrs = nr.view(256, 32, 16)[b, it].tolist(); cns = cnt.view(256, 32, 16)[b, it].tolist()
        c = max(i for i in range(16) if st[i] <= r) if r < sum(nrs) else -1
        k = r - st[c] if c >= 0 else -1
        print((b, it, r), "centre", c, "k", k, "cnt", cns[c] if c >= 0 else None, "R", sum(nrs),
              "t0", t0[b, it, r, :1].view(torch.int32).tolist(), [round(x, 4) for x in t0[b, it, r, 1:].tolist()],
              "t1", t1[b, it, r, :1].view(torch.int32).tolist(), [round(x, 4) for x in t1[b, it, r, 1:].tolist()],
              "idx", idx[b, it * 16 + c, k].item() if c >= 0 else None)
