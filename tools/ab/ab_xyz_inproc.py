"""in ONE process: per-launch times and pass time of a workload with the K-row SA kernel's tables built WITH the coordinate term
(engine.SA_XYZ_TABLES = True: pcr_dense_pm_xyz_f32 / pcr_sa_params.pq_has_xyz, ABI 17) and without, four alternations;
usage: ab_xyz_inproc.py WORKLOAD"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "point-cloud-reid_amd")]
import torch
import bench
from pcr_amd import engine
from pcr_amd import testing as T
wl = sys.argv[1]
desc, kind, n, bl, pairs = bench.WORKLOADS[wl]
model, sd = bench.build_model(kind, bl)
s1, s2 = T.synthetic_pairs(pairs, n, seed=1234, kind="box" if kind == "ssg" else "randn")
s1, s2 = s1.cuda(), s2.cuda()
with torch.no_grad():
    for _ in range(3):
        out = bench.hot_path(model, s1, s2)
torch.cuda.synchronize()
acc, outs = {}, {}
for rep in range(4):
    for flag in (False, True):
        engine.SA_XYZ_TABLES = flag
        tot = bench.profile_kernels(model, s1, s2, reps=5, detail=True)
        torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        with torch.no_grad():
            for _ in range(10):
                out = bench.hot_path(model, s1, s2)
        e1.record(); torch.cuda.synchronize()
        outs[flag] = out.float().cpu()
        acc.setdefault(("pass", flag), []).append(e0.elapsed_time(e1) / 10)
        for k, v in tot.items():
            if "sa_fused" in k or "sa_tables" in k:
                acc.setdefault((k, flag), []).append(v[0] / v[1])
for k in sorted({k for k, _ in acc}):
    a, b = acc[(k, False)], acc[(k, True)]
    print("%-58s in-launch: %s | xyz tables: %s" % (k, " ".join("%.3f" % x for x in a), " ".join("%.3f" % x for x in b)))
with engine.precision("f32"), torch.no_grad():
    ref = bench.hot_path(model, s1, s2).float().cpu()
print("max |dlogit| between the two forms %.2e; against the f32 path: in-launch %.2e, xyz tables %.2e" % (
    float((outs[False] - outs[True]).abs().max()), float((outs[False] - ref).abs().max()), float((outs[True] - ref).abs().max())))
