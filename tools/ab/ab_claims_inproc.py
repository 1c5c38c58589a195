"""in ONE process: per-launch times of the pt1024 pass with engine.SA_CLAIMS toggled, alternating (same clocks, same allocations)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "point-cloud-reid_amd")]
import torch
import bench
from pcr_amd import testing as T, engine
wl = sys.argv[1] if len(sys.argv) > 1 else "pt1024"
desc, kind, n, bl, pairs = bench.WORKLOADS[wl]
model, sd = bench.build_model(kind, bl)
s1, s2 = T.synthetic_pairs(pairs, n, seed=1234, kind="randn")
s1, s2 = s1.cuda(), s2.cuda()
with torch.no_grad():
    for _ in range(3):
        out = bench.hot_path(model, s1, s2)
torch.cuda.synchronize()
acc = {}
for rep in range(4):
    for on in (True, False):
        engine.SA_CLAIMS = on
        tot = bench.profile_kernels(model, s1, s2, reps=5, detail=True)
        # whole pass, wall clock over 10 eager passes
        torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        with torch.no_grad():
            for _ in range(10):
                bench.hot_path(model, s1, s2)
        e1.record(); torch.cuda.synchronize()
        acc.setdefault(("pass", on), []).append(e0.elapsed_time(e1) / 10)
        for k, v in tot.items():
            if "sa_fused" in k:
                acc.setdefault((k, on), []).append(v[0] / v[1])
for k in sorted({k for k, _ in acc}):
    a, b = acc[(k, True)], acc[(k, False)]
    print("%-60s claims on %s  | off %s" % (k, " ".join("%.3f" % x for x in a), " ".join("%.3f" % x for x in b)))
