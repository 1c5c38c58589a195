"""per-launch device times of one inference pass of a bench workload (events on the launch stream):
python tools/infer_detail.py pointnet256 [pairs]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "point-cloud-reid_amd")]
import torch
import bench
from pcr_amd import testing as T
wl = sys.argv[1] if len(sys.argv) > 1 else "pointnet256"
desc, kind, n, bl, dpairs = bench.WORKLOADS[wl]
pairs = int(sys.argv[2]) if len(sys.argv) > 2 else dpairs
model, sd = bench.build_model(kind, bl)
s1, s2 = T.synthetic_pairs(pairs, n, seed=1234, kind="box" if kind == "ssg" else "randn")
s1, s2 = s1.cuda(), s2.cuda()
with torch.no_grad():
    for _ in range(2):
        bench.hot_path(model, s1, s2)
torch.cuda.synchronize()
tot = bench.profile_kernels(model, s1, s2, reps=3, detail=True)
for k, v in sorted(tot.items(), key=lambda kv: -kv[1][0]):
    print("%-56s %8.3f ms x%-3d %8.1f TFLOP/s(ref ops) %8.1f GB/s  %s" % (k, v[0], v[1], v[2] / (v[0] * 1e-3) / 1e12, v[3] / (v[0] * 1e-3) / 1e9, v[5]))
print("sum", sum(v[0] for v in tot.values()), "pairs", pairs)
