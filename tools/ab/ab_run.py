"""one process of an A/B: per-launch times of a workload's pass + a checksum of its logits (env knobs are read once per process)"""
import os, sys, hashlib
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "point-cloud-reid_amd")]
import torch
import bench
from pcr_amd import testing as T
wl = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
desc, kind, n, bl, dpairs = bench.WORKLOADS[wl]
pairs = int(sys.argv[3]) if len(sys.argv) > 3 else dpairs
model, sd = bench.build_model(kind, bl)
s1, s2 = T.synthetic_pairs(pairs, n, seed=1234, kind="box" if kind == "ssg" else "randn")
s1, s2 = s1.cuda(), s2.cuda()
with torch.no_grad():
    for _ in range(3):
        out = bench.hot_path(model, s1, s2)
torch.cuda.synchronize()
tot = bench.profile_kernels(model, s1, s2, reps=5, detail=True)
tag = " ".join("%s=%s" % (k, v) for k, v in os.environ.items() if k.startswith("PCR_") and k != "PCR_LIB_TAG")
for k, v in sorted(tot.items(), key=lambda kv: -kv[1][0]):
    if flt in k:
        print("%-28s %-56s %8.3f ms x%d" % (tag, k, v[0], v[1]))
print("%-28s sum %.3f ms  logits sha %s" % (tag, sum(v[0] for v in tot.values()), hashlib.sha1(out.cpu().numpy().tobytes()).hexdigest()[:12]))
