"""in ONE process (tuning build, PCR_LIB_TAG=tune): per-launch times of a workload's pass under two values of PCR_SA_DBG (re-read per
launch), alternating four times; usage: ab_dbg_inproc.py WORKLOAD DBG_A DBG_B [name filter]"""
import os, sys, hashlib
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "point-cloud-reid_amd")]
import torch
import bench
from pcr_amd import testing as T
wl, da, db = sys.argv[1], sys.argv[2], sys.argv[3]
flt = sys.argv[4] if len(sys.argv) > 4 else "sa_fused"
var = os.environ.get("PCR_AB_VAR", "PCR_SA_DBG")
desc, kind, n, bl, pairs = bench.WORKLOADS[wl]
model, sd = bench.build_model(kind, bl)
s1, s2 = T.synthetic_pairs(pairs, n, seed=1234, kind="box" if kind == "ssg" else "randn")
s1, s2 = s1.cuda(), s2.cuda()
os.environ[var] = da
with torch.no_grad():
    for _ in range(3):
        out = bench.hot_path(model, s1, s2)
torch.cuda.synchronize()
acc, sha = {}, {}
for rep in range(4):
    for d in (da, db):
        os.environ[var] = d
        tot = bench.profile_kernels(model, s1, s2, reps=5, detail=True)
        torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        with torch.no_grad():
            for _ in range(10):
                out = bench.hot_path(model, s1, s2)
        e1.record(); torch.cuda.synchronize()
        sha[d] = hashlib.sha1(out.float().cpu().numpy().tobytes()).hexdigest()[:12]
        acc.setdefault(("pass", d), []).append(e0.elapsed_time(e1) / 10)
        for k, v in tot.items():
            if flt in k:
                acc.setdefault((k, d), []).append(v[0] / v[1])
for k in sorted({k for k, _ in acc}):
    a, b = acc[(k, da)], acc[(k, db)]
    print("%-58s %s=%s: %s | %s: %s" % (k, var, da, " ".join("%.3f" % x for x in a), db, " ".join("%.3f" % x for x in b)))
print("logits sha", sha)
