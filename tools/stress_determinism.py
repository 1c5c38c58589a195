"""the hot path on the same batch, N times over: every pass must return the bits of the first one (no float atomics, fixed
reduction orders, shape-only dispatch) -- a soak for sporadic faults (round 6: the table kernels' packed-fma halves,
sa_kernels_impl.h).  GPU box: python tools/stress_determinism.py [passes per workload]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "point-cloud-reid_amd")]
import torch
import bench
from pcr_amd import testing as T
n_pass = int(sys.argv[1]) if len(sys.argv) > 1 else 100
bad_total = 0
for wl in ("pt1024", "pt4096", "ssg1024", "pointnet256", "pt128", "dgcnn256"):
    desc, kind, n, bl, pairs = bench.WORKLOADS[wl]
    model, sd = bench.build_model(kind, bl)
    s1, s2 = T.synthetic_pairs(pairs, n, seed=4321, kind="box" if kind == "ssg" else "randn")
    s1, s2 = s1.cuda(), s2.cuda()
    with torch.no_grad():
        first = bench.hot_path(model, s1, s2).clone()
        bad = 0
        for _ in range(n_pass):
            out = bench.hot_path(model, s1, s2)
            bad += int(not torch.equal(out, first))
    torch.cuda.synchronize()
    print("%-12s %d pairs x %d passes: %d passes differ from the first; guard level %s" % (wl, pairs, n_pass, bad, (model.guard_state() or {}).get("level")))
    bad_total += bad
    del model, s1, s2
    torch.cuda.empty_cache()
sys.exit(1 if bad_total else 0)
