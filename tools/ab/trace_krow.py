"""trace build (PCR_LIB_TAG=trace: -DPCR_TUNING=1 -DPCR_SA_TRACE_BUILD): one pt1024 pass with PCR_SA_TRACE set; the K-row launches
dump their first 24 blocks' phase stamps (tools/trace_stream.py FILE krow <ordinal>)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "point-cloud-reid_amd")]
import torch
import bench
from pcr_amd import testing as T
wl = sys.argv[1] if len(sys.argv) > 1 else "pt1024"
desc, kind, n, bl, pairs = bench.WORKLOADS[wl]
model, sd = bench.build_model(kind, bl)
s1, s2 = T.synthetic_pairs(pairs, n, seed=1234, kind="randn")
s1, s2 = s1.cuda(), s2.cuda()
with torch.no_grad():
    for _ in range(3):
        bench.hot_path(model, s1, s2)
torch.cuda.synchronize()
