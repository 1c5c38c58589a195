import sys, os, time, cProfile, pstats
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch, bench
from pcr_amd import testing as T, engine
model, sd = bench.build_model("pt", [128, 64, 32])
s1, s2 = T.synthetic_pairs(512, 128, seed=1, kind="randn")
s1, s2 = s1.cuda(), s2.cuda()
for prec in ("f32", "bf16x3"):
    with engine.precision(prec), torch.no_grad():
        for _ in range(3): bench.hot_path(model, s1, s2)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20): bench.hot_path(model, s1, s2)
        t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
        print(prec, "host ms/step %.3f  total ms/step %.3f" % ((t1 - t0) / 20 * 1e3, (t2 - t0) / 20 * 1e3))
with torch.no_grad():
    pr = cProfile.Profile(); pr.enable()
    for _ in range(20): bench.hot_path(model, s1, s2)
    pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(25)
