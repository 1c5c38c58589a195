"""per-launch times of the gallery128 step (bench.profile_kernels over forward_inference + match_gallery)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "point-cloud-reid_amd")]
import torch
import bench
from pcr_amd import testing as T, engine
desc, kind, n, bl, pairs = bench.WORKLOADS["gallery128"]
G = int(round(pairs ** 0.5))
model, sd = bench.build_pt_model(bl)
clouds = T.synthetic_clouds(2 * G, n, seed=1234, kind="randn").cuda()
ii, jj = torch.meshgrid(torch.arange(G), torch.arange(G, 2 * G), indexing="ij")
combos = torch.stack([ii.reshape(-1), jj.reshape(-1)], dim=1).cuda()


def step():
    xyz, h = model.forward_inference(clouds)
    return model.match_gallery(h, xyz, combos)


with torch.no_grad():
    for _ in range(3):
        step()
torch.cuda.synchronize()
for rep in range(2):
    tot = bench.profile_kernels(None, None, None, reps=5, detail=True, fn=step)
    for k, v in sorted(tot.items(), key=lambda kv: -kv[1][0])[:8]:
        print("%-64s %8.3f ms x%d" % (k, v[0], v[1]))
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    with torch.no_grad():
        for _ in range(10):
            step()
    e1.record(); torch.cuda.synchronize()
    print("eager step %.3f ms, launches summed %.3f ms" % (e0.elapsed_time(e1) / 10, sum(v[0] for v in tot.values())))
