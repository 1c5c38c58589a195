"""per-launch device times of one training step (events on the launch stream), pt128_train shape: python tools/train_detail.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "point-cloud-reid_amd")]
import torch
import bench
from pcr_amd import engine, train, testing as T

pairs, n, bl = 256, 128, [128, 64, 32]
model, _ = bench.build_pt_model(bl)
model.train()
s1, s2 = T.synthetic_pairs(pairs, n, seed=4321, kind="randn")
ids1 = torch.arange(pairs)
ids2 = torch.where(torch.rand(pairs) < 0.5, ids1, ids1 + pairs)
zero = torch.zeros(1, dtype=torch.long, device="cuda")
data = dict(sparse_1=list(s1.cuda()), sparse_2=list(s2.cuda()), dense_1=list(s1.cuda()), dense_2=list(s2.cuda()),
            label_1=[zero] * pairs, label_2=[zero] * pairs, id_1=[i.view(1).cuda() for i in ids1], id_2=[i.view(1).cuda() for i in ids2])
tr = train.Trainer(model, max_iters=20, lr=3e-4, grad_clip=1.0)
for _ in range(3):
    tr.step(data)
torch.cuda.synchronize()
engine.PROFILE = []
tr.step(data)
torch.cuda.synchronize()
rec, engine.PROFILE = engine.PROFILE, None
tot = {}
for name, e0, e1, flops, nbytes, _, _arith in rec:
    t = tot.setdefault(name, [0.0, 0, 0.0, 0.0])
    t[0] += e0.elapsed_time(e1); t[1] += 1; t[2] += flops; t[3] += nbytes
for k, v in sorted(tot.items(), key=lambda kv: -kv[1][0]):
    print("%-52s %8.3f ms x%-3d %7.2f TFLOP/s %8.1f GB/s" % (k, v[0], v[1], v[2] / (v[0] * 1e-3) / 1e12, v[3] / (v[0] * 1e-3) / 1e9))
print("sum", sum(v[0] for v in tot.values()))
