"""experiment: capture one training step (forward + backward [+ update]) into a HIP graph and time its replay"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "point-cloud-reid_amd")]
import torch
import bench
from pcr_amd import engine, train, lazylog, train_ops, testing as T

pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 256
with_opt = len(sys.argv) > 2 and sys.argv[2] == "opt"
n, bl = 128, [128, 64, 32]
model, _ = bench.build_pt_model(bl)
model.train()
s1, s2 = T.synthetic_pairs(pairs, n, seed=4321, kind="randn")
ids1 = torch.arange(pairs)
ids2 = torch.where(torch.rand(pairs) < 0.5, ids1, ids1 + pairs)
zero = torch.zeros(1, dtype=torch.long, device="cuda")
data = dict(sparse_1=list(s1.cuda()), sparse_2=list(s2.cuda()), dense_1=list(s1.cuda()), dense_2=list(s2.cuda()),
            label_1=[zero] * pairs, label_2=[zero] * pairs, id_1=[i.view(1).cuda() for i in ids1], id_2=[i.view(1).cuda() for i in ids2])
tr = train.Trainer(model, max_iters=1000, lr=3e-4, grad_clip=1.0)
for _ in range(3):
    tr.step(data)
torch.cuda.synchronize()
t0 = time.time()
for _ in range(10):
    tr.step(data)
torch.cuda.synchronize()
print("eager: %.2f ms/step" % ((time.time() - t0) * 100))

orig_add = lazylog.LazyScalars.add_device
def add_device(self, names, values, ints=None):
    if torch.cuda.is_current_stream_capturing():
        self._static = getattr(self, "_static", []) + [(list(names), values)]
        return
    return orig_add(self, names, values, ints)
lazylog.LazyScalars.add_device = add_device

g = torch.cuda.CUDAGraph()
tr.optimizer.zero_grad(set_to_none=True)
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    with torch.cuda.graph(g, stream=side, capture_error_mode="relaxed"):
        train_ops.prepack(model)
        out = model.train_step(data, None)
        out["loss"].backward()
        if with_opt:
            tr._set_hyper()
            norm = tr.optimizer.step(max_norm=tr.grad_clip)
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
print("captured; loss at capture", float(out["loss"]))
for _ in range(3):
    g.replay()
torch.cuda.synchronize()
t0 = time.time()
for _ in range(20):
    g.replay()
torch.cuda.synchronize()
print("graph replay (%s): %.2f ms/step" % ("fwd+bwd+update" if with_opt else "fwd+bwd", (time.time() - t0) * 50))
print("loss after replays", float(out["loss"]))
