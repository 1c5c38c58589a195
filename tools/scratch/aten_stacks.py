import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "point-cloud-reid_amd")]
import torch
import bench
from pcr_amd import train, testing as T
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
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    tr.step(data)
    torch.cuda.synchronize()
ka = prof.key_averages(group_by_input_shape=True)
want = ("aten::copy_", "aten::fill_", "aten::add_", "aten::add", "aten::cat", "aten::clone", "aten::zeros", "aten::contiguous", "aten::zero_", "aten::roll", "aten::stack", "aten::to", "aten::_to_copy", "aten::sub", "aten::neg", "aten::mul", "aten::div", "aten::sum", "aten::mean", "aten::where", "aten::zeros_like", "aten::empty_like", "aten::slice_backward", "aten::select_backward", "aten::t", "aten::transpose")
rows = [e for e in ka if e.key in want]
rows.sort(key=lambda e: (e.key, -e.count))
for e in rows:
    print("%4d %-22s %s" % (e.count, e.key, str(e.input_shapes)[:150]))
