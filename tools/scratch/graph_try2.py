import os, sys, faulthandler
faulthandler.enable()
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "point-cloud-reid_amd")]
import torch
import bench
from pcr_amd import train, testing as T
order = [bool(int(c)) for c in sys.argv[1]]
same_batch = len(sys.argv) > 2 and sys.argv[2] == "same"
PAIRS = int(os.environ.get('GP', '8'))
def batch(seed, pairs=None, n=128):
    pairs = pairs or PAIRS
    s1, s2 = T.synthetic_pairs(pairs, n, seed=seed, kind="randn")
    ids1 = torch.arange(pairs)
    ids2 = torch.where(torch.arange(pairs) % 2 == 0, ids1, ids1 + 100)
    zero = torch.zeros(1, dtype=torch.long, device="cuda")
    return dict(sparse_1=list(s1.cuda()), sparse_2=list(s2.cuda()), dense_1=list(s1.cuda()), dense_2=list(s2.cuda()),
                label_1=[zero] * pairs, label_2=[zero] * pairs,
                id_1=[i.view(1).cuda() for i in ids1], id_2=[i.view(1).cuda() for i in ids2])
fixed = batch(5)
for mode in order:
    m, _ = bench.build_pt_model([128, 64, 32])
    m.train()
    tr = train.Trainer(m, max_iters=12, lr=1e-3, grad_clip=1.0, graph=mode)
    tr.graph_warmup = int(os.environ.get('GW', '3'))
    for it in range(7):
        out = tr.step(fixed if same_batch else batch(10 + it))
        print(mode, it, float(out["loss"]), flush=True)
    print("graph still on:", tr.graph, flush=True)
