"""sensitivity of the 4-step trajectory of test_training_steps_agree... to a 1e-7 perturbation, per training arithmetic"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "point-cloud-reid_amd")]
import torch
import bench
from pcr_amd import train, train_ops as TO, testing as T
s1, s2 = T.synthetic_pairs(8, 128, seed=2, kind="randn")
dev = "cuda"
ids1, ids2 = torch.arange(8), torch.tensor([0, 1, 2, 3, 9, 9, 9, 9])
data = dict(sparse_1=list(s1.to(dev)), sparse_2=list(s2.to(dev)), dense_1=list(s1.to(dev)), dense_2=list(s2.to(dev)),
            label_1=[torch.zeros(1, dtype=torch.long, device=dev)] * 8, label_2=[torch.zeros(1, dtype=torch.long, device=dev)] * 8,
            id_1=[i.view(1).to(dev) for i in ids1], id_2=[i.view(1).to(dev) for i in ids2])
def hx(m, a, b):
    xyz1, xyz2, h1, h2 = m.siamese_forward(a, b)
    return h1, h2, xyz1, xyz2
for prec in ("f32", "bf16x3", "bf16x3_all"):
    TO.set_train_precision(prec)
    res = []
    for variant in ("hip", "hip", "hip+eps", "torch"):
        m, _ = bench.build_pt_model([128, 64, 32]); m.train()
        if variant == "hip+eps":
            with torch.no_grad():
                for p in m.parameters():
                    p.mul_(1.0 + 1e-7)
        tr = train.Trainer(m, max_iters=8, lr=1e-3, grad_clip=1.0, fused=(variant != "torch"))
        losses = [float(tr.step(data)["loss"].detach()) for _ in range(4)]
        m.eval()
        with torch.no_grad():
            e = m.match_forward_inference(*hx(m, s1.to(dev), s2.to(dev)))
        res.append((variant, losses, e))
    base = res[0]
    for v, l, e in res[1:]:
        print(prec, "hip vs", v, "loss diffs", ["%.1e" % abs(a - b) for a, b in zip(base[1], l)], "eval logit diff %.2e" % float((e - base[2]).abs().max()))
