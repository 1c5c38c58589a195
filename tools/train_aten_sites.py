"""which Python lines issue the ATen ops left in one training step (forward + backward), pt128_train shape:
python tools/train_aten_sites.py   (TorchDispatchMode: every dispatched aten op with the innermost repo frame)"""
import os, sys, traceback, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "point-cloud-reid_amd")]
import torch
from torch.utils._python_dispatch import TorchDispatchMode
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

SKIP = ("aten.view", "aten.detach", "aten.slice", "aten.as_strided", "aten.empty", "aten.select", "aten.permute",
        "aten.transpose", "aten.t.", "aten.expand", "aten.unsqueeze", "aten.squeeze", "aten._unsafe_view", "aten.alias",
        "aten.split", "aten.unbind", "aten.reshape", "aten._reshape_alias", "aten.new_empty", "aten.lift_fresh")
seen = collections.Counter()


class Sites(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if not name.startswith(SKIP):
            site = "?"
            for fr in reversed(traceback.extract_stack()):
                if ("point-cloud-reid_amd" in fr.filename or fr.filename.endswith("bench.py")) and "tools/" not in fr.filename:
                    site = "%s:%d %s" % (os.path.basename(fr.filename), fr.lineno, fr.name)
                    break
            shp = tuple(tuple(a.shape) for a in args if torch.is_tensor(a))[:2]
            seen[(name, site, str(shp))] += 1
        return func(*args, **(kwargs or {}))


with Sites():
    tr.step(data)
torch.cuda.synchronize()
for (nm, site, shp), c in sorted(seen.items(), key=lambda kv: (-kv[1], kv[0])):
    print("%4d  %-28s %-46s %s" % (c, nm, site, shp))
print("total dispatched (non-view) ops:", sum(seen.values()))
