import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "oracle"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "point-cloud-reid_amd"))
import torch
from pcr_amd import engine, testing as T
from mmdet3d.models.pointnet2_utils import Self_Attention
for (B, Lq) in [(5, 512), (5, 128), (1, 512), (16, 512), (5, 1024)]:
    g = torch.Generator().manual_seed(64 + Lq)
    m = Self_Attention(64, 2)
    sd = T.seeded_state_dict(T.manifest_of(m), 3)
    m.load_state_dict(sd); m = m.cuda().eval()
    x, xyz = torch.randn(B, 64, Lq, generator=g).cuda(), torch.randn(B, Lq, 3, generator=g).cuda()
    outs = {}
    for prec in ("f32", "bf16x3"):
        with engine.precision(prec), torch.no_grad():
            outs[prec] = m(x, xyz).cpu()
    d = (outs["f32"] - outs["bf16x3"]).abs()
    print("B", B, "Lq", Lq, "max", float(d.max()), "per-cloud max", [round(float(d[b].max()), 5) for b in range(min(B, 6))])
    if float(d.max()) > 1e-3:
        b = int(d.amax(dim=(1, 2)).argmax())
        print("  bad cloud", b, "bad channels", (d[b].amax(dim=1) > 1e-3).nonzero().flatten().tolist()[:40])
        print("  bad token blocks", sorted(set(((d[b].amax(dim=0) > 1e-3).nonzero().flatten() // 32).tolist())))
