import torch, sys
sys.path.insert(0, "point-cloud-reid_amd")
from pcr_amd import engine
g = torch.Generator().manual_seed(1)
for B in (64, 512, 1000, 4096):
    x_pm = torch.randn(B, 128, 256, generator=g).cuda()
    w = torch.randn(64, 256, generator=g) / 16
    bias = torch.randn(64, generator=g).cuda()
    wp = engine.pack_weight_dual(w, torch.device("cuda"))
    v = x_pm.transpose(1, 2)
    with engine.precision("f32"):
        a = engine.dense(v, wp, 64, None, bias, 0)
    with engine.precision("bf16x3"):
        b = engine.dense(v, wp, 64, None, bias, 0)
    d = (a - b).abs().amax(dim=(1, 2))
    bad = (d > 1e-3).nonzero().flatten()
    print(B, float(d.max()), bad[:10].tolist(), len(bad))
    if len(bad):
        bb = int(bad[0]); dd = (a[bb] - b[bb]).abs()
        print(" cloud", bb, "bad tokens", (dd.amax(0) > 1e-3).nonzero().flatten()[:40].tolist(), "bad chans", (dd.amax(1) > 1e-3).nonzero().flatten()[:40].tolist())
