"""fused vs unfused attention tail against float64 torch autograd: error of every gradient, both ways"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "tests"), os.path.join(ROOT, "point-cloud-reid_amd")]
import copy, torch
import test_gpu_train_chain as TC
from pcr_amd import train_ops as TO
for (d, c1, hid, out, resid) in TC.SHAPES:
    B, Ln = 16, 128
    m = TC.Tail(d, c1, hid, out, seed=3).cuda()
    g = torch.Generator().manual_seed(5)
    msg, res = torch.randn(B, d, Ln, generator=g).cuda(), torch.randn(B, c1, Ln, generator=g).cuda()
    go = torch.randn(B, out, Ln, generator=g).cuda()
    of, gf = TC._run(lambda mm, a, b, r: TO.attn_tail(mm, a, b, r), m, msg, res, resid, go)
    ou, gu = TC._run(TC._unfused, m, msg, res, resid, go)
    m64 = copy.deepcopy(m).double()
    o64, g64 = TC._run(TC._torch, m64, msg.double(), res.double(), resid, go.double())
    ot, gt = TC._run(TC._torch, m, msg, res, resid, go)
    print((d, c1, hid, out, resid), "out: fused %.2e unfused %.2e torch32 %.2e" % (TC._rel(of.double(), o64), TC._rel(ou.double(), o64), TC._rel(ot.double(), o64)))
    for k in g64:
        print("   %-14s fused %.2e unfused %.2e torch32 %.2e" % (k, TC._rel(gf[k].double(), g64[k]), TC._rel(gu[k].double(), g64[k]), TC._rel(gt[k].double(), g64[k])))
