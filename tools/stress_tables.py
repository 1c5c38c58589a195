"""pcr_dense_pm_xyz_f32: the persistent kernel (channel-major input) against the one-shot kernel (point-major input of the same
values), bit for bit, on batches with far more tiles than resident workgroups, many times over (GPU box):
python tools/stress_tables.py [repeats]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "point-cloud-reid_amd")]
import torch
from pcr_amd import _lib as L, engine
lib = L.load()
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
total = bad = 0
for rep in range(reps):
    for (D, cout, B, N, q_rows, q_off) in ((32, 128, 1024, 1024, 512, 64), (64, 256, 1024, 512, 256, 128), (32, 128, 300, 256, 256, 128),
                                          (64, 128, 777, 128, 64, 64), (32, 256, 512, 2048, 1024, 128)):
        g = torch.Generator().manual_seed(rep * 10 + D + cout)
        feat = torch.randn(B, D, N, generator=g).cuda()
        feat_pm = feat.transpose(1, 2).contiguous()
        xyz = torch.randn(B, N, 3, generator=g).cuda()
        w64 = (torch.randn(cout, D, generator=g) * 0.2)
        wp = engine.pack_weight_bf(w64, torch.device("cuda"))
        w64 = w64.double()
        wxyz = torch.randn(cout, 4, generator=g).cuda()
        ys = []
        for x, pm in ((feat, 0), (feat_pm, 1)):
            y = torch.full((B, N, cout), float("nan"), device="cuda")
            L.check(lib.pcr_dense_pm_xyz_f32(L.ptr(x), L.ptr(wp), L.ptr(xyz), L.ptr(wxyz), L.ptr(y), B, D, cout, N, pm, 1, q_rows, q_off,
                                             L.stream_ptr()), "pcr_dense_pm_xyz_f32")
            ys.append(y)
        same = (ys[0] == ys[1]) | (torch.isnan(ys[0]) & torch.isnan(ys[1]))
        total += same.numel()
        nb = int((~same).sum())
        bad += nb
        if nb:
            idx = (~same).nonzero()
            want = (torch.einsum("od,bdn->bno", (engine_w := None) or w64.cuda(), feat.double()) +
                    torch.einsum("oc,bnc->bno", wxyz[:, :3].double(), xyz.double()) + wxyz[:, 3].double())
            e0 = (ys[0].double() - want).abs()[~same]
            e1 = (ys[1].double() - want).abs()[~same]
            print("  rep %d shape %s: %d differing; |persistent - torch| max %.2e, |one-shot - torch| max %.2e; first %s; clouds %s tokens %s couts %s" % (
                rep, (D, cout, B, N, q_rows, q_off), nb, float(e0.max()), float(e1.max()), idx[:3].tolist(),
                torch.unique(idx[:, 0])[:6].tolist(), torch.unique(idx[:, 1])[:12].tolist(), torch.unique(idx[:, 2])[:12].tolist()))
        del ys, feat, feat_pm
print("stress_tables: %d repeats, %.2e elements compared, %d differing" % (reps, total, bad))
sys.exit(1 if bad else 0)
