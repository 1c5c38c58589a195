"""Randomised sweep of the grouped set-abstraction kernels on the GPU box (not part of the test suite): random
(mode, D, K, S, N, widths), kNN-style rows (all K valid) and ball-query-style rows (first cnt genuine, rest repeat the
first -- ragged persistent kernel), channel- and point-major features / outputs, against plain torch fp32.
python tools/fuzz_sa.py [seconds] [seed]"""
import os
import sys
import time

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "point-cloud-reid_amd")]
from pcr_amd import engine, testing as T   # noqa: E402


def main(budget=None, seed=None, max_cases=None):
    """budget seconds / seed from the command line when not given; max_cases bounds the sweep for the pytest slice
    (tests/test_gpu_fuzz.py: fixed seeds, fixed case counts)"""
    if budget is None:
        budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    rng = np.random.default_rng(seed if seed is not None else (int(sys.argv[2]) if len(sys.argv) > 2 else 0))
    t0, n, worst, kinds = time.time(), 0, 0.0, {}
    while time.time() - t0 < budget and (max_cases is None or n < max_cases):
        g = torch.Generator().manual_seed(int(rng.integers(0, 1 << 30)))
        mode = int(rng.integers(0, 2))
        D = int(rng.choice([0, 3, 8, 16, 32, 64, 128]))
        K = int(rng.choice([4, 16, 20, 32, 48, 64]))
        N = int(rng.integers(max(K, 8), 600))
        S = int(rng.integers(1, N + 1)) if mode == 1 else int(rng.integers(1, N + 1))
        widths = [(32, 32, 32), (64, 64, 64), (128, 128, 128), (64, 64, 128), (128, 128, 256), (24, 40, 72),
                  (32, 64, 128), (256, 256, 256), (512, 512, 512), (128, 320, 384)][rng.integers(0, 10)]
        if max(widths) > 256 or (widths[0] > 128 and mode == 0):     # the wide (mul = 2 / 4) layers: 64 rows at most per tile
            D = int(rng.choice([0, 64, 128, 256]))
        B = int(rng.integers(1, 4))
        cin = 3 + (2 * D if mode == 0 else D)
        convs, bns, last = [], [], cin
        for w in widths:
            c, b = nn.Conv2d(last, w, 1), nn.BatchNorm2d(w)
            with torch.no_grad():
                b.running_mean.copy_(torch.randn(w, generator=g) * 0.1)
                b.running_var.copy_(torch.rand(w, generator=g) + 0.5)
                b.weight.copy_(1 + 0.1 * torch.randn(w, generator=g))
                b.bias.copy_(0.1 * torch.randn(w, generator=g))
            convs.append(c)
            bns.append(b.eval())
            last = w
        xyz = T.synthetic_clouds(B, N, int(rng.integers(0, 1 << 30)), "box")
        feat = torch.randn(B, D, N, generator=g) if D else None
        idx = torch.randint(0, N, (B, S, K), generator=g, dtype=torch.int32)
        cnt = None
        if mode == 1 and rng.random() < 0.6:          # ball-query rows: entries [cnt, K) repeat entry 0
            cnt = torch.randint(1, K + 1, (B, S), generator=g, dtype=torch.int32)
            if rng.random() < 0.5:
                cnt = torch.clamp(cnt, max=int(rng.integers(1, 6)))        # sparse groups, like r = 0.2
            ar = torch.arange(K).view(1, 1, K)
            idx = torch.where(ar < cnt.unsqueeze(-1), idx, idx[:, :, :1].expand(-1, -1, K)).contiguous()
        cidx = torch.randint(0, N, (B, S), generator=g, dtype=torch.int32) if mode == 1 else None
        if mode == 0 and S > N:
            continue
        with torch.no_grad():
            ci = cidx.long() if cidx is not None else torch.arange(S).expand(B, S)
            gather = lambda t, ix: torch.gather(t, 1, ix.reshape(B, -1, 1).expand(-1, -1, t.shape[-1])).view(*ix.shape, t.shape[-1])  # noqa: E731
            rel = gather(xyz, idx.long()) - gather(xyz, ci).unsqueeze(2)
            rows = rel
            if D:
                pts = feat.permute(0, 2, 1)
                nb = gather(pts, idx.long())
                if mode == 0:
                    cf = gather(pts, ci).unsqueeze(2)
                    rows = torch.cat([rel, cf.expand(-1, -1, K, -1), nb - cf], -1)
                else:
                    rows = torch.cat([rel, nb], -1)
            x = rows.permute(0, 3, 1, 2)
            for c, b in zip(convs, bns):
                x = F.relu(b(c(x)))
            want = x.max(dim=3)[0]
        plan = engine.SaPlan(convs, bns, "cuda", mode, fast=True)
        fpm = bool(D) and rng.random() < 0.5
        opm = rng.random() < 0.5
        fg = None if feat is None else (feat.cuda().transpose(1, 2).contiguous().transpose(1, 2) if fpm else feat.cuda())
        out = plan.run(xyz.cuda(), fg, idx.cuda(), None if cidx is None else cidx.cuda(),
                       cnt=None if cnt is None else cnt.cuda(), out_point_major=opm)
        err = float((out.cpu() - want).abs().max())
        worst = max(worst, err)
        key = "mode%d%s" % (mode, "_ragged" if cnt is not None else "")
        if cnt is not None and plan.wants_row_table(N, K, 0.0):
            # the same launch fed with the ball query's row table layout (built here from idx / cnt): the same bits
            tab = np.zeros((B, (S + 15) // 16, 16 * K, 4), dtype=np.float32)
            xn, cn, ixn, cin_ = xyz.numpy(), cnt.numpy(), idx.numpy(), cidx.numpy()
            for b in range(B):
                for it in range((S + 15) // 16):
                    r = 0
                    for c in range(it * 16, min(S, it * 16 + 16)):
                        for k in range((max(int(cn[b, c]), 1) + 1) & ~1):
                            i = int(ixn[b, c, k])
                            tab[b, it, r, 0] = np.int32(i).view(np.float32)
                            tab[b, it, r, 1:] = xn[b, i] - xn[b, int(cin_[b, c])]
                            r += 1
            out2 = plan.run(xyz.cuda(), fg, None, cidx.cuda(), cnt=cnt.cuda(), out_point_major=opm,
                            rows=torch.from_numpy(tab).reshape(-1).cuda(), K=K)
            assert torch.equal(out2, out), (mode, D, K, S, N, widths, B, "row table")
            key += "_table"
        kinds[key] = kinds.get(key, 0) + 1
        assert err < 5e-5 * max(1.0, float(want.abs().max())), (mode, D, K, S, N, widths, B, cnt is not None, fpm, opm, err)
        n += 1
    print("sa fuzz ok: %d layers in %.0f s %s, worst |d| %.1e" % (n, time.time() - t0, kinds, worst))

    return n, worst

if __name__ == "__main__":
    main()
