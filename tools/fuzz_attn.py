"""Randomised sweep of the linear-attention kernels on the GPU box (not part of the test suite): Self_Attention
(d = 32/64/128), FP_SA (the backbone's three (C1,C2,d,out) shapes and others) and corss_attention at random token
counts, against the torch-eager oracle.  python tools/fuzz_attn.py [seconds] [seed]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "point-cloud-reid_amd"), os.path.join(ROOT, "oracle")]
import model_oracle as MO          # noqa: E402
from mmdet3d.models.attention import corss_attention                      # noqa: E402
from mmdet3d.models.pointnet2_utils import FP_SA, Self_Attention         # noqa: E402
from pcr_amd import testing as T   # noqa: E402


def main(budget=None, seed=None, max_cases=None):
    """budget seconds / seed from the command line when not given; max_cases bounds the sweep for the pytest slice
    (tests/test_gpu_fuzz.py: fixed seeds, fixed case counts)"""
    if budget is None:
        budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    rng = np.random.default_rng(seed if seed is not None else (int(sys.argv[2]) if len(sys.argv) > 2 else 0))
    t0, n, worst, kinds = time.time(), 0, 0.0, {}
    tt = lambda *s: torch.from_numpy(rng.standard_normal(s).astype(np.float32))     # noqa: E731
    while time.time() - t0 < budget and (max_cases is None or n < max_cases):
        case = ["self", "fp", "cross"][rng.integers(0, 3)]
        B = int(rng.integers(1, 5))
        Lq, Sk = int(rng.integers(1, 700)), int(rng.integers(1, 700))
        seed = int(rng.integers(0, 1000))
        if case == "self":
            d = int(rng.choice([32, 64, 128, 256, 512]))
            if d > 128:
                Lq = min(Lq, 200)
            m = Self_Attention(d, 2)
            sd = T.seeded_state_dict(T.manifest_of(m), seed)
            m.load_state_dict(sd)
            feat, xyz = tt(B, d, Lq), tt(B, Lq, 3)
            with torch.no_grad():
                want = MO.self_attention(sd, feat, xyz)
                got = m.cuda().eval()(feat.cuda(), xyz.cuda()).cpu()
        elif case == "fp":
            c1, c2, d, out = [(64, 128, 64, 128), (32, 128, 64, 64), (3, 64, 64, 32), (16, 64, 64, 64),
                              (64, 64, 32, 32), (128, 128, 128, 64), (128, 256, 128, 256), (64, 256, 128, 128),
                              (3, 128, 128, 64), (256, 512, 256, 512), (128, 512, 256, 256), (3, 256, 256, 128)][rng.integers(0, 12)]
            if d > 128:
                Lq, Sk = min(Lq, 200), min(Sk, 200)
            m = FP_SA(0, c1, c2, d, out, 2)
            sd = T.seeded_state_dict(T.manifest_of(m), seed)
            m.load_state_dict(sd)
            f1, x1, f2, x2 = tt(B, c1, Lq), tt(B, Lq, 3), tt(B, c2, Sk), tt(B, Sk, 3)
            with torch.no_grad():
                want = MO.fp_sa(sd, f1, x1, f2, x2)
                got = m.cuda().eval()(f1.cuda(), x1.cuda(), f2.cuda(), x2.cuda()).cpu()
        else:
            d = int(rng.choice([32, 64, 128]))
            m = corss_attention(d, 2)
            sd = T.seeded_state_dict(T.manifest_of(m), seed)
            m.load_state_dict(sd)
            s, sx, t, tx = tt(B, d, Lq), tt(B, Lq, 3), tt(B, d, Sk), tt(B, Sk, 3)
            with torch.no_grad():
                want = MO.cross_attention(sd, s, sx, t, tx)
                got = m.cuda().eval()(s.cuda(), sx.cuda(), t.cuda(), tx.cuda()).cpu()
        err = float((got - want).abs().max())
        worst = max(worst, err)
        kinds[case] = kinds.get(case, 0) + 1
        assert err < 1e-4, (case, B, Lq, Sk, seed, err)
        n += 1
    print("attention fuzz ok: %d blocks in %.0f s %s, worst |d| %.1e" % (n, time.time() - t0, kinds, worst))

    return n, worst

if __name__ == "__main__":
    main()
