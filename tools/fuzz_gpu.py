"""Randomised parity sweep on the GPU box (not part of the test suite): random shapes through the neighbour-search
ops and the DGCNN / local-attention kernels against the CPU oracle.  python tools/fuzz_gpu.py [seconds] [seed]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "point-cloud-reid_amd"), os.path.join(ROOT, "oracle")]
import model_oracle as MO          # noqa: E402
import point_ops as PO             # noqa: E402
from mmdet3d import ops            # noqa: E402
from mmdet3d.models.attention import local_self_attention   # noqa: E402
from pcr_amd import dgcnn_engine as DE, engine as E, testing as T   # noqa: E402


def main(budget=None, seed=None, max_cases=None):
    """budget seconds / seed from the command line when not given; max_cases bounds the sweep for the pytest slice
    (tests/test_gpu_fuzz.py: fixed seeds, fixed case counts)"""
    if budget is None:
        budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    rng = np.random.default_rng(seed if seed is not None else (int(sys.argv[2]) if len(sys.argv) > 2 else 0))
    t0, n = time.time(), 0
    counts = {}
    while time.time() - t0 < budget and (max_cases is None or n < max_cases):
        case = rng.integers(0, 5)
        kind = ["randn", "box", "dup"][rng.integers(0, 3)]
        if case == 0:      # feature kNN
            B, C, N = int(rng.integers(1, 4)), int(rng.integers(1, 130)), int(rng.integers(2, 700))
            K = int(rng.integers(1, min(N, 64) + 1))
            x = torch.from_numpy(rng.standard_normal((B, C, N)).astype(np.float32))
            if rng.random() < 0.3:      # duplicated points: exact ties
                x = x[:, :, torch.from_numpy(rng.integers(0, max(1, N // 3), N))].contiguous()
            got = DE.knn_feat(x.cuda(), K).cpu().numpy()
            want = PO.knn_feat(x.numpy(), K)
            assert (got == want).all(), ("knn_feat", B, C, N, K)
        elif case == 1:    # prefix kNN (Point-Transformer grouping)
            N = int(rng.integers(8, 1500))
            S, K = int(rng.integers(1, N + 1)), int(rng.integers(1, min(N, 64) + 1))
            xyz = T.synthetic_clouds(2, N, int(rng.integers(0, 1 << 30)), kind)
            got = E.knn_prefix(xyz.cuda(), S, K).cpu().numpy()
            assert (got == PO.knn_prefix(xyz.numpy(), S, K)).all(), ("knn_prefix", N, S, K, kind)
        elif case == 2:    # FPS + ball query
            N = int(rng.integers(8, 2000))
            M, K = int(rng.integers(1, N + 1)), int(rng.integers(1, 65))
            xyz = T.synthetic_clouds(2, N, int(rng.integers(0, 1 << 30)), kind)
            idx = ops.furthest_point_sample(xyz.cuda(), M)
            assert (idx.cpu().numpy() == PO.fps(xyz.numpy(), M)).all(), ("fps", N, M, kind)
            centres = torch.gather(xyz, 1, idx.cpu().long().unsqueeze(-1).expand(-1, -1, 3)).contiguous()
            r = float(rng.uniform(0.05, 1.0))
            bq = ops.ball_query(0.0, r, K, xyz.cuda(), centres.cuda())
            assert (bq.cpu().numpy() == PO.ball_query(0.0, r, K, xyz.numpy(), centres.numpy())).all(), ("bq", N, M, K)
        elif case == 3:    # local self attention module
            N = int(rng.integers(8, 300))
            knum = int(rng.integers(1, min(N, 64) + 1))
            m = local_self_attention(64, 2, knum=knum, pos_size=64)
            sd = T.seeded_state_dict(T.manifest_of(m), int(rng.integers(0, 1000)))
            m.load_state_dict(sd)
            feat = torch.from_numpy(rng.standard_normal((2, 64, N)).astype(np.float32))
            xyz = torch.from_numpy(rng.standard_normal((2, N, 3)).astype(np.float32))
            with torch.no_grad():
                want = MO.local_self_attention(sd, feat, xyz, 2, knum)
                got = m.cuda().eval()(feat.cuda(), xyz.cuda()).cpu()
            assert float((got - want).abs().max()) < 1e-4, ("local_attn", N, knum, float((got - want).abs().max()))
        else:              # dense (all launch shapes incl. chunked) vs torch
            cin = int(rng.choice([3, 7, 64, 100, 128, 256, 512, 640, 768, 1024]))
            cout = int(rng.choice([9, 64, 128, 200, 256, 512, 1024]))
            L, B = int(rng.integers(1, 300)), int(rng.integers(1, 4))
            x = torch.from_numpy(rng.standard_normal((B, cin, L)).astype(np.float32))
            w = torch.from_numpy((rng.standard_normal((cout, cin)) / np.sqrt(cin)).astype(np.float32))
            sh = torch.from_numpy(rng.standard_normal(cout).astype(np.float32))
            want = torch.einsum("oc,bcl->bol", w, x) + sh.view(1, -1, 1)
            got = E.dense(x.cuda(), E.pack_weight(w, "cuda"), cout, None, sh.cuda(), act=0).cpu()
            assert float((got - want).abs().max()) < 3e-4 * max(1.0, float(want.abs().max())), ("dense", cin, cout, L)
        counts[int(case)] = counts.get(int(case), 0) + 1
        n += 1
    print("fuzz ok: %d cases in %.0f s, by kind %s" % (n, time.time() - t0, counts))

    return n, counts

if __name__ == "__main__":
    main()
