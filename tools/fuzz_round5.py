"""Randomised parity sweep of the kernels round 5 touched last (not part of the test suite), on the GPU box:
python tools/fuzz_round5.py [seconds] [seed]
  * pcr_knn_prefix_f32 up to 4096 points against the C oracle (register and LDS kernels, lane-local compaction);
  * pcr_knn_prefix2_f32 against two single searches, entry for entry;
  * pcr_fps_ball_query_rows_f32 against the separate sampling / gather / row-table launches;
  * pcr_dense_pm_prec_f32: whole-tile batches (persistent kernel) against the same rows in a cut batch (tile kernel)."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "point-cloud-reid_amd"), os.path.join(ROOT, "oracle")]
import point_ops as PO             # noqa: E402
from mmdet3d import ops            # noqa: E402
from mmdet3d.ops import point_ops as GPO   # noqa: E402
from pcr_amd import engine as E, testing as T, _lib as L   # noqa: E402


def _cloud(rng, B, N, kind):
    if kind == "lattice":
        return torch.from_numpy(rng.integers(0, int(rng.integers(3, 12)), (B, N, 3)).astype(np.float32) * np.float32(0.25))
    return T.synthetic_clouds(B, N, int(rng.integers(0, 1 << 30)), kind)


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    t0, counts = time.time(), {}
    while time.time() - t0 < budget:
        case = int(rng.integers(0, 4))
        kind = ["randn", "box", "dup", "lattice"][rng.integers(0, 4)]
        if case == 0:
            N = int(rng.choice([int(rng.integers(513, 1025)), int(rng.integers(1025, 2049)), int(rng.integers(2049, 4097))]))
            S, K = int(rng.integers(1, min(N, 300) + 1)), int(rng.integers(1, 65))
            xyz = _cloud(rng, 2, N, kind)
            got = E.knn_prefix(xyz.cuda(), S, K).cpu().numpy()
            assert (got == PO.knn_prefix(xyz.numpy(), S, K)).all(), ("knn_prefix", N, S, K, kind)
        elif case == 1:
            N = int(rng.integers(8, 4097))
            S = int(rng.integers(1, N + 1))
            S2 = int(rng.integers(0, S + 1))
            K = int(rng.integers(1, min(N, 64) + 1))
            K2 = int(rng.integers(K, min(N, 64) + 1))
            xyz = _cloud(rng, int(rng.integers(1, 4)), N, kind).cuda()
            a, b = E.knn_prefix2(xyz, S, K, S2, K2)
            assert torch.equal(a, E.knn_prefix(xyz, S, K)), ("knn2 a", N, S, K, S2, K2, kind)
            if S2:
                assert torch.equal(b, E.knn_prefix(xyz, S2, K2)), ("knn2 b", N, S, K, S2, K2, kind)
        elif case == 2:
            N = int(rng.integers(4, 1025))
            M = int(rng.integers(2, N + 1))
            K = 2 * int(rng.integers(1, 33))
            r = float(rng.uniform(0.05, 1.0))
            if not GPO.fps_ball_query_rows_ok(N, M, K):
                continue
            B = int(rng.integers(1, 6))
            xyz = _cloud(rng, B, N, kind).cuda().contiguous()
            idx, new_xyz, cnt, rows = GPO.fps_ball_query_rows(xyz, M, r, K)
            want_idx = ops.furthest_point_sample(xyz, M)
            assert torch.equal(idx, want_idx), ("fps_bq idx", N, M, K, kind)
            want_xyz = ops.gather_points(xyz.transpose(1, 2).contiguous(), want_idx).transpose(1, 2).contiguous()
            assert torch.equal(new_xyz, want_xyz), ("fps_bq xyz", N, M, K, kind)
            _, want_cnt, want_rows = GPO.ball_query_rows(r, K, xyz, want_xyz)
            assert torch.equal(cnt, want_cnt), ("fps_bq cnt", N, M, K, kind)
            c = np.maximum(cnt.cpu().numpy(), 1)
            c = (c + 1) // 2 * 2
            nit = (M + 15) // 16
            pad = np.zeros((B, nit * 16), dtype=np.int64)
            pad[:, :M] = c
            used = (pad.reshape(B, nit, 16).sum(axis=2) + 31) // 32 * 32
            a = rows.view(B, nit, 16 * K, 4).cpu().numpy().view(np.uint32)
            w = want_rows.view(B, nit, 16 * K, 4).cpu().numpy().view(np.uint32)
            for bi in range(B):
                for it in range(nit):
                    u = int(used[bi, it])
                    assert np.array_equal(a[bi, it, :u], w[bi, it, :u]), ("fps_bq rows", N, M, K, kind, bi, it)
        else:
            cin, cout = int(rng.choice([32, 64, 128])), int(rng.choice([64, 128]))
            Lr = 8 * int(rng.integers(2, 40))
            B = 8 * int(rng.integers(1, 5))          # B L a multiple of 64
            prec = ["bf16x3", "bf16"][rng.integers(0, 2)]
            x = torch.from_numpy(rng.standard_normal((B, Lr, cin)).astype(np.float32)).cuda()
            w = torch.from_numpy(rng.standard_normal((cout, cin)).astype(np.float32)) / cin ** 0.5
            wp = E.pack_weight_bf(w, torch.device("cuda"))
            lib = L.load()

            def table(xs):
                b, l, _ = xs.shape
                y = torch.full((b, l, cout), float("nan"), device="cuda")
                L.check(lib.pcr_dense_pm_prec_f32(L.ptr(xs), L.ptr(wp), L.ptr(y), b, cin, cout, l, 1, E.PRECISIONS[prec],
                                                  L.stream_ptr()), "pcr_dense_pm_prec_f32")
                return y
            full = table(x)
            cut = Lr - 1
            assert (B * cut) % 64 != 0
            assert torch.equal(full[:, :cut], table(x[:, :cut].contiguous())), ("dense_pm", B, Lr, cin, cout, prec)
        counts[case] = counts.get(case, 0) + 1
    print("fuzz_round5 ok:", counts, "in %.0f s" % (time.time() - t0))


if __name__ == "__main__":
    main()
