"""GPU: the training-mode HIP kernels (include/pcr.h section C) against plain torch autograd on the same graph --
forward values and every gradient -- plus bit-reproducibility of the gradients (no float atomics anywhere)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from pcr_amd import testing as T

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return float((a - b).abs().max()) / max(1e-6, float(b.abs().max()))


def _torch_sa(sa, xyz, feats, idx):
    """the reference graph of the grouped MLP (pointnet2_utils.py:242-288, 333-357) in plain torch"""
    B, S, K = idx.shape
    li = idx.long()
    g = torch.gather(xyz.unsqueeze(1).expand(-1, S, -1, -1), 2, li.unsqueeze(-1).expand(-1, -1, -1, 3))
    x = (g - xyz[:, :S].unsqueeze(2)).permute(0, 3, 1, 2)
    if feats is not None:
        pts = feats.permute(0, 2, 1)
        centre = pts[:, :S].unsqueeze(2)
        nb = torch.gather(pts.unsqueeze(1).expand(-1, S, -1, -1), 2, li.unsqueeze(-1).expand(-1, -1, -1, pts.shape[-1]))
        x = torch.cat([x, centre.expand(-1, -1, K, -1).permute(0, 3, 1, 2), (nb - centre).permute(0, 3, 1, 2)], dim=1)
    for conv, bn in zip(sa.mlp_convs, sa.mlp_bns):
        x = F.relu(bn(conv(x)))
    return x.max(dim=3)[0]


@pytest.mark.parametrize("B,N,S,K,D,widths", [(3, 128, 128, 32, 0, (32, 32, 32)), (4, 128, 64, 48, 32, (64, 64, 64)),
                                              (2, 64, 32, 48, 64, (128, 128, 128)), (3, 100, 37, 20, 16, (32, 64, 96)),
                                              (2, 300, 150, 16, 8, (64, 64, 128))])
def test_sa_edge_train_matches_torch_autograd(B, N, S, K, D, widths):
    import copy
    from mmdet3d.models.pointnet2_utils import PointNetSetAbstractionEdgeSA
    from pcr_amd import engine, train_ops
    sa = PointNetSetAbstractionEdgeSA(npoint=None, radius=0.3, nsample=K, mlp=[2 * D] + list(widths),
                                      sampling="RANDOM", use_xyz=True, use_knn=True)
    sa.load_state_dict(T.seeded_state_dict(T.manifest_of(sa), 5))
    sa = sa.cuda().train()
    ref = copy.deepcopy(sa)
    xyz = T.synthetic_clouds(B, N, seed=3, kind="randn").cuda()
    g = torch.Generator().manual_seed(1)
    feats = torch.randn(B, D, N, generator=g).cuda().requires_grad_(True) if D else None
    feats_r = feats.detach().clone().requires_grad_(True) if D else None
    idx = engine.knn_prefix(xyz, S, K)
    out = train_ops.sa_edge_train(sa, xyz, feats, idx)
    want = _torch_sa(ref, xyz, feats_r, idx)
    assert _rel(out, want) < 2e-5, _rel(out, want)
    w = torch.randn(out.shape, generator=g).cuda()
    (out * w).sum().backward()
    (want * w).sum().backward()
    worst = {}
    if D:
        worst["feats"] = _rel(feats.grad, feats_r.grad)
    for (k, p), (_, q) in zip(sa.named_parameters(), ref.named_parameters()):
        if k.startswith("mlp_") and q.grad is not None:
            if "convs" in k and k.endswith("bias"):
                # a conv bias in front of BatchNorm has a zero gradient up to rounding: absolute comparison
                assert float((p.grad - q.grad).abs().max()) < 1e-4 * max(1.0, float(w.abs().sum()) ** 0.5), k
                continue
            worst[k] = _rel(p.grad, q.grad)
    print(worst)
    assert max(worst.values()) < 2e-4, worst
    for (k, p), (_, q) in zip(sa.named_buffers(), ref.named_buffers()):
        if k.startswith("mlp_bns"):
            assert _rel(p.float(), q.float()) < 1e-5, k      # running statistics, num_batches_tracked


def test_train_dense_and_gradients_are_reproducible():
    from pcr_amd import train_ops
    g = torch.Generator().manual_seed(2)
    for B, cin, cin2, cout, Ln, relu in ((3, 64, 0, 128, 200, False), (2, 3, 64, 128, 77, True), (4, 128, 128, 256, 64, True),
                                         (1, 131, 0, 32, 1000, False)):
        x = torch.randn(B, cin, Ln, generator=g).cuda().requires_grad_(True)
        x2 = torch.randn(B, cin2, Ln, generator=g).cuda().requires_grad_(True) if cin2 else None
        W = (torch.randn(cout, cin + cin2, generator=g) / (cin + cin2) ** 0.5).cuda().requires_grad_(True)
        b = torch.randn(cout, generator=g).cuda().requires_grad_(True)
        res = None if relu else torch.randn(B, cout, Ln, generator=g).cuda().requires_grad_(True)
        go = torch.randn(B, cout, Ln, generator=g).cuda()

        def run(fn):
            for t in (x, x2, W, b, res):
                if t is not None:
                    t.grad = None
            y = fn()
            (y * go).sum().backward()
            return [y.detach()] + [None if t is None else t.grad.clone() for t in (x, x2, W, b, res)]
        ours = run(lambda: train_ops.dense(x, W, b, x2=x2, res=res, relu=relu))
        again = run(lambda: train_ops.dense(x, W, b, x2=x2, res=res, relu=relu))

        def torch_fn():
            xin = x if x2 is None else torch.cat([x, x2], dim=1)
            y = torch.einsum("oc,bcl->bol", W, xin) + b.view(1, -1, 1)
            if res is not None:
                y = y + res
            return F.relu(y) if relu else y
        want = run(torch_fn)
        for a, r, name in zip(ours, want, ("y", "dx", "dx2", "dW", "db", "dres")):
            if a is not None:
                assert _rel(a, r) < 2e-5, (name, _rel(a, r), B, cin, cin2, cout, Ln)
        for a, r in zip(ours, again):
            if a is not None:
                assert torch.equal(a, r)          # fixed-order reductions: bit-identical from run to run
