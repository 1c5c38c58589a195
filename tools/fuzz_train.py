"""Randomised parity sweep of the TRAINING launches on the GPU box (not part of the test suite): random shapes through the
train-dense forward / backward, the token norm, the linear-attention core, the pair pooling and the grouped edge MLP
against torch autograd on the same formulas.  python tools/fuzz_train.py [seconds] [seed]"""
import copy
import os
import sys
import time

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "point-cloud-reid_amd"), os.path.join(ROOT, "tests")]
from pcr_amd import engine, testing as T, train_ops as TO      # noqa: E402


def _rel(a, b):
    return float((a - b).abs().max()) / max(1e-6, float(b.abs().max()))


def _cmp(ours, want, gated):
    """forward (first entry) always tight; gradients tight too unless a ReLU gates them: an activation within rounding
    of zero is then gated differently by the two summation orders (deterministic for a given input, ~1 case in 50),
    which moves a few entries by O(1e-2) -- at most 1 % of a tensor's entries may be off by more than 2e-2 of its scale
    (a real defect moves all of them)"""
    worst = _rel(ours[0].cpu(), want[0].cpu())
    for a, b in zip(ours[1:], want[1:]):
        a, b = a.cpu(), b.cpu()
        if not gated:
            worst = max(worst, _rel(a, b))
        else:
            scale = max(1e-6, float(b.abs().max()))
            frac = float(((a - b).abs() > 2e-2 * scale).float().mean())
            worst = max(worst, 0.0 if frac <= 0.01 or ((a - b).abs() > 2e-2 * scale).sum() <= 2 else 1.0)
    return worst


def _grads(fn, tensors, go):
    for t in tensors:
        t.grad = None
    y = fn()
    (y * go).sum().backward()
    return [y.detach()] + [t.grad.clone() for t in tensors]


def case_dense(rng, g):
    B = int(rng.integers(1, 7))
    cin = int(rng.choice([3, 8, 32, 64, 67, 96, 128, 131, 200, 256]))
    cin2 = int(rng.choice([0, 0, 32, 64])) if cin <= 128 else 0
    cout = int(rng.choice([1, 32, 64, 96, 128, 192, 256, 300, 384]))
    Ln = int(rng.choice([1, 4, 20, 32, 33, 64, 96, 100, 128, 200, 257, 512]))
    if rng.integers(0, 6) == 0:       # many short clouds: a workgroup then strides over several clouds (carried dW / stats)
        B, Ln = int(rng.choice([700, 900, 1700])), int(rng.choice([16, 32, 50, 64, 128]))
        cin, cin2, cout = min(cin, 128), 0, min(cout, 128)
    relu = bool(rng.integers(0, 2))
    has_b = bool(rng.integers(0, 2))
    c32 = lambda v: (v + 31) // 32 * 32      # noqa: E731
    if (max(c32(cout), c32(cin + cin2)) + c32(cin + cin2)) * 65 * 4 > 150 * 1024:
        cout = 128                           # (the backward keeps dy and f(x) tiles in LDS: documented limit of the launch)
    x = torch.randn(B, cin, Ln, generator=g).cuda().requires_grad_(True)
    x2 = torch.randn(B, cin2, Ln, generator=g).cuda().requires_grad_(True) if cin2 else None
    W = (torch.randn(cout, cin + cin2, generator=g) / (cin + cin2) ** 0.5).cuda().requires_grad_(True)
    b = torch.randn(cout, generator=g).cuda().requires_grad_(True) if has_b else None
    res = None if relu or rng.integers(0, 2) else torch.randn(B, cout, Ln, generator=g).cuda().requires_grad_(True)
    go = torch.randn(B, cout, Ln, generator=g).cuda()
    ts = [t for t in (x, x2, W, b, res) if t is not None]

    def torch_fn():
        xin = x if x2 is None else torch.cat([x, x2], dim=1)
        y = torch.einsum("oc,bcl->bol", W, xin)
        if b is not None:
            y = y + b.view(1, -1, 1)
        if res is not None:
            y = y + res
        return F.relu(y) if relu else y
    ours = _grads(lambda: TO.dense(x, W, b, x2=x2, res=res, relu=relu), ts, go)
    want = _grads(torch_fn, ts, go)
    return _cmp(ours, want, relu), (B, cin, cin2, cout, Ln, relu, has_b, res is not None)


def case_tnorm(rng, g):
    B, Ln = int(rng.integers(1, 5)), int(rng.choice([1, 7, 32, 33, 64, 100, 128, 300]))
    if rng.integers(0, 8) == 0:
        B = int(rng.choice([300, 700]))
    C, G = [(32, 1), (64, 1), (128, 1), (256, 1), (96, 1), (128, 32), (256, 32), (96, 6), (40, 1)][int(rng.integers(0, 9))]   # (groups of 2 are degenerate: dx cancels to ~0)
    res, relu = bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
    x = torch.randn(B, C, Ln, generator=g).cuda().requires_grad_(True)
    r = torch.randn(B, C, Ln, generator=g).cuda().requires_grad_(True) if res else None
    norm = (torch.nn.LayerNorm(C) if G == 1 else torch.nn.GroupNorm(G, C)).cuda()
    with torch.no_grad():
        norm.weight.copy_(1 + 0.1 * torch.randn(C, generator=g))
        norm.bias.copy_(0.1 * torch.randn(C, generator=g))
    go = torch.randn(B, C, Ln, generator=g).cuda()
    ts = [x, norm.weight, norm.bias] + ([r] if res else [])
    xc = x.detach().cpu().requires_grad_(True)       # (CPU reference: torch's GPU GroupNorm backward is unreliable here)
    rc = r.detach().cpu().requires_grad_(True) if res else None
    normc = copy.deepcopy(norm).cpu()
    tc = [xc, normc.weight, normc.bias] + ([rc] if res else [])

    def torch_fn():
        rows = xc.permute(0, 2, 1).reshape(B * Ln, C)
        y = normc(rows).reshape(B, Ln, C).permute(0, 2, 1)
        if res:
            y = y + rc
        return F.relu(y) if relu else y
    ours = _grads(lambda: TO.tnorm(x, norm, res=r, relu=relu), ts, go)
    want = _grads(torch_fn, tc, go.cpu())
    return _cmp(ours, want, relu), (B, C, Ln, G, res, relu)


def case_linattn(rng, g):
    from test_gpu_train_ops import _t_linattn
    B = int(rng.integers(1, 5))
    d, H = [(32, 2), (64, 2), (128, 2), (64, 4), (64, 1), (128, 4)][int(rng.integers(0, 6))]
    Lq = int(rng.choice([2, 16, 32, 37, 64, 96, 100, 128, 200]))
    fused = bool(rng.integers(0, 2))
    Sk = Lq if fused else int(rng.choice([2, 16, 32, 45, 64, 70, 128, 150]))   # (one key: dq = 0 in exact arithmetic)
    go = torch.randn(B, d, Lq, generator=g).cuda()
    if fused:
        qkv = torch.randn(B, 3 * d, Lq, generator=g).cuda().requires_grad_(True)
        ours = _grads(lambda: TO.LinAttnQKV.apply(qkv, H, 1e-6), [qkv], go)
        want = _grads(lambda: _t_linattn(qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:], H), [qkv], go)
    else:
        q = torch.randn(B, d, Lq, generator=g).cuda().requires_grad_(True)
        k = torch.randn(B, d, Sk, generator=g).cuda().requires_grad_(True)
        v = torch.randn(B, d, Sk, generator=g).cuda().requires_grad_(True)
        ours = _grads(lambda: TO.LinAttn.apply(q, k, v, H, 1e-6), [q, k, v], go)
        want = _grads(lambda: _t_linattn(q, k, v, H), [q, k, v], go)
    return max(_rel(a, r) for a, r in zip(ours, want)), (B, d, H, Lq, Sk, fused)


def case_pool(rng, g):
    P, C, Ln = int(rng.integers(1, 6)), int(rng.choice([32, 64, 100, 128])), int(rng.choice([1, 31, 64, 77, 128]))
    o = torch.randn(2 * P, C, Ln, generator=g).cuda().requires_grad_(True)
    go = torch.randn(P, 2 * C, generator=g).cuda()

    def torch_fn():
        x = torch.cat([o[:P], o[P:]], dim=2)
        return torch.cat([x.max(dim=2)[0], x.mean(dim=2)], dim=1)
    ours = _grads(lambda: TO.PoolPair.apply(o), [o], go)
    want = _grads(torch_fn, [o], go)
    return max(_rel(a, r) for a, r in zip(ours, want)), (P, C, Ln)


def case_sa(rng, g):
    from mmdet3d.models.pointnet2_utils import PointNetSetAbstractionEdgeSA
    from test_gpu_train_ops import _torch_sa
    B = int(rng.integers(2, 5))
    if rng.integers(0, 8) == 0:       # more clouds than workgroup rows
        B = int(rng.choice([200, 400]))
    N = int(rng.choice([48, 64, 100, 128, 150, 256]))
    if B > 100:
        N = int(rng.choice([48, 64]))
    S = int(rng.integers(8, N + 1))
    K = int(rng.choice([8, 16, 20, 32, 48]))
    K = min(K, N)
    D = int(rng.choice([0, 8, 16, 32, 64]))
    widths = [(32, 32, 32), (64, 64, 64), (128, 128, 128), (32, 64, 96), (64, 64, 128), (24, 40, 72)][int(rng.integers(0, 6))]
    sa = PointNetSetAbstractionEdgeSA(npoint=None, radius=0.3, nsample=K, mlp=[2 * D] + list(widths),
                                      sampling="RANDOM", use_xyz=True, use_knn=True)
    wseed, cseed, kind = int(rng.integers(0, 1000)), int(rng.integers(0, 1000)), ["randn", "box"][int(rng.integers(0, 2))]
    sa.load_state_dict(T.seeded_state_dict(T.manifest_of(sa), wseed))
    sa = sa.cuda().train()
    ref = copy.deepcopy(sa)
    xyz = T.synthetic_clouds(B, N, seed=cseed, kind=kind).cuda()
    if os.environ.get("PCR_FUZZ_DUMP"):
        torch.save(dict(g_state=g.get_state(), wseed=wseed, cseed=cseed, kind=kind, dims=(B, N, S, K, D, widths)),
                   os.environ["PCR_FUZZ_DUMP"])
    feats = torch.randn(B, D, N, generator=g).cuda().requires_grad_(True) if D else None
    feats_r = feats.detach().clone().requires_grad_(True) if D else None
    idx = engine.knn_prefix(xyz, S, K)
    out = TO.sa_edge_train(sa, xyz, feats, idx)
    want = _torch_sa(ref, xyz, feats_r, idx)
    worst = _rel(out, want)                  # forward: tight
    w = torch.randn(out.shape, generator=g).cuda()
    (out * w).sum().backward()
    (want * w).sum().backward()

    def bad(a, b):
        # the max over K routes a whole gradient row to ONE neighbour, and every ReLU gates on the sign of an
        # activation: a near-tie / an activation within rounding of zero resolved differently by the two summation
        # orders moves that row, and through the BatchNorm sums (only B S K ~ 10^3 rows here) every entry of the
        # channel by ~1e-3 (seen at ~1 case in 50, deterministic for a given input).  Gradients are therefore compared
        # element-wise: at most 1 % of the entries may be off by more than 2e-2 of the tensor's scale (a real bug moves
        # all of them by O(1)); the dedicated test pins fixed shapes to 2e-4
        scale = max(1e-6, float(b.abs().max()))
        frac = float(((a - b).abs() > 2e-2 * scale).float().mean())
        return 0.0 if frac <= 0.01 else 1.0
    detail = {"out": worst}
    if D:
        detail["feats"] = _rel(feats.grad, feats_r.grad)
        worst = max(worst, bad(feats.grad, feats_r.grad))
    for (k, p), (_, q) in zip(sa.named_parameters(), ref.named_parameters()):
        if k.startswith("mlp_") and q.grad is not None:
            if "convs" in k and k.endswith("bias"):
                continue        # zero true gradient in front of BatchNorm: rounding noise only
            detail[k] = _rel(p.grad, q.grad)
            # (parameter gradients are sums over ~10^3 rows here: one re-routed / re-gated row shifts them by up to ~1e-1 of
            # their scale -- observed 8e-2 with every activation within 7e-7 of zero accounted for)
            worst = max(worst, 0.0 if detail[k] < 0.2 else 1.0)
    if worst >= 5e-5:
        print("sa detail", {k: "%.1e" % v for k, v in detail.items()})
    return worst, (B, N, S, K, D, widths, wseed, cseed, kind)


def main(budget=None, seed=None, max_cases=None):
    if budget is None:
        budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    rng = np.random.default_rng(seed if seed is not None else (int(sys.argv[2]) if len(sys.argv) > 2 else 0))
    g = torch.Generator().manual_seed(int(rng.integers(0, 1 << 30)))
    cases = [("dense", case_dense, 5e-5), ("tnorm", case_tnorm, 5e-5), ("linattn", case_linattn, 1e-4),
             ("pool", case_pool, 1e-5), ("sa", case_sa, 5e-5)]
    t0, n, kinds, worst = time.time(), 0, {}, {}
    while time.time() - t0 < budget and (max_cases is None or n < max_cases):
        name, fn, tol = cases[int(rng.integers(0, len(cases)))]
        st_rng, st_g = rng.bit_generator.state, g.get_state()
        try:
            err, shape = fn(rng, g)
        except Exception:
            print("case", name, "raised")
            raise
        kinds[name] = kinds.get(name, 0) + 1
        worst[name] = max(worst.get(name, 0.0), err)
        if not err < tol:
            # replay the same case in isolation (fresh host-side caches): a pass here means state leaked between cases
            TO._PAD_CACHE.clear()
            torch.cuda.synchronize()
            torch.cuda.empty_cache()
            rng.bit_generator.state = st_rng
            g.set_state(st_g)
            err2, _ = fn(rng, g)
            print("replay in isolation: %.3e" % err2)
            raise SystemExit("MISMATCH %s %s: %.3e (tolerance %.1e)" % (name, shape, err, tol))
        n += 1
    print("train fuzz ok: %d cases in %.0f s %s, worst %s" % (n, time.time() - t0, kinds,
                                                               {k: "%.1e" % v for k, v in worst.items()}))
    return n


if __name__ == "__main__":
    main()
