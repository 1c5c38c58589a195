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


def test_split_bf16_backward_of_the_128_wide_layers_stays_on_the_f32_gradients():
    """TRAIN_PRECISION defaults to bf16x3 (dx and dW of the 128 x 128 grouped-MLP backward on the bf16 matrix
    core, three MFMAs per product); set_train_precision('f32') restores the reference's arithmetic.  Both ways on the same
    inputs: every gradient within 2e-5 of its tensor's scale, and the modes really differ (the switch is live)."""
    import copy
    from mmdet3d.models.pointnet2_utils import PointNetSetAbstractionEdgeSA
    from pcr_amd import engine, train_ops
    B, N, S, K, D, widths = 2, 64, 32, 48, 64, (128, 128, 128)
    sa = PointNetSetAbstractionEdgeSA(npoint=None, radius=0.3, nsample=K, mlp=[2 * D] + list(widths),
                                      sampling="RANDOM", use_xyz=True, use_knn=True)
    sa.load_state_dict(T.seeded_state_dict(T.manifest_of(sa), 5))
    xyz = T.synthetic_clouds(B, N, seed=3, kind="randn").cuda()
    g = torch.Generator().manual_seed(1)
    feats0 = torch.randn(B, D, N, generator=g).cuda()
    idx = engine.knn_prefix(xyz, S, K)
    w = torch.randn(B, widths[2], S, generator=g).cuda()
    grads = {}
    prev = train_ops.TRAIN_PRECISION
    try:
        for mode in ("f32", "bf16x3"):
            train_ops.set_train_precision(mode)
            m = copy.deepcopy(sa).cuda().train()
            f = feats0.clone().requires_grad_(True)
            (train_ops.sa_edge_train(m, xyz, f, idx) * w).sum().backward()
            grads[mode] = dict({k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None}, feats=f.grad.clone())
    finally:
        train_ops.set_train_precision(prev)
    worst, differ = 0.0, False
    for k, a in grads["f32"].items():
        b = grads["bf16x3"][k]
        if "convs" in k and k.endswith("bias"):      # (zero up to rounding in front of a BatchNorm: no scale to compare on)
            continue
        worst = max(worst, _rel(b, a))
        differ = differ or not torch.equal(a, b)
    assert worst < 2e-5, worst
    assert differ


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


# ---- attention / norm / pooling Functions against torch autograd on the reference's formulas ----
def _t_linattn(q, k, v, H, eps=1e-6):
    """LinearAttention.forward (pointnet2_utils.py:26-47) on (B,d,L) tensors"""
    B, d, Lq = q.shape
    qq = q.permute(0, 2, 1).reshape(B, Lq, H, d // H)
    kk = k.permute(0, 2, 1).reshape(B, -1, H, d // H)
    vv = v.permute(0, 2, 1).reshape(B, -1, H, d // H)
    Q, K = F.elu(qq) + 1, F.elu(kk) + 1
    s = vv.size(1)
    vv = vv / s
    KV = torch.einsum("nshd,nshv->nhdv", K, vv)
    Z = 1 / (torch.einsum("nlhd,nhd->nlh", Q, K.sum(dim=1)) + eps)
    out = torch.einsum("nlhd,nhdv,nlh->nlhv", Q, KV, Z) * s
    return out.reshape(B, Lq, d).permute(0, 2, 1)


def _grads(fn, tensors, go):
    for t in tensors:
        t.grad = None
    y = fn()
    (y * go).sum().backward()
    return [y.detach()] + [t.grad.clone() for t in tensors]


@pytest.mark.parametrize("B,d,H,Lq,Sk,fused", [(3, 32, 2, 128, 128, True), (2, 64, 2, 100, 37, False), (4, 128, 2, 32, 32, True),
                                               (2, 64, 2, 200, 64, False), (2, 128, 2, 96, 96, True),
                                               (2, 128, 2, 70, 45, False)])
def test_linear_attention_core_matches_torch(B, d, H, Lq, Sk, fused):
    from pcr_amd import train_ops as TO
    g = torch.Generator().manual_seed(4)
    if fused:
        qkv = torch.randn(B, 3 * d, Lq, generator=g).cuda().requires_grad_(True)
        go = torch.randn(B, d, Lq, generator=g).cuda()
        ours = _grads(lambda: TO.LinAttn.apply(qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:], H, 1e-6), [qkv], go)
        want = _grads(lambda: _t_linattn(qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:], H), [qkv], go)
        one = _grads(lambda: TO.LinAttnQKV.apply(qkv, H, 1e-6), [qkv], go)       # the one-input form of the training graph
        assert all(torch.equal(a, b) for a, b in zip(one, ours))
    else:
        q = torch.randn(B, d, Lq, generator=g).cuda().requires_grad_(True)
        k = torch.randn(B, d, Sk, generator=g).cuda().requires_grad_(True)
        v = torch.randn(B, d, Sk, generator=g).cuda().requires_grad_(True)
        go = torch.randn(B, d, Lq, generator=g).cuda()
        ours = _grads(lambda: TO.LinAttn.apply(q, k, v, H, 1e-6), [q, k, v], go)
        want = _grads(lambda: _t_linattn(q, k, v, H), [q, k, v], go)
    for a, r in zip(ours, want):
        assert _rel(a, r) < 2e-5, _rel(a, r)


@pytest.mark.parametrize("B,C,Ln,G,res,relu", [(3, 64, 128, 1, True, False), (2, 128, 33, 1, False, False),
                                               (1, 128, 256, 8, True, True), (1, 96, 70, 6, False, True)])
def test_token_norm_matches_torch(B, C, Ln, G, res, relu):
    from pcr_amd import train_ops as TO
    g = torch.Generator().manual_seed(5)
    x = torch.randn(B, C, Ln, generator=g).cuda().requires_grad_(True)
    r = torch.randn(B, C, Ln, generator=g).cuda().requires_grad_(True) if res else None
    norm = (torch.nn.LayerNorm(C) if G == 1 else torch.nn.GroupNorm(G, C)).cuda()
    with torch.no_grad():
        norm.weight.copy_(1 + 0.1 * torch.randn(C, generator=g))
        norm.bias.copy_(0.1 * torch.randn(C, generator=g))
    go = torch.randn(B, C, Ln, generator=g).cuda()
    ts = [x, norm.weight, norm.bias] + ([r] if res else [])
    # the torch reference runs on the CPU: torch 2.10+rocm7.0's GPU GroupNorm backward returns wrong d gamma / d beta
    # for 2-D (256, 128) inputs (checked against the CPU and against direct sums), which is exactly the match head's shape
    import copy
    xc = x.detach().cpu().requires_grad_(True)
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
    for a, b_ in zip(ours, want):
        assert _rel(a.cpu(), b_) < 2e-5, _rel(a.cpu(), b_)


def test_pair_pooling_matches_torch():
    from pcr_amd import train_ops as TO
    g = torch.Generator().manual_seed(6)
    o = torch.randn(10, 64, 77, generator=g).cuda().requires_grad_(True)
    go = torch.randn(5, 128, generator=g).cuda()

    def torch_fn():
        x = torch.cat([o[:5], o[5:]], dim=2)
        return torch.cat([x.max(dim=2)[0], x.mean(dim=2)], dim=1)
    ours = _grads(lambda: TO.PoolPair.apply(o), [o], go)
    want = _grads(torch_fn, [o], go)
    for a, b_ in zip(ours, want):
        assert _rel(a, b_) < 1e-6


def test_attention_blocks_match_torch_graph():
    """Self_Attention / FP_SA / corss_attention in training mode (HIP Functions) against the reference's formulas in
    plain torch (pointnet2_utils.py:90-114, 407-437; attention.py:192-219)"""
    import copy
    from mmdet3d.models.pointnet2_utils import Self_Attention, FP_SA
    from mmdet3d.models.attention import corss_attention
    from pcr_amd import train_graph as TG

    def t_block(m, q_in, k_in, v_in, res_in, residual):
        B, Lq, _ = q_in.shape
        d, h = m.q_proj.weight.shape[0], m.nhead
        q = m.q_proj(q_in).permute(0, 2, 1)
        k = m.k_proj(k_in).permute(0, 2, 1)
        v = m.v_proj(v_in).permute(0, 2, 1)
        msg = _t_linattn(q, k, v, h).permute(0, 2, 1)
        msg = m.norm1(m.merge(msg))
        msg = m.norm2(m.mlp(torch.cat([res_in, msg], dim=2)))
        return res_in + msg if residual else msg
    g = torch.Generator().manual_seed(7)
    rnd = lambda *s: torch.randn(*s, generator=g).cuda()     # noqa: E731
    cases = []
    m = Self_Attention(64, 2)
    cases.append(("self", m, lambda mm, f, x, f2, x2: TG.self_attention(mm, f, TG._cm(x)),
                  lambda mm, f, x, f2, x2: t_block(mm, f.permute(0, 2, 1) + mm.pos_mlp(x), f.permute(0, 2, 1) + mm.pos_mlp(x),
                                                   f.permute(0, 2, 1) + mm.pos_mlp(x), f.permute(0, 2, 1), True).permute(0, 2, 1),
                  (64, 50, 64, 50)))
    m = FP_SA(0, 32, 128, 64, 64, 2)
    cases.append(("fp", m, lambda mm, f, x, f2, x2: TG.fp_sa(mm, f, f2, TG._cm(x2)),
                  lambda mm, f, x, f2, x2: t_block(mm, f.permute(0, 2, 1), f2.permute(0, 2, 1),
                                                   f2.permute(0, 2, 1) + mm.pos_mlp2(x2), f.permute(0, 2, 1), False).permute(0, 2, 1),
                  (32, 128, 128, 40)))
    m = corss_attention(64, 2)
    cases.append(("cross", m, lambda mm, f, x, f2, x2: TG.cross_attention(mm, f, f2, TG._cm(x2)),
                  lambda mm, f, x, f2, x2: t_block(mm, f.permute(0, 2, 1), f2.permute(0, 2, 1),
                                                   f2.permute(0, 2, 1) + mm.pos_mlp(x2), f.permute(0, 2, 1), True).permute(0, 2, 1),
                  (64, 100, 64, 100)))
    for name, m, ours_fn, torch_fn, (c1, Lq, c2, Sk) in cases:
        m.load_state_dict(T.seeded_state_dict(T.manifest_of(m), 9))
        m = m.cuda().train()
        ref = copy.deepcopy(m)
        f, x = rnd(3, c1, Lq).requires_grad_(True), rnd(3, Lq, 3)
        f2, x2 = rnd(3, c2, Sk).requires_grad_(True), rnd(3, Sk, 3)
        fr, f2r = f.detach().clone().requires_grad_(True), f2.detach().clone().requires_grad_(True)
        out = ours_fn(m, f, x, f2, x2)
        want = torch_fn(ref, fr, x, f2r, x2)
        assert _rel(out, want) < 2e-5, (name, _rel(out, want))
        go = rnd(*out.shape)
        (out * go).sum().backward()
        (want * go).sum().backward()
        worst = {"f": _rel(f.grad, fr.grad)}
        if name != "self":
            worst["f2"] = _rel(f2.grad, f2r.grad)
        for (k, p), (_, q) in zip(m.named_parameters(), ref.named_parameters()):
            if q.grad is not None:
                worst[k] = _rel(p.grad, q.grad)
        assert max(worst.values()) < 1e-4, (name, worst)


def _opt_params(seed):
    g = torch.Generator().manual_seed(seed)
    shapes = [(64, 67), (64,), (128, 128), (1,), (3, 5, 7), (4099,), (300, 33)]
    return [torch.nn.Parameter(torch.randn(*s, generator=g).cuda()) for s in shapes]


@pytest.mark.parametrize("max_norm", [None, 0.5, 1e6])
def test_fused_adamw_matches_torch_adamw_and_clip(max_norm):
    """pcr_adamw_step_f32 / pcr_grad_sumsq_f32 against clip_grad_norm_ + torch.optim.AdamW over several iterations with
    a cyclic lr / beta1, a parameter that gets no gradient on some iterations, and gradients that are views"""
    from pcr_amd.optim import FusedAdamW
    pa, pb = _opt_params(5), _opt_params(5)
    oa = FusedAdamW(pa, lr=3e-4, weight_decay=0.01)
    ob = torch.optim.AdamW(pb, lr=3e-4, weight_decay=0.01, foreach=False)
    g = torch.Generator().manual_seed(9)
    for it in range(6):
        lr, b1 = 3e-4 * (1 + it), 0.9 - 0.01 * it
        for o in (oa, ob):
            for grp in o.param_groups:
                grp["lr"], grp["betas"] = lr, (b1, 0.999)
        for i, (a, b) in enumerate(zip(pa, pb)):
            if i == 3 and it % 2 == 0:
                a.grad = b.grad = None          # no gradient this iteration: untouched, its step count stays
                continue
            gr = torch.randn(*a.shape, generator=g).cuda() * (10.0 if it == 2 else 1.0)
            if i == 0:
                pad = torch.zeros(64, 96, device="cuda")
                pad[:, :67] = gr
                a.grad = pad[:, :67]            # a non-contiguous view, as the padded dW images are
            else:
                a.grad = gr.clone()
            b.grad = gr.clone()
        if max_norm is None:
            na = oa.step()
            assert na is None
        else:
            na = oa.step(max_norm=max_norm)
            nb = torch.nn.utils.clip_grad_norm_([p for p in pb if p.grad is not None], max_norm, norm_type=2)
            assert float(na) == pytest.approx(float(nb), rel=1e-6)
            for a, b in zip(pa, pb):            # the clipped gradients stay visible in .grad
                if b.grad is not None:
                    assert torch.allclose(a.grad, b.grad, rtol=1e-6, atol=1e-9)
        ob.step()
        for a, b in zip(pa, pb):
            assert torch.allclose(a, b, rtol=1e-6, atol=1e-7), (it, float((a - b).abs().max()))
    # state_dict has torch.optim.AdamW's layout and moves both ways
    sa, sb = oa.state_dict(), ob.state_dict()
    assert set(sa["state"]) == set(sb["state"])
    for k in sb["state"]:
        assert float(sa["state"][k]["step"]) == float(sb["state"][k]["step"])
        assert torch.allclose(sa["state"][k]["exp_avg"], sb["state"][k]["exp_avg"], rtol=1e-5, atol=1e-6)
        assert torch.allclose(sa["state"][k]["exp_avg_sq"], sb["state"][k]["exp_avg_sq"], rtol=1e-5, atol=1e-7)
    pc = _opt_params(5)
    with torch.no_grad():
        for c, b in zip(pc, pb):
            c.copy_(b)
    oc = FusedAdamW(pc, lr=3e-4, weight_decay=0.01)
    import copy
    oc.load_state_dict(copy.deepcopy(sb))       # resume the torch run with the HIP optimizer (own state tensors)
    for c, b in zip(pc, pb):
        gr = torch.randn(*b.shape, generator=g).cuda()
        c.grad, b.grad = gr.clone(), gr.clone()
    oc.step()
    ob.step()
    for c, b in zip(pc, pb):
        assert torch.allclose(c, b, rtol=1e-6, atol=1e-7)


def test_fused_adamw_bumps_versions_so_caches_see_the_update():
    """the update is written through raw pointers: Tensor._version must still move, or the padded-bias cache of the
    training launches and the inference launch plans (keyed by data_ptr + version) would keep serving old weights"""
    from pcr_amd import engine, train_ops as TO
    from pcr_amd.optim import FusedAdamW
    lin = torch.nn.Linear(8, 20).cuda()
    opt = FusedAdamW(lin.parameters(), lr=0.1)
    v0 = [p._version for p in lin.parameters()]
    key0 = engine.param_version(lin)
    pad0 = TO.pad32(lin.bias, 20).clone()
    for p in lin.parameters():
        p.grad = torch.ones_like(p)
    opt.step(max_norm=1.0)
    assert all(p._version > v for p, v in zip(lin.parameters(), v0))
    assert engine.param_version(lin) != key0
    pad1 = TO.pad32(lin.bias, 20)
    assert torch.equal(pad1[:20], lin.bias.detach()) and not torch.equal(pad1, pad0)


def test_prepacked_weights_equal_fresh_packs_and_follow_updates():
    """train_ops.prepack: one launch packs every conv / linear weight; pack_dev / pack_both must return the same images
    as the per-layer launch, and stop hitting the cache as soon as a weight's version moves"""
    from pcr_amd import train_ops as TO
    net = torch.nn.Sequential(torch.nn.Linear(67, 40), torch.nn.Conv1d(40, 96, 1), torch.nn.Conv2d(96, 33, 1)).cuda()
    ws = [net[0].weight, net[1].weight.view(96, 40), net[2].weight.view(33, 96)]
    TO._PREPACKS.clear()
    fresh = [tuple(t.clone() for t in TO.pack_both(w)) for w in ws]          # cache empty: per-layer launches
    TO.prepack(net)
    for w, (f0, f1) in zip(ws, fresh):
        assert TO._lookup(w) is not None
        c0, c1 = TO.pack_both(w)
        assert torch.equal(c0, f0) and torch.equal(c1, f1)
        assert torch.equal(TO.pack_dev(w), f0) and torch.equal(TO.pack_dev(w, transpose=True), f1)
    with torch.no_grad():
        net[1].weight.mul_(2.0)                                              # an update: the version moves
    assert TO._lookup(ws[1]) is None and TO._lookup(ws[0]) is not None
    assert torch.equal(TO.pack_both(ws[1])[0], 2.0 * fresh[1][0])            # packed on the spot from the new values
    TO.prepack(net)
    assert torch.equal(TO.pack_both(ws[1])[0], 2.0 * fresh[1][0]) and TO._lookup(ws[1]) is not None
    # a second model gets its own table (the first one's images stay valid), and a dropped model takes its table along
    net2 = torch.nn.Sequential(torch.nn.Linear(16, 8)).cuda()
    TO.prepack(net2)
    assert TO._lookup(net2[0].weight) is not None and TO._lookup(ws[0]) is not None and len(TO._PREPACKS) == 2
    del net2
    import gc
    gc.collect()
    assert len(TO._PREPACKS) == 1
    TO._PREPACKS.clear()


def test_prepacked_bf16_images_equal_fresh_packs_and_follow_updates():
    """the bf16 hi / lo images of the fused chains' weights (train_ops.pack_both_bf): the first request packs on the spot and
    registers the weight, the next prepack() launch refreshes it with the f32 images (pcr_pack_weights_multi_f32 kind 1) --
    same bits as the per-layer launches, for row / column counts that are not multiples of 16 / 32 too -- and a stale image
    is never handed out"""
    from pcr_amd import train_ops as TO
    net = torch.nn.Sequential(torch.nn.Linear(67, 128), torch.nn.Linear(128, 32), torch.nn.Linear(40, 96),
                              torch.nn.Linear(64, 64)).cuda()
    ws = [m.weight for m in net]
    TO._PREPACKS.clear()
    fresh = [tuple(t.clone() for t in TO.pack_both_bf(w)) for w in ws]       # no table yet: per-layer launches
    TO.prepack(net)
    tab = TO._PREPACKS[net]
    for w in ws[:3]:
        assert tab.lookup_bf(w) is None          # registered by this request, packed by the NEXT refresh
    for w, (f0, f1) in zip(ws[:3], fresh):       # (misses pack on the spot: still the right bits)
        c0, c1 = TO.pack_both_bf(w)
        assert torch.equal(c0, f0) and torch.equal(c1, f1)
    f32_before = [tuple(t.clone() for t in TO.pack_both(w)) for w in ws]
    TO.prepack(net)                              # table rebuilt with the three bf16 entries, one launch
    for w, (f0, f1), (g0, g1) in zip(ws[:3], fresh, f32_before):
        hit = tab.lookup_bf(w)
        assert hit is not None and torch.equal(hit[0], f0) and torch.equal(hit[1], f1)
        c0, c1 = TO.pack_both(w)                 # the f32 images stayed where they were, same bits
        assert torch.equal(c0, g0) and torch.equal(c1, g1)
    assert tab.entries[ws[3].data_ptr()][6] is None          # never requested: no bf16 image kept for it
    with torch.no_grad():
        ws[0].mul_(0.5)
    assert tab.lookup_bf(ws[0]) is None          # stale: not handed out
    c0, c1 = TO.pack_both_bf(ws[0])
    assert not torch.equal(c0, fresh[0][0])
    TO.prepack(net)
    hit = tab.lookup_bf(ws[0])
    assert hit is not None and torch.equal(hit[0], c0) and torch.equal(hit[1], c1)
    TO._PREPACKS.clear()


def test_trainer_gradients_do_not_depend_on_where_the_weight_images_come_from(grad_floor):
    """one Trainer iteration with the one-launch weight refresh (fused=True: images from the prepack table, the bf16 ones
    after their registration) against the same iteration with per-layer packs (fused=False): identical weights in, so
    every gradient must come out bit for bit the same -- at iteration 0 (bf16 images packed on the spot in both) and at
    iteration 2 (prepacked in one, on the spot in the other; the optimizers are bypassed so the weights stay equal)"""
    import bench
    from pcr_amd import train
    s1, s2 = T.synthetic_pairs(8, 128, seed=2, kind="randn")
    dev = "cuda"
    ids1, ids2 = torch.arange(8), torch.tensor([0, 1, 2, 3, 9, 9, 9, 9])
    data = dict(sparse_1=list(s1.to(dev)), sparse_2=list(s2.to(dev)), dense_1=list(s1.to(dev)), dense_2=list(s2.to(dev)),
                label_1=[torch.zeros(1, dtype=torch.long, device=dev)] * 8,
                label_2=[torch.zeros(1, dtype=torch.long, device=dev)] * 8,
                id_1=[i.view(1).to(dev) for i in ids1], id_2=[i.view(1).to(dev) for i in ids2])
    grads = []
    for fused in (True, False):
        m, _ = bench.build_pt_model([128, 64, 32])
        m.train()
        tr = train.Trainer(m, max_iters=8, lr=0.0, grad_clip=None, fused=fused)    # lr 0: the weights never move ...
        per_iter = []
        for _ in range(3):
            tr.step(data)
            per_iter.append({k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None})
            with torch.no_grad():
                for p in m.parameters():
                    p.add_(0.0)                                                  # ... but their versions do
        grads.append(per_iter)
    for it in (0, 2):
        a, b = grads[0][it], grads[1][it]
        assert set(a) == set(b) and len(a) > 100
        for k in a:
            assert torch.equal(a[k], b[k]), (it, k, float((a[k] - b[k]).abs().max()))


def test_fused_adamw_is_reproducible_and_refuses_host_tensors():
    from pcr_amd import _lib as L
    from pcr_amd.optim import FusedAdamW
    outs = []
    for _ in range(2):
        ps = _opt_params(7)
        o = FusedAdamW(ps, lr=1e-3)
        g = torch.Generator().manual_seed(1)
        for _ in range(3):
            for p in ps:
                p.grad = torch.randn(*p.shape, generator=g).cuda()
            n = o.step(max_norm=1.0)
        outs.append([p.detach().clone() for p in ps] + [n.clone()])
    assert all(torch.equal(a, b) for a, b in zip(*outs))
    cpu = [torch.nn.Parameter(torch.randn(4))]
    cpu[0].grad = torch.randn(4)
    with pytest.raises(L.PcrError):
        FusedAdamW(cpu).step()


@pytest.mark.parametrize("B,C,N,K,H", [(3, 64, 128, 48, 2), (2, 32, 50, 7, 4), (1, 48, 33, 33, 3)])
def test_local_attention_core_matches_torch_autograd(B, C, N, K, H):
    """train_ops.LocalAttn (one query token per point over its K neighbours, per-edge gradients folded by the grouping
    backward) against the reference's formulation in plain torch: LinearAttention on the gathered (B N, K, C) tensor"""
    from pcr_amd import train_ops as TO
    g = torch.Generator().manual_seed(C + K)
    qkv = torch.randn(B, 3 * C, N, generator=g).cuda().requires_grad_(True)
    idx = torch.stack([torch.stack([torch.randperm(N, generator=g)[:K] for _ in range(N)]) for _ in range(B)]).int().cuda()
    go = torch.randn(B, C, N, generator=g).cuda()
    out = TO.LocalAttn.apply(qkv, idx, H, 1e-6)
    out.backward(go)
    got = qkv.grad.clone()
    q2 = qkv.detach().clone().requires_grad_(True)
    rows = q2.permute(0, 2, 1)                                                   # (B,N,3C)
    li = idx.long()
    nb = torch.gather(rows.unsqueeze(1).expand(-1, N, -1, -1), 2, li.unsqueeze(-1).expand(-1, -1, -1, 3 * C))   # (B,N,K,3C)
    D = C // H
    Q = (F.elu(rows[..., :C]) + 1).reshape(B * N, 1, H, D)
    Kf = (F.elu(nb[..., C:2 * C]) + 1).reshape(B * N, K, H, D)
    V = nb[..., 2 * C:].reshape(B * N, K, H, D) / K
    KV = torch.einsum("nshd,nshv->nhdv", Kf, V)
    Z = 1 / (torch.einsum("nlhd,nhd->nlh", Q, Kf.sum(dim=1)) + 1e-6)
    want = (torch.einsum("nlhd,nhdv,nlh->nlhv", Q, KV, Z) * K).reshape(B, N, C).permute(0, 2, 1)
    assert _rel(out, want) < 2e-5, _rel(out, want)
    want.backward(go)
    assert _rel(got, q2.grad) < 2e-5, _rel(got, q2.grad)
    # bit-reproducible
    qkv.grad = None
    TO.LocalAttn.apply(qkv, idx, H, 1e-6).backward(go)
    assert torch.equal(qkv.grad, got)
