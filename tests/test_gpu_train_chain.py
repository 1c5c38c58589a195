"""GPU: the fused per-token chains of the attention blocks in training mode (csrc/train_chain_kernels.hip, round 5)
against the unfused launches they replace (one launch per layer: train_ops.dense / tnorm, themselves pinned to torch
autograd and to the reference's train_step goldens) and against plain torch autograd -- forward values, every
gradient, ragged token counts, bit-reproducibility."""
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return float((a - b).abs().max()) / max(1e-6, float(b.abs().max()))


class Tail(nn.Module):
    """merge / norm1 / mlp / norm2 of an attention block (models/pointnet2_utils.py:63-88, 380-405; attention.py:165-190)"""

    def __init__(self, d, c1, hid, out, seed):
        super().__init__()
        g = torch.Generator().manual_seed(seed)
        self.merge = nn.Linear(d, d, bias=False)
        self.mlp = nn.Sequential(nn.Linear(c1 + d, hid, bias=False), nn.ReLU(True), nn.Linear(hid, out, bias=False))
        self.norm1, self.norm2 = nn.LayerNorm(d), nn.LayerNorm(out)
        with torch.no_grad():
            for p in self.parameters():
                p.copy_(torch.randn(p.shape, generator=g) * (0.5 if p.dim() == 1 else 1.0 / p.shape[-1] ** 0.5) +
                        (1.0 if p.dim() == 1 and p is not self.norm1.bias and p is not self.norm2.bias else 0.0))


def _unfused(m, msg, res, residual):
    from pcr_amd import train_ops as TO
    n1 = TO.tnorm(TO.dense(msg, m.merge.weight), m.norm1)
    f0 = TO.dense(res, m.mlp[0].weight, x2=n1, relu=True)
    return TO.tnorm(TO.dense(f0, m.mlp[2].weight), m.norm2, res=res if residual else None)


def _torch(m, msg, res, residual):
    x = m.norm1(m.merge(msg.permute(0, 2, 1)))
    y = m.norm2(m.mlp(torch.cat([res.permute(0, 2, 1), x], dim=2)))
    if residual:
        y = y + res.permute(0, 2, 1)
    return y.permute(0, 2, 1)


def _well_conditioned_tokens(m, msg, res, margin=2e-4):
    """(B,1,L) mask of the tokens none of whose hidden pre-activations W0 [res ; LN1(Wm msg)] lies within `margin` of
    zero (float64).  A ReLU whose argument is ~1e-5 from zero opens in one arithmetic and stays shut in another -- ~1 token
    in 5000 between the f32 chains and split bf16 -- and that token's whole gradient row then differs by several per
    cent: the same near-tie effect the max-pool tests meet.  Every layer of the chain is per token,
    so zeroing the upstream gradient of those tokens removes them from every gradient of the comparison."""
    with torch.no_grad():
        m64 = m.double()
        x = m64.norm1(m64.merge(msg.double().permute(0, 2, 1)))
        h = m64.mlp[0](torch.cat([res.double().permute(0, 2, 1), x], dim=2))
        ok = (h.abs().amin(dim=2) > margin).float().unsqueeze(1)
        m.float()
    return ok


def _run(fn, m, msg, res, residual, go):
    msg = msg.detach().clone().requires_grad_(True)
    res = res.detach().clone().requires_grad_(True)
    for p in m.parameters():
        p.grad = None
    out = fn(m, msg, res, residual)
    (out * go).sum().backward()
    grads = {"msg": msg.grad, "res": res.grad}
    grads.update({k: p.grad.clone() for k, p in m.named_parameters()})
    return out.detach(), grads


# arithmetic of the chains' matrix phases (train_ops.TRAIN_PRECISION): "f32" = the unfused launches' own fmaf chains;
# "bf16x3" (the default) = the gradient products (dx, dW) as split bf16 on the bf16 matrix core, forward and recomputation
# f32: forward values and ReLU masks are the unfused graph's, gradients carry ~2^-17 per product; "bf16x3_all" (opt-in) =
# the forward as split bf16 too.  Bounds per arithmetic:
# (forward vs unfused, forward vs torch, gradient floor vs torch, gradient vs unfused)
TOL = {"f32": (2e-6, 1e-5, 2e-5, 3e-5), "bf16x3": (2e-6, 1e-5, 3e-5, 3e-5), "bf16x3_all": (3e-5, 3e-5, 5e-5, 5e-5)}


@pytest.fixture(params=["f32", "bf16x3", "bf16x3_all"])
def prec(request):
    from pcr_amd import train_ops as TO
    prev = TO.set_train_precision(request.param)
    yield request.param
    TO.set_train_precision(prev)


SHAPES = [(32, 32, 64, 32, True), (64, 64, 128, 64, True), (64, 64, 128, 64, False), (64, 64, 128, 128, False),
          (64, 32, 128, 64, False), (64, 3, 128, 32, False)]


@pytest.mark.parametrize("d,c1,hid,out,residual", SHAPES)
@pytest.mark.parametrize("B,Ln", [(5, 128), (3, 100), (7, 37), (300, 64)])
def test_fused_tail_equals_the_unfused_launches_and_torch(d, c1, hid, out, residual, B, Ln, prec):
    from pcr_amd import train_ops as TO
    from pcr_amd import _lib as L
    assert L.load().pcr_attn_tail_ok(d, c1, hid, out, int(residual)) == 1
    m = Tail(d, c1, hid, out, seed=d + c1 + out).cuda()
    g = torch.Generator().manual_seed(B * 1000 + Ln)
    msg, res = torch.randn(B, d, Ln, generator=g).cuda(), torch.randn(B, c1, Ln, generator=g).cuda()
    go = torch.randn(B, out, Ln, generator=g).cuda() * _well_conditioned_tokens(m, msg, res)

    def fused(mm, a, b, r):
        y = TO.attn_tail(mm, a, b, r)
        assert y is not None
        return y
    o_f, g_f = _run(fused, m, msg, res, residual, go)
    o_u, g_u = _run(_unfused, m, msg, res, residual, go)
    o_t, g_t = _run(_torch, m, msg, res, residual, go)
    t_fu, t_ft, t_g, t_gu = TOL[prec]
    assert _rel(o_f, o_u) < t_fu and _rel(o_f, o_t) < t_ft, (_rel(o_f, o_u), _rel(o_f, o_t))
    for k in g_t:
        # against torch autograd the fused launch must be no further away than the unfused launches are (+ rounding)
        e_f, e_u = _rel(g_f[k], g_t[k]), _rel(g_u[k], g_t[k])
        assert e_f < max(t_g, 2 * e_u), (k, e_f, e_u)
        assert _rel(g_f[k], g_u[k]) < t_gu, (k, _rel(g_f[k], g_u[k]))
    # bit-reproducible: partial sums are reduced in a fixed order, no float atomics
    o_f2, g_f2 = _run(fused, m, msg, res, residual, go)
    assert torch.equal(o_f, o_f2) and all(torch.equal(g_f[k], g_f2[k]) for k in g_f)


def test_shapes_without_an_instantiation_keep_the_unfused_graph():
    from pcr_amd import train_ops as TO
    m = Tail(128, 128, 256, 128, seed=1).cuda()            # SA3's block (d = 128): not fused
    msg, res = torch.randn(2, 128, 32).cuda(), torch.randn(2, 128, 32).cuda()
    assert TO.attn_tail(m, msg, res, True) is None
    m = Tail(64, 64, 128, 64, seed=1).cuda()
    prev, TO.FUSED_CHAINS = TO.FUSED_CHAINS, False
    try:
        assert TO.attn_tail(m, torch.randn(2, 64, 64).cuda(), torch.randn(2, 64, 64).cuda(), True) is None
    finally:
        TO.FUSED_CHAINS = prev


# ---- the head: position MLP + residual add + projections -------------------------------------------------------------
class Head(nn.Module):
    def __init__(self, c, hd, d, n, seed):
        super().__init__()
        g = torch.Generator().manual_seed(seed)
        self.pos_mlp = nn.Sequential(nn.Linear(3, hd), nn.ReLU(True), nn.Linear(hd, c))
        self.proj = nn.ModuleList([nn.Linear(c, d, bias=False) for _ in range(n)])
        with torch.no_grad():
            for p in self.parameters():
                p.copy_(torch.randn(p.shape, generator=g) * (0.3 if p.dim() == 1 else 1.0 / p.shape[-1] ** 0.5))


def _head_unfused(m, x, xyz, src):
    from pcr_amd import train_ops as TO
    h = TO.dense(xyz, m.pos_mlp[0].weight, m.pos_mlp[0].bias, relu=True)
    fp = TO.dense(h, m.pos_mlp[2].weight, m.pos_mlp[2].bias, res=x)
    return torch.cat([TO.dense(fp if (src >> j) & 1 else x, p.weight) for j, p in enumerate(m.proj)], dim=1)


def _head_torch(m, x, xyz, src):
    xt = x.permute(0, 2, 1)
    fp = xt + m.pos_mlp(xyz.permute(0, 2, 1))
    return torch.cat([p(fp if (src >> j) & 1 else xt) for j, p in enumerate(m.proj)], dim=2).permute(0, 2, 1)


def _run_head(fn, m, x, xyz, src, go):
    x = x.detach().clone().requires_grad_(True)
    for p in m.parameters():
        p.grad = None
    out = fn(m, x, xyz, src)
    (out * go).sum().backward()
    grads = {"x": x.grad}
    grads.update({k: p.grad.clone() for k, p in m.named_parameters()})
    return out.detach(), grads


HEADS = [(32, 32, 32, 3, 7), (64, 64, 64, 3, 7), (64, 64, 64, 2, 2), (128, 64, 64, 2, 2), (64, 64, 64, 3, 4)]


@pytest.mark.parametrize("c,hd,d,n,src", HEADS)
@pytest.mark.parametrize("B,Ln", [(5, 128), (3, 100), (7, 37), (300, 32)])
def test_fused_head_equals_the_unfused_launches_and_torch(c, hd, d, n, src, B, Ln, prec):
    from pcr_amd import train_ops as TO
    m = Head(c, hd, d, n, seed=c + d + n).cuda()
    g = torch.Generator().manual_seed(B * 1000 + Ln)
    x, xyz = torch.randn(B, c, Ln, generator=g).cuda(), torch.randn(B, 3, Ln, generator=g).cuda()
    go = torch.randn(B, n * d, Ln, generator=g).cuda()

    def fused(mm, a, z, s):
        y = TO.attn_head(mm.pos_mlp, a, z, tuple(p.weight for p in mm.proj), s)
        assert y is not None
        return y
    o_f, g_f = _run_head(fused, m, x, xyz, src, go)
    o_u, g_u = _run_head(_head_unfused, m, x, xyz, src, go)
    o_t, g_t = _run_head(_head_torch, m, x, xyz, src, go)
    t_fu, t_ft, t_g, t_gu = TOL[prec]
    assert _rel(o_f, o_u) < t_fu and _rel(o_f, o_t) < t_ft, (_rel(o_f, o_u), _rel(o_f, o_t))
    for k in g_t:
        e_f, e_u = _rel(g_f[k], g_t[k]), _rel(g_u[k], g_t[k])
        assert e_f < max(t_g, 2 * e_u), (k, e_f, e_u)
        assert _rel(g_f[k], g_u[k]) < t_gu, (k, _rel(g_f[k], g_u[k]))
    o_f2, g_f2 = _run_head(fused, m, x, xyz, src, go)
    assert torch.equal(o_f, o_f2) and all(torch.equal(g_f[k], g_f2[k]) for k in g_f)


def test_key_value_attention_on_the_fused_buffer_equals_separate_tensors():
    from pcr_amd import train_ops as TO
    g = torch.Generator().manual_seed(3)
    B, d, Lq, Sk = 4, 64, 100, 37
    q = torch.randn(B, d, Lq, generator=g).cuda().requires_grad_(True)
    kv = torch.randn(B, 2 * d, Sk, generator=g).cuda().requires_grad_(True)
    go = torch.randn(B, d, Lq, generator=g).cuda()
    out = TO.LinAttnKV.apply(q, kv, 2, 1e-6)
    (out * go).sum().backward()
    q2 = q.detach().clone().requires_grad_(True)
    k2 = kv.detach()[:, :d].clone().requires_grad_(True)
    v2 = kv.detach()[:, d:].clone().requires_grad_(True)
    ref = TO.LinAttn.apply(q2, k2, v2, 2, 1e-6)
    (ref * go).sum().backward()
    assert torch.equal(out, ref) and torch.equal(q.grad, q2.grad)
    assert torch.equal(kv.grad[:, :d], k2.grad) and torch.equal(kv.grad[:, d:], v2.grad)


def test_attention_core_reads_the_partner_clouds_keys_without_a_rolled_copy():
    """LinAttnQKV(kv_roll = b) on one (2b, 3d, L) buffer == LinAttn(q, roll(k, b), roll(v, b)): the matching stages' "halves
    swapped" (ReIDNet.py:231-247) addressed inside the core, forward and every gradient bit for bit"""
    from pcr_amd import train_ops as TO
    g = torch.Generator().manual_seed(4)
    b, d, Ln = 5, 64, 100
    qkv = torch.randn(2 * b, 3 * d, Ln, generator=g).cuda().requires_grad_(True)
    go = torch.randn(2 * b, d, Ln, generator=g).cuda()
    out = TO.LinAttnQKV.apply(qkv, 2, 1e-6, b)
    (out * go).sum().backward()
    x = qkv.detach().clone().requires_grad_(True)
    ref = TO.LinAttn.apply(x[:, :d].contiguous(), torch.roll(x[:, d:2 * d], b, 0).contiguous(),
                           torch.roll(x[:, 2 * d:], b, 0).contiguous(), 2, 1e-6)
    (ref * go).sum().backward()
    assert torch.equal(out, ref) and torch.equal(qkv.grad, x.grad)


def test_matching_stages_on_the_fused_path_equal_the_unfused_graph():
    """train_graph.match_logits (xcorr_eff + point-cat): fused heads / tails with the partner addressed by the attention
    core against the graph of rounds 2-4 (rolled copies of the batch, one launch per layer): logits and the gradients of
    the encodings and of every matching parameter"""
    import sys
    import os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    from pcr_amd import train_graph as TG
    from pcr_amd import train_ops as TO
    model, _ = bench.build_pt_model([128, 64, 32])
    model.train()
    # (the chains in the unfused launches' own arithmetic: this test is about the graph -- partner addressing, fused
    # buffers -- and split bf16 would add its ReLU near-tie flips to the comparison; the arithmetic itself is bounded above)
    prev_prec = TO.set_train_precision("f32")
    g = torch.Generator().manual_seed(8)
    b, n = 6, 128
    h = torch.randn(2 * b, 64, n, generator=g).cuda()
    xyz = torch.randn(2 * b, n, 3, generator=g).cuda()

    def run(fused):
        prev, TO.FUSED_CHAINS = TO.FUSED_CHAINS, fused
        try:
            model.zero_grad(set_to_none=True)
            hh = h.clone().requires_grad_(True)
            logits, _ = TG.match_logits(model, hh[:b], xyz[:b], hh[b:], xyz[b:])
            (logits * torch.arange(1, b + 1, device="cuda").float()).sum().backward()
            grads = {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}
            return logits.detach(), hh.grad.clone(), grads
        finally:
            TO.FUSED_CHAINS = prev
    try:
        lf, hf, gf = run(True)
        lu, hu, gu = run(False)
    finally:
        TO.set_train_precision(prev_prec)
    assert _rel(lf, lu) < 1e-5 and _rel(hf, hu) < 3e-5, (_rel(lf, lu), _rel(hf, hu))
    assert set(gf) == set(gu) and len(gf) >= 30
    for k in gu:
        assert _rel(gf[k], gu[k]) < 1e-4, (k, _rel(gf[k], gu[k]))


# ---- partial-sum reductions into compact gradients ------------------------------------------------------------------------
def test_gradients_arrive_compact_and_a_parameter_used_twice_is_summed_correctly(prec):
    """train_ops.reduce_regions hands autograd compact contiguous gradients (no padded views to clone), and a parameter that
    enters the graph twice -- once through a fused chain, once sliced and re-joined into a dense layer -- gets the sum of
    both contributions (the hazard that ruled out deferring the reductions to the end of the pass)"""
    from pcr_amd import train_ops as TO
    m = Tail(64, 64, 128, 64, seed=2).cuda()
    g = torch.Generator().manual_seed(12)
    msg, x = torch.randn(9, 64, 128, generator=g).cuda(), torch.randn(9, 64, 128, generator=g).cuda()
    go = torch.randn(9, 64, 128, generator=g).cuda() * _well_conditioned_tokens(m, msg, x)

    def run(tail):
        for p in m.parameters():
            p.grad = None
        w_cat = torch.cat([m.merge.weight[:32], m.merge.weight[32:]], dim=0)
        y = TO.dense(tail(m, msg, x, True), w_cat)
        (y * go).sum().backward()
        return {k: p.grad.clone() for k, p in m.named_parameters()}
    a = run(lambda mm, a_, b_, r: TO.attn_tail(mm, a_, b_, r))
    b = run(_unfused)
    assert all(v.is_contiguous() for v in a.values())
    for k in a:
        assert _rel(a[k], b[k]) < TOL[prec][3], (k, _rel(a[k], b[k]))
