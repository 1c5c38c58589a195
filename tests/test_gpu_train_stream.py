"""GPU: the wave-autonomous train-dense kernels (csrc/train_stream_kernels.hip) against the tile kernels of
csrc/train_kernels.hip on the same launches (y / dx bit for bit: same MFMA products in the same k order; the partial
sums within rounding), and the grouped-MLP Function through them against plain torch autograd."""
import ctypes

import pytest
import torch

from pcr_amd import _lib as L

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return float((a - b).abs().max()) / max(1e-6, float(b.abs().max()))


class _policy:
    """pcr_set_stream_min_blocks for the duration of a with-block"""

    def __init__(self, n):
        self.n = n

    def __enter__(self):
        self.old = L.load().pcr_set_stream_min_blocks(self.n)

    def __exit__(self, *a):
        L.load().pcr_set_stream_min_blocks(self.old)


@pytest.mark.parametrize("B,cin,cout,Ln", [(3, 32, 32, 4096), (5, 32, 32, 96), (2, 64, 64, 3072), (7, 64, 64, 32),
                                           (3, 32, 64, 160), (3, 64, 32, 640), (130, 32, 32, 64), (2, 128, 128, 1536),
                                           (37, 128, 128, 32)])
def test_stream_forward_equals_tile_forward(B, cin, cout, Ln):
    from pcr_amd import train_ops as TO
    g = torch.Generator().manual_seed(B * 1000 + Ln)
    x = torch.randn(B, cin, Ln, generator=g).cuda()
    W = (torch.randn(cout, cin, generator=g) / cin ** 0.5).cuda()
    bias = torch.randn(cout, generator=g).cuda()
    isc, ish = (torch.rand(cin, generator=g) + 0.5).cuda(), (torch.randn(cin, generator=g) * 0.3).cuda()
    wp = TO.pack_dev(W)
    out = {}
    for tag, n in (("tile", 1 << 30), ("stream", 0)):
        with _policy(n):
            y, st = TO.tdense_fwd(x, wp, cout, isc=isc, ish=ish, in_relu=True, bias=bias, want_stats=True)
            y0, _ = TO.tdense_fwd(x, wp, cout, bias=None)
        out[tag] = (y, st.sum(0), y0, st.shape[0])
    ref = torch.einsum("oc,bcl->bol", W.double(), torch.relu(x.double() * isc.double()[None, :, None] + ish.double()[None, :, None])) \
        + bias.double()[None, :, None]
    assert torch.equal(out["tile"][0], out["stream"][0])
    assert torch.equal(out["tile"][2], out["stream"][2])
    assert _rel(out["stream"][0].double(), ref) < 1e-5
    want = torch.stack([ref.sum((0, 2)), (ref * ref).sum((0, 2))])
    assert _rel(out["stream"][1].double(), want) < 1e-5 and _rel(out["tile"][1].double(), want) < 1e-5


@pytest.mark.parametrize("B,c,S,K,mode", [(3, 32, 128, 32, 1), (3, 32, 128, 32, 3), (2, 32, 64, 48, 3), (5, 32, 3, 32, 1),
                                          (130, 32, 2, 16, 3), (4, 32, 40, 20, 3), (3, 64, 64, 48, 1), (2, 64, 64, 48, 3),
                                          (9, 64, 1, 32, 3), (70, 64, 2, 16, 1)])
def test_stream_backward_equals_tile_backward(B, c, S, K, mode):
    from pcr_amd import train_ops as TO
    Ln = S * K
    g = torch.Generator().manual_seed(B * 1000 + Ln + mode)
    x = torch.randn(B, c, Ln, generator=g).cuda()
    y = torch.randn(B, c, Ln, generator=g).cuda()
    W = (torch.randn(c, c, generator=g) / c ** 0.5).cuda()
    k = dict(ka=(torch.rand(c, generator=g) + 0.5).cuda(), kb=(torch.randn(c, generator=g) * 0.05).cuda(),
             kc=(torch.randn(c, generator=g) * 0.05).cuda())
    isc, ish = (torch.rand(c, generator=g) + 0.5).cuda(), (torch.randn(c, generator=g) * 0.3).cuda()
    iinv = 1.0 / isc
    wpT = TO.pack_dev(W, transpose=True)
    kw = dict(dy_mode=mode, y=y, k=k, isc=isc, ish=ish, iinv=iinv, in_relu=True, wpT=wpT, want_dstats=True)
    if mode == 3:
        gin = torch.randn(B, c, S, generator=g).cuda()
        kw.update(argmax=torch.randint(0, K, (B, c, S), generator=g, dtype=torch.int32).cuda(),
                  pooled=torch.randn(B, c, S, generator=g).cuda(), K=K, S=S)
    else:
        gin = torch.randn(B, c, Ln, generator=g).cuda()
    out = {}
    for tag, n in (("tile", 1 << 30), ("stream", 0)):
        with _policy(n):
            r = TO.tdense_bwd(gin, x, c, **kw)
        out[tag] = r
    if mode == 3:
        # the routed form (what SaEdgeTrain passes): g zeroed where the pooled activation is, no `pooled` tensor
        kr = dict(kw, pooled=None)
        gz = torch.where(kw["pooled"] > 0, gin, torch.zeros_like(gin))
        with _policy(1 << 30):
            t2 = TO.tdense_bwd(gz, x, c, **kr)
        assert torch.equal(t2["dx"], out["tile"]["dx"]) and torch.equal(t2["dW"], out["tile"]["dW"])
        with _policy(0):
            out["stream"] = TO.tdense_bwd(gz, x, c, **kr)
        kw_stream = (gz, kr)
    else:
        kw_stream = (gin, kw)
    a, b = out["tile"], out["stream"]
    assert b["dstats"].shape[0] != a["dstats"].shape[0] or B * Ln < 64          # (the two forms have different grids)
    assert torch.equal(a["dx"], b["dx"])
    # float64 reference of the sums
    gd = gin.double()
    if mode == 3:
        full = torch.zeros(B, c, S, K, dtype=torch.float64, device="cuda")
        sel = torch.where(kw["pooled"] > 0, gd, torch.zeros_like(gd))
        full.scatter_(3, kw["argmax"].long().unsqueeze(-1), sel.unsqueeze(-1))
        gd = full.reshape(B, c, Ln)
    dy = k["ka"].double()[None, :, None] * gd + k["kb"].double()[None, :, None] * y.double() + k["kc"].double()[None, :, None]
    pre = x.double() * isc.double()[None, :, None] + ish.double()[None, :, None]
    fx = torch.relu(pre)
    dW = torch.einsum("bol,bcl->oc", dy, fx)
    db = dy.sum((0, 2))
    dx = torch.einsum("oc,bol->bcl", W.double(), dy) * (pre > 0)
    ds = torch.stack([dx.sum((0, 2)), (dx * x.double()).sum((0, 2))])
    for r in (a, b):
        assert _rel(r["dx"].double(), dx) < 1e-5
        assert _rel(r["dW"].double(), dW) < 1e-5
        assert _rel(r["db"].double(), db) < 1e-5
        assert _rel(r["dstats"].sum(0).double(), ds) < 1e-5
    # and run to run identical
    with _policy(0):
        r2 = TO.tdense_bwd(kw_stream[0], x, c, **kw_stream[1])
    assert torch.equal(r2["dW"], b["dW"]) and torch.equal(r2["db"], b["db"]) and torch.equal(r2["dstats"], b["dstats"])


@pytest.mark.parametrize("B,N,S,K,D,widths", [(3, 128, 128, 32, 0, (32, 32, 32)), (4, 128, 64, 48, 32, (64, 64, 64))])
def test_sa_edge_train_through_the_stream_kernels(B, N, S, K, D, widths):
    from test_gpu_train_ops import test_sa_edge_train_matches_torch_autograd as body
    with _policy(0):
        body(B, N, S, K, D, widths)


@pytest.mark.parametrize("B,c,S,K", [(3, 32, 128, 32), (2, 64, 64, 48), (2, 32, 32, 48), (5, 32, 6, 64), (4, 64, 3, 96),
                                     (70, 64, 2, 48)])
def test_fused_pooling_equals_the_pool_kernel(B, c, S, K):
    """the last layer's launch with pool=(K, gamma): the winners it leaves (raw y at the winning row, the row) against
    pcr_sa_pool_fwd_f32 on the stored y with the same scale / shift, for both signs of gamma"""
    from pcr_amd import train_ops as TO
    Ln = S * K
    g = torch.Generator().manual_seed(B + c + K)
    x = torch.randn(B, c, Ln, generator=g).cuda()
    W = (torch.randn(c, c, generator=g) / c ** 0.5).cuda()
    bias = torch.randn(c, generator=g).cuda()
    isc, ish = (torch.rand(c, generator=g) + 0.5).cuda(), (torch.randn(c, generator=g) * 0.3).cuda()
    gamma = torch.randn(c, generator=g).cuda()
    gamma[0] = 0.0
    wp = TO.pack_dev(W)
    with _policy(0):
        y, st, won = TO.tdense_fwd(x, wp, c, isc=isc, ish=ish, in_relu=True, bias=bias, want_stats=True, pool=(K, gamma))
        y_plain, st_plain = TO.tdense_fwd(x, wp, c, isc=isc, ish=ish, in_relu=True, bias=bias, want_stats=True)
    assert won is not None
    assert torch.equal(y, y_plain) and torch.allclose(st.sum(0), st_plain.sum(0), rtol=1e-5, atol=1e-3)
    ymax, arg = won
    scale = gamma * 0.7                                   # (invstd > 0: the sign of the BatchNorm scale is gamma's)
    shift = (torch.randn(c, generator=g) * 0.5).cuda()
    lib = L.load()
    pooled = torch.empty(B, c, S, device="cuda")
    am = torch.empty(B, c, S, dtype=torch.int32, device="cuda")
    ym = torch.empty(B, c, S, device="cuda")
    L.check(lib.pcr_sa_pool_fwd_f32(L.ptr(y), L.ptr(scale), L.ptr(shift), L.ptr(pooled), L.ptr(am), L.ptr(ym), B, c, S, K,
                                    L.stream_ptr()), "pcr_sa_pool_fwd_f32")
    fused = torch.relu(ymax * scale[None, :, None] + shift[None, :, None])
    assert torch.allclose(fused, pooled, rtol=0, atol=1e-6)
    open_ = pooled > 0                                     # (where the ReLU is closed the row is arbitrary: no gradient)
    yv = y.view(B, c, S, K)
    assert torch.equal(torch.gather(yv, 3, arg.long().unsqueeze(-1)).squeeze(-1), ymax)      # arg points at ymax
    assert torch.equal(ymax[open_], ym[open_])
    assert torch.equal(arg[open_], am[open_])
