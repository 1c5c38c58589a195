"""Small per-sample ops of the model path that are not part of a larger fused launch."""
import torch

from . import _lib as L


def pool_both(x):
    """(B,C,L) -> (B,2C) = [max over L, mean over L]  (ReIDNet.get_pooled_feats, pool_type='both')"""
    L.require_cuda(x)
    x = x.contiguous()
    B, C, Ln = x.shape
    out = torch.empty((B, 2 * C), dtype=torch.float32, device=x.device)
    L.check(L.load().pcr_pool_both_f32(L.ptr(x), L.ptr(out), B, C, Ln, L.stream_ptr()), "pcr_pool_both_f32")
    return out


def pool_channel_max(x, window):
    """(B,C,L) -> (B,L) (or (B,L,C//window) when C > window): ReIDNet.get_pooled_feats with pool_type='max', i.e.
    nn.MaxPool1d(window) over the channels of the permuted tensor followed by squeeze(-1) (ReIDNet.py:145,526-528)"""
    L.require_cuda(x)
    x = x.contiguous()
    B, C, Ln = x.shape
    if window > C:
        raise L.PcrError("MaxPool1d(%d) over %d channels has no output" % (window, C))
    G = C // window
    out = torch.empty((B, Ln, G), dtype=torch.float32, device=x.device)
    L.check(L.load().pcr_channel_max_f32(L.ptr(x), L.ptr(out), B, C, Ln, window, L.stream_ptr()), "pcr_channel_max_f32")
    return out.squeeze(-1)


# ---- LinearRes on channel-major token tensors --------------------------------------------------
from . import engine as _E


def groupnorm(x, gn, res=None, relu=False):
    """x (B,C,L): GroupNorm over channel groups per token [+ res] [relu]"""
    L.require_cuda(x)
    L.require_default_eps(gn)
    x = x.contiguous()
    B, C, Ln = x.shape
    y = torch.empty_like(x)
    g = gn.weight.detach().to(x.device).float().contiguous()
    b = gn.bias.detach().to(x.device).float().contiguous()
    L.check(L.load().pcr_groupnorm_f32(L.ptr(x), L.ptr(g), L.ptr(b), L.ptr(res), L.ptr(y), B, C, Ln, gn.num_groups,
                                       1 if relu else 0, L.stream_ptr()), "pcr_groupnorm_f32")
    return y


class _LinResPlan:
    def __init__(self, m, device):
        self.w1 = _E.pack_weight_dual(m.linear1.weight, device)
        self.w2 = _E.pack_weight_dual(m.linear2.weight, device)
        self.wt = _E.pack_weight_dual(m.transform[0].weight, device) if m.transform is not None else None
        self.n_out = m.linear1.weight.shape[0]


def _linres_plan(m, device):
    key = (str(device), _E.param_version(m))
    if getattr(m, "_pcr_key", None) != key:
        object.__setattr__(m, "_pcr_plan", _LinResPlan(m, device))
        object.__setattr__(m, "_pcr_key", key)
    return m._pcr_plan


def dense_gn(x, wp, cout, gn, res=None, relu=False):
    """[relu](GroupNorm(W x) [+ res]) on a channel-major tensor: one launch (pcr_dense_gn_f32) when the group size is
    4 / 8 / 16 / 32 channels, pcr_dense_f32 + pcr_groupnorm_f32 otherwise"""
    L.require_cuda(x)
    L.require_default_eps(gn)
    if cout // gn.num_groups not in (4, 8, 16, 32):
        return groupnorm(_E.dense(x, wp, cout), gn, res=res, relu=relu)
    x = x.contiguous()
    B, cin, Ln = x.shape
    y = torch.empty((B, cout, Ln), dtype=torch.float32, device=x.device)
    g = gn.weight.detach().to(x.device).float().contiguous()
    b = gn.bias.detach().to(x.device).float().contiguous()
    if res is not None:
        res = res.contiguous()
    bf = getattr(wp, "_pcr_bf", None)
    if _E.PRECISION != "f32" and bf is not None and L.load().pcr_dense_prec_ok(cin, cout, Ln):
        with _E._prof("dense_gn[cin=%d,cout=%d,L=%d]" % (cin, cout, Ln), 2.0 * B * Ln * cin * cout,
                      4.0 * B * Ln * (cin + cout * (2 if res is not None else 1)), arith=_E.PRECISION):
            L.check(L.load().pcr_dense_gn_prec_f32(L.ptr(x), L.ptr(bf), L.ptr(g), L.ptr(b), L.ptr(res), L.ptr(y), B, cin,
                                                   cout, Ln, gn.num_groups, 1 if relu else 0,
                                                   _E.PRECISIONS[_E.PRECISION], L.stream_ptr()), "pcr_dense_gn_prec_f32")
        return y
    with _E._prof("dense_gn[cin=%d,cout=%d,L=%d]" % (cin, cout, Ln), 2.0 * B * Ln * cin * cout,
                  4.0 * B * Ln * (cin + cout * (2 if res is not None else 1)), arith="f32"):
        L.check(L.load().pcr_dense_gn_f32(L.ptr(x), L.ptr(wp), L.ptr(g), L.ptr(b), L.ptr(res), L.ptr(y), B, cin, cout,
                                          Ln, gn.num_groups, 1 if relu else 0, L.stream_ptr()), "pcr_dense_gn_f32")
    return y


def linear_res_cm(m, x):
    """LinearRes on a channel-major tensor x (B,n_in,L) -> (B,n_out,L) (lanegcn_nets.py:228-241)"""
    p = _linres_plan(m, x.device)
    out = dense_gn(x, p.w1, p.n_out, m.norm1, relu=True)
    short = dense_gn(x, p.wt, p.n_out, m.transform[1]) if p.wt is not None else x
    return dense_gn(out, p.w2, p.n_out, m.norm2, res=short, relu=True)


def linear_res(m, x):
    """module-style call on (M,n_in) rows"""
    L.require_cuda(x)
    return linear_res_cm(m, x.t().contiguous().unsqueeze(0)).squeeze(0).t().contiguous()


def downsample_points(seq, h):
    """ReIDNet.downsample ([LinearRes, LinearRes, Linear]) applied per point to h (B,C,N) -> (B,C',N)
    (reference: ReIDNet.siamese_forward :316-324 reshapes to (B*N,C) rows; per-token GroupNorm makes the
    channel-major evaluation identical)"""
    from mmdet3d.models.lanegcn_nets import LinearRes
    x = h.contiguous()
    for m in seq:
        if isinstance(m, LinearRes):
            x = linear_res_cm(m, x)
        elif isinstance(m, torch.nn.Linear):
            key = (str(x.device), _E.param_version(m))
            if getattr(m, "_pcr_key", None) != key:
                object.__setattr__(m, "_pcr_w", _E.pack_weight(m.weight, x.device))
                object.__setattr__(m, "_pcr_b", None if m.bias is None else m.bias.detach().to(x.device).float().contiguous())
                object.__setattr__(m, "_pcr_key", key)
            x = _E.dense(x, m._pcr_w, m.weight.shape[0], None, m._pcr_b, 0)
        else:
            raise L.PcrError("downsample stack may contain LinearRes and Linear only (got %s)" % type(m).__name__)
    return x
