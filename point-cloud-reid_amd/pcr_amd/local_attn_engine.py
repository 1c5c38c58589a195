"""local_self_attention forward (reference: mmdet3d/models/attention.py:262-296) as libpcr_hip.so launches:
pcr_knn_feat_f32 (feature-space kNN, k = knum) -> pcr_dense_f32 (first layer of the position MLP) ->
pcr_dense_pm_f32 (q | k | v of feat + pos(xyz) for every POINT, point-major: the reference projects the gathered
(B*N, K, C) neighbour tensor, i.e. every point's key / value K times) -> pcr_local_attn_f32 (one query token per
point over its K neighbours) -> merge + LayerNorm, feed-forward on cat[feat, msg], LayerNorm + residual through the
dense / group-norm kernels (LayerNorm over C channels per token = GroupNorm with one group)."""
import ctypes
import types

import torch

from . import _lib as L
from . import engine as E
from . import rows
from .dgcnn_engine import knn_feat


class _Plan:
    def __init__(self, m, device):
        d = m.d_model
        w1, b1 = m.pos_mlp_knn[0].weight.detach().double().cpu(), m.pos_mlp_knn[0].bias.detach().double().cpu()
        w2, b2 = m.pos_mlp_knn[2].weight.detach().double().cpu(), m.pos_mlp_knn[2].bias.detach().double().cpu()
        if w2.shape[0] != d:
            raise L.PcrError("local_self_attention: pos_size must equal d_model (the position code is ADDED to the "
                             "features, attention.py:281-282)")
        self.w1 = E.pack_weight(w1.float(), device)
        self.b1 = E._dev32(b1.float(), device)
        self.hid = w1.shape[0]
        # [q | k | v] of (feat + W2 h + b2) = [W, W W2, W b2] . [feat ; h ; 1]
        blocks = []
        for proj in (m.q_proj_knn, m.k_proj_knn, m.v_proj_knn):
            w = proj.weight.detach().double().cpu()
            blocks.append(torch.cat([w, w @ w2, (w @ b2).unsqueeze(1)], dim=1))
        self.wqkv = E.pack_weight(torch.cat(blocks, dim=0).float(), device)
        self.wm = E.pack_weight(m.merge_knn.weight, device)
        self.wf0 = E.pack_weight(m.mlp_knn[0].weight, device)
        self.wf2 = E.pack_weight(m.mlp_knn[2].weight, device)
        self.ln1 = types.SimpleNamespace(num_groups=1, weight=m.norm1_knn.weight, bias=m.norm1_knn.bias,
                                         eps=m.norm1_knn.eps)
        self.ln2 = types.SimpleNamespace(num_groups=1, weight=m.norm2_knn.weight, bias=m.norm2_knn.bias,
                                         eps=m.norm2_knn.eps)


def forward(m, feat, xyz):
    L.require_cuda(feat, xyz)
    if m.training:
        raise L.PcrError("local_self_attention: the HIP path implements eval-mode inference; call .eval()")
    key = (str(feat.device), E.param_version(m))
    if getattr(m, "_pcr_key", None) != key:
        object.__setattr__(m, "_pcr_plan", _Plan(m, feat.device))
        object.__setattr__(m, "_pcr_key", key)
    p = m._pcr_plan
    feat = feat.contiguous().float()
    B, C, N = feat.shape
    if C != m.d_model or C > 64:
        raise L.PcrError("local_self_attention: d_model=%d features expected (<= 64), got %d" % (m.d_model, C))
    if m.knum > N:
        raise L.PcrError("local_self_attention: knum=%d neighbours need at least that many points (N=%d)" % (m.knum, N))
    idx = knn_feat(feat, m.knum)
    h = E.dense(xyz.contiguous().float().transpose(1, 2), p.w1, p.hid, None, p.b1, act=1)        # (B,hid,N)
    x = torch.cat([feat, h, torch.ones((B, 1, N), dtype=torch.float32, device=feat.device)], dim=1)
    qkv = torch.empty((B, N, 3 * C), dtype=torch.float32, device=feat.device)
    lib = L.load()
    with E._prof("local_qkv", 2.0 * B * N * x.shape[1] * 3 * C, 4.0 * B * N * (x.shape[1] + 3 * C)):
        L.check(lib.pcr_dense_pm_f32(L.ptr(x), L.ptr(p.wqkv), L.ptr(qkv), B, x.shape[1], 3 * C, N, 0, L.stream_ptr()),
                "pcr_dense_pm_f32")
    msg = torch.empty((B, C, N), dtype=torch.float32, device=feat.device)
    with E._prof("local_attn", 4.0 * B * N * m.knum * C, 4.0 * B * N * (m.knum * (2 * C + 1) + 2 * C)):
        L.check(lib.pcr_local_attn_f32(L.ptr(qkv), L.ptr(idx), L.ptr(msg), B, N, C, m.knum, m.nhead,
                                       ctypes.c_float(1e-6), L.stream_ptr()), "pcr_local_attn_f32")
    m1 = rows.dense_gn(msg, p.wm, C, p.ln1)
    f = E.dense(torch.cat([feat, m1], dim=1), p.wf0, 2 * C, act=1)
    return rows.dense_gn(f, p.wf2, C, p.ln2, res=feat)
