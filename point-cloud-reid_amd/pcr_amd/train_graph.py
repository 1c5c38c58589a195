"""Training-mode forward of the Point-Transformer ReID path as a differentiable graph on the device.

STATUS: inference is hand-written HIP end to end; training is NOT yet.  This module
makes `ReIDNet.train_step` usable today: neighbour search runs on the HIP kernel (pcr_knn_prefix_f32,
indices carry no gradient), neighbour gathers run on the HIP grouping op with its HIP scatter-add
backward (pcr_group_fwd/bwd_f32), and the dense math (1x1 convs with BatchNorm in batch-statistics mode,
linear attention, LayerNorm/GroupNorm, BCE) is expressed with torch autograd ops on the GPU, following the
reference's graph (mmdet3d/models/pointnet2_utils.py:55-114, 242-288, 333-437; attention.py:192-219;
lanegcn_nets.py:228-241).  Gradients are pinned to the reference by tests/golden/pt_train_step_n128.npz.
Fused forward/backward kernels for these layers are the next training milestone; nothing here is used in
eval mode.
"""
import torch
import torch.nn.functional as F

from . import engine
from mmdet3d.ops.point_ops import grouping_operation


def linear_attention(q, k, v, eps=1e-6):
    Q = F.elu(q) + 1
    K = F.elu(k) + 1
    s = v.size(1)
    v = v / s
    KV = torch.einsum("nshd,nshv->nhdv", K, v)
    Z = 1 / (torch.einsum("nlhd,nhd->nlh", Q, K.sum(dim=1)) + eps)
    return torch.einsum("nlhd,nhdv,nlh->nlhv", Q, KV, Z) * s


def attention_block(m, q_in, k_in, v_in, res_in, residual):
    """m: module with q_proj/k_proj/v_proj/merge/mlp/norm1/norm2; inputs (B,L,C) token-major"""
    B, L, _ = q_in.shape
    d, h = m.q_proj.weight.shape[0], m.nhead
    q = m.q_proj(q_in).view(B, L, h, d // h)
    k = m.k_proj(k_in).view(B, -1, h, d // h)
    v = m.v_proj(v_in).view(B, -1, h, d // h)
    msg = m.norm1(m.merge(linear_attention(q, k, v).reshape(B, L, d)))
    msg = m.norm2(m.mlp(torch.cat([res_in, msg], dim=2)))
    return res_in + msg if residual else msg


def self_attention(m, feat, xyz):
    f = feat.permute(0, 2, 1)
    fp = f + m.pos_mlp(xyz)
    return attention_block(m, fp, fp, fp, f, True).permute(0, 2, 1)


def fp_sa(m, feat1, xyz1, feat2, xyz2):
    f1, f2 = feat1.permute(0, 2, 1), feat2.permute(0, 2, 1)
    return attention_block(m, f1, f2, f2 + m.pos_mlp2(xyz2), f1, False).permute(0, 2, 1)


def cross_attention(m, search, search_xyz, template, template_xyz):
    s, t = search.permute(0, 2, 1), template.permute(0, 2, 1)
    return attention_block(m, s, t, t + m.pos_mlp(template_xyz), s, True).permute(0, 2, 1)


def sa_edge_layer(sa, xyz, feats, s):
    """PointNetSetAbstractionEdgeSA in training mode: (B,N,3), (B,D,N)|None -> (B,S,3), (B,D',S).  Neighbour search,
    the per-point tables of layer 1, the three conv + BatchNorm(batch statistics) + ReLU layers, the max over K and
    all of their backward are HIP launches (pcr_amd/train_ops.py: SaEdgeTrain)."""
    from . import train_ops
    xyz = xyz.contiguous()
    idx = engine.knn_prefix(xyz.detach(), s, sa.nsample)                      # HIP, (B,S,K) int32
    new_xyz = xyz[:, :s]
    x = train_ops.sa_edge_train(sa, xyz.detach(), None if feats is None else feats.contiguous(), idx)
    return new_xyz, self_attention(sa.self_attention, x, new_xyz)


def backbone(bb, pointcloud, numpoints):
    xyz = pointcloud[..., 0:3].contiguous()
    l_xyz, l_feat = [xyz], [None]
    for i, sa in enumerate(bb.SA_modules):
        nx, nf = sa_edge_layer(sa, l_xyz[i], l_feat[i], numpoints[i])
        l_xyz.append(nx)
        l_feat.append(nf)
    l_feat[0] = xyz.transpose(1, 2).contiguous()
    for i in (2, 1, 0):
        l_feat[i] = fp_sa(bb.FP_modules[i].interpolation, l_feat[i], l_xyz[i], l_feat[i + 1], l_xyz[i + 1])
    return xyz, bb.cov_final(l_feat[0])


def linear_res(m, x):
    out = F.relu(m.norm1(m.linear1(x)))
    out = m.norm2(m.linear2(out))
    out = out + (m.transform(x) if m.transform is not None else x)
    return F.relu(out)


def match_logits(model, h1, xyz1, h2, xyz2):
    a1 = cross_attention(model.cross_stage1, h1, xyz1, h2, xyz2)
    a2 = cross_attention(model.cross_stage1, h2, xyz2, h1, xyz1)
    o1 = cross_attention(model.cross_stage2, a1, xyz1, a2, xyz2)
    o2 = cross_attention(model.cross_stage2, a2, xyz2, a1, xyz1)
    x = torch.cat([o1, o2], dim=2)
    pooled = torch.cat([x.max(dim=2)[0], x.mean(dim=2)], dim=1)
    x = linear_res(model.match_head[0], pooled)
    return model.match_head[1](x).squeeze(1), torch.cat([o1, o2], dim=0)
