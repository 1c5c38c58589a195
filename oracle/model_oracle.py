"""TEST INFRASTRUCTURE ONLY -- PyTorch-eager CPU restatement of the reference's siamese
ReID model path (functional, driven by a state_dict).

It replays the same ATen op sequence as the reference (expanded-matmul distances + argsort
kNN, unfused Conv/BN/ReLU/max, einsum linear attention) so that (a) its outputs are pinned to
the reference by the golden fixtures in tests/golden (tests/test_oracle_model.py), and (b) it
can stand in for the reference on the GPU box, where /root/reference does not exist: as the
checker for the HIP path and as the timed "port" CPU baseline of bench.py.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.

Reference citations (bentherien/point-cloud-reid):
  mmdet3d/models/pointnet2_utils.py  LinearAttention :14-47, Self_Attention :55-114,
      random_point_sample :139-149, square_distance :169-188, topk/knn_point :190-216,
      sample_and_group_edge :242-288, PointNetSetAbstractionEdgeSA :309-360, FP_SA :362-437
  mmdet3d/models/backbone_net.py     Pointnet_Backbone.forward :96-124
  mmdet3d/models/attention.py        corss_attention :157-219
  mmdet3d/models/pointnet.py         STN3d :10-45, STNkd :48-85, PointNetEncoder :88-127
  mmdet3d/models/lanegcn_nets.py     LinearRes :193-241
  mmdet3d/models/ReIDNet.py          siamese_forward :311-332, xcorr_eff :231-247,
      get_pooled_feats :526-534, match_forward_inference :444-462
"""
import math

import torch
import torch.nn.functional as F

BN_EPS = 1e-5
LN_EPS = 1e-5
ATTN_EPS = 1e-6


def _sub(sd, prefix):
    n = len(prefix)
    return {k[n:]: v for k, v in sd.items() if k.startswith(prefix)}


def _bn(x, p, name, dim):
    shape = [1, -1] + [1] * (dim - 2)
    inv = torch.rsqrt(p[name + ".running_var"] + BN_EPS)
    return F.batch_norm(x, p[name + ".running_mean"], p[name + ".running_var"],
                        p[name + ".weight"], p[name + ".bias"], False, 0.0, BN_EPS)


def _ln(x, p, name):
    return F.layer_norm(x, (x.shape[-1],), p[name + ".weight"], p[name + ".bias"], LN_EPS)


def _lin(x, p, name):
    return F.linear(x, p[name + ".weight"], p.get(name + ".bias"))


def linear_attention(q, k, v):
    """q [B,L,H,D], k,v [B,S,H,D]  (pointnet2_utils.py:20-47)"""
    Q = F.elu(q) + 1
    K = F.elu(k) + 1
    s = v.size(1)
    v = v / s
    KV = torch.einsum("nshd,nshv->nhdv", K, v)
    Z = 1 / (torch.einsum("nlhd,nhd->nlh", Q, K.sum(dim=1)) + ATTN_EPS)
    return torch.einsum("nlhd,nhdv,nlh->nlhv", Q, KV, Z) * s


def attention_block(p, q_in, k_in, v_in, res_in, nhead, residual, pos_name=None):
    """Shared tail of Self_Attention / FP_SA / corss_attention.
    q_in [B,L,C1], k_in,v_in [B,S,C2] (already position-encoded where the block does so),
    res_in [B,L,C1] is what is concatenated in front of the message for the feed-forward."""
    B, L, _ = q_in.shape
    d = p["q_proj.weight"].shape[0]
    q = _lin(q_in, p, "q_proj").view(B, L, nhead, d // nhead)
    k = _lin(k_in, p, "k_proj").view(B, -1, nhead, d // nhead)
    v = _lin(v_in, p, "v_proj").view(B, -1, nhead, d // nhead)
    msg = linear_attention(q, k, v).reshape(B, L, d)
    msg = _ln(_lin(msg, p, "merge"), p, "norm1")
    msg = F.linear(torch.cat([res_in, msg], dim=2), p["mlp.0.weight"])
    msg = F.linear(F.relu(msg), p["mlp.2.weight"])
    msg = _ln(msg, p, "norm2")
    return res_in + msg if residual else msg


def _pos_mlp(p, name, xyz):
    return _lin(F.relu(_lin(xyz, p, name + ".0")), p, name + ".2")


def self_attention(p, feat, xyz, nhead=2):
    """feat [B,C,N], xyz [B,N,3] -> [B,C,N]   (pointnet2_utils.py:90-114)"""
    f = feat.permute(0, 2, 1)
    fp = f + _pos_mlp(p, "pos_mlp", xyz)
    return attention_block(p, fp, fp, fp, f, nhead, True).permute(0, 2, 1)


def fp_sa(p, feat1, xyz1, feat2, xyz2, nhead=2):
    """fine <- coarse cross attention, no residual (pointnet2_utils.py:407-437)"""
    f1 = feat1.permute(0, 2, 1)
    f2 = feat2.permute(0, 2, 1)
    f2p = f2 + _pos_mlp(p, "pos_mlp2", xyz2)
    return attention_block(p, f1, f2, f2p, f1, nhead, False).permute(0, 2, 1)


def cross_attention(p, search, search_xyz, template, template_xyz, nhead=2):
    """attention.py:192-219 (search_xyz unused there as well)"""
    s = search.permute(0, 2, 1)
    t = template.permute(0, 2, 1)
    tp = t + _pos_mlp(p, "pos_mlp", template_xyz)
    return attention_block(p, s, t, tp, s, nhead, True).permute(0, 2, 1)


def local_self_attention(p, feat, xyz, nhead=2, knum=48, knn_fn=None):
    """attention.py:262-296: every point attends (linear attention, one query token) to its knum feature-space
    neighbours; parameter names carry the reference's `_knn` suffix.  feat (B,C,N), xyz (B,N,3) -> (B,C,N)."""
    knn_fn = knn_fn or knn_feat_oracle
    B, C, N = feat.shape
    idx = knn_fn(feat, knum)                                            # (B,N,K)
    ft = feat.permute(0, 2, 1)                                          # (B,N,C)
    gi = idx.reshape(B, N * knum)
    fea_knn = torch.gather(ft, 1, gi.unsqueeze(-1).expand(B, N * knum, C)).reshape(B * N, knum, C)
    xyz_knn = torch.gather(xyz, 1, gi.unsqueeze(-1).expand(B, N * knum, 3)).reshape(B * N, knum, 3)
    sf = ft.reshape(B * N, 1, C)
    q_in = sf + _pos_mlp(p, "pos_mlp_knn", xyz.reshape(B * N, 1, 3))
    kv_in = fea_knn + _pos_mlp(p, "pos_mlp_knn", xyz_knn)
    d = C // nhead
    q = F.linear(q_in, p["q_proj_knn.weight"]).view(B * N, 1, nhead, d)
    k = F.linear(kv_in, p["k_proj_knn.weight"]).view(B * N, knum, nhead, d)
    v = F.linear(kv_in, p["v_proj_knn.weight"]).view(B * N, knum, nhead, d)
    msg = linear_attention(q, k, v).reshape(B * N, 1, C)
    msg = F.layer_norm(F.linear(msg, p["merge_knn.weight"]), (C,), p["norm1_knn.weight"], p["norm1_knn.bias"], LN_EPS)
    msg = F.linear(torch.cat([sf, msg], dim=2), p["mlp_knn.0.weight"])
    msg = F.linear(F.relu(msg), p["mlp_knn.2.weight"])
    msg = F.layer_norm(msg, (C,), p["norm2_knn.weight"], p["norm2_knn.bias"], LN_EPS)
    return (sf + msg).view(B, N, C).permute(0, 2, 1)


def match_xcorr(sd, h1, xyz1, h2, xyz2, knum=48, stages=None, knn_fn=None, head_ng=8):
    """match_type='xcorr' (ReIDNet.py:250-256, 445-449): cross -> local -> cross -> local on the search branch only,
    pool 'both', match head"""
    a = cross_attention(_sub(sd, "cross_stage1."), h1, xyz1, h2, xyz2)
    b = local_self_attention(_sub(sd, "local_stage1."), a, xyz1, 2, knum, knn_fn)
    c = cross_attention(_sub(sd, "cross_stage2."), b, xyz1, h2, xyz2)
    d = local_self_attention(_sub(sd, "local_stage2."), c, xyz1, 2, knum, knn_fn)
    pooled = pool_both(d)
    x = linear_res(_linres_params(sd, "match_head.0.", head_ng), pooled)
    logits = F.linear(x, sd["match_head.1.weight"], sd["match_head.1.bias"]).squeeze(1)
    if stages is not None:
        stages.update(xc_a=a, xc_b=b, xc_c=c, xc_d=d, pooled=pooled, logits=logits)
    return logits


def pt_pairs_xcorr(sd, s1, s2, backbone_list, nsample=(32, 48, 48), knum=48, stages=None, knn_fn=None):
    """Point-Transformer backbone + the 'baseline-orig' matching (reid_pts_point-transformer_baseline_orig.py)"""
    b = s1.shape[0]
    xyz, h = pt_backbone(_sub(sd, "backbone."), torch.cat([s1, s2], 0), backbone_list, nsample, stages)
    return match_xcorr(sd, h[:b], xyz[:b], h[b:], xyz[b:], knum, stages, knn_fn)


def square_distance(src, dst):
    d = -2 * torch.matmul(src, dst.permute(0, 2, 1))
    d += torch.sum(src ** 2, -1).unsqueeze(-1)
    d += torch.sum(dst ** 2, -1).unsqueeze(1)
    return d


def farthest_point_sample_py(xyz, npoint, start):
    """farthest_point_sample (pointnet2_utils.py:116-137) with the reference's torch.randint start replaced by `start`"""
    B, N, _ = xyz.shape
    centroids = torch.zeros(B, npoint, dtype=torch.long)
    distance = torch.ones(B, N) * 1e10
    farthest = start.long().clone()
    batch = torch.arange(B)
    for i in range(npoint):
        centroids[:, i] = farthest
        centroid = xyz[batch, farthest, :].view(B, 1, 3)
        dist = torch.sum((xyz - centroid) ** 2, -1)
        mask = dist < distance
        distance[mask] = dist[mask]
        farthest = torch.max(distance, -1)[1]
    return centroids


def query_ball_point_py(radius, nsample, xyz, new_xyz, return_dist=False):
    """query_ball_point (pointnet2_utils.py:218-240)"""
    B, N, _ = xyz.shape
    S = new_xyz.shape[1]
    group_idx = torch.arange(N).view(1, 1, N).repeat(B, S, 1)
    sqrdists = square_distance(new_xyz, xyz)
    group_idx[sqrdists > radius ** 2] = N
    group_idx = group_idx.sort(dim=-1)[0][:, :, :nsample]
    first = group_idx[:, :, 0].view(B, S, 1).repeat(1, 1, nsample)
    mask = group_idx == N
    group_idx[mask] = first[mask]
    return (group_idx, sqrdists) if return_dist else group_idx


def knn_prefix(xyz, s, k):
    """first-s centres, k nearest of all N by full argsort (pointnet2_utils.py:190-216)"""
    d = square_distance(xyz[:, :s], xyz)
    return torch.argsort(d, dim=-1)[:, :, :k]


def _gather_rows(points, idx):
    """points [B,N,C], idx [B,S,K] -> [B,S,K,C]"""
    B, S, K = idx.shape
    C = points.shape[-1]
    flat = idx.reshape(B, S * K, 1).expand(-1, -1, C)
    return torch.gather(points, 1, flat).view(B, S, K, C)


def sa_edge_layer(p, xyz, feats, s, k, stages=None, tag="", centre_idx=None, group_idx=None):
    """One PointNetSetAbstractionEdgeSA (pointnet2_utils.py:333-360) with RANDOM (prefix)
    sampling and kNN grouping (or, for the dormant branches of sample_and_group_edge :262-272, the given centre
    indices [B,S] and group indices [B,S,K]).  xyz [B,N,3]; feats [B,D,N] or None -> (new_xyz, [B,D',S])"""
    idx = knn_prefix(xyz, s, k) if group_idx is None else group_idx
    if centre_idx is None:
        centre_idx = torch.arange(s).repeat(xyz.shape[0], 1)
    new_xyz = torch.gather(xyz, 1, centre_idx.unsqueeze(-1).expand(-1, -1, 3))
    g = _gather_rows(xyz, idx) - new_xyz.unsqueeze(2)
    if feats is not None:
        pts = feats.permute(0, 2, 1)
        centre = torch.gather(pts, 1, centre_idx.unsqueeze(-1).expand(-1, -1, pts.shape[-1])).unsqueeze(2)
        nb = _gather_rows(pts, idx)
        g = torch.cat([g, centre.expand(-1, -1, k, -1), nb - centre], dim=-1)
    x = g.permute(0, 3, 1, 2)
    for i in range(3):
        x = F.conv2d(x, p[f"mlp_convs.{i}.weight"], p[f"mlp_convs.{i}.bias"])
        x = F.relu(_bn(x, p, f"mlp_bns.{i}", 4))
    x = x.max(dim=3)[0]
    if stages is not None:
        stages[tag + "_knn_sorted"] = torch.sort(idx, dim=-1)[0]
        stages[tag + "_mlp"] = x
    out = self_attention(_sub(p, "self_attention."), x, new_xyz)
    if stages is not None:
        stages[tag + "_out"] = out
    return new_xyz, out


def pt_backbone(p, pc, backbone_list, nsample=(32, 48, 48), stages=None):
    """Pointnet_Backbone.forward (backbone_net.py:96-124). pc [B,N,3] -> (xyz, h [B,conv_out,N])"""
    xyz = pc[..., :3].contiguous()
    l_xyz, l_feat = [xyz], [None]
    for i in range(3):
        nx, nf = sa_edge_layer(_sub(p, f"SA_modules.{i}."), l_xyz[i], l_feat[i],
                               backbone_list[i], nsample[i], stages, f"sa{i}")
        l_xyz.append(nx)
        l_feat.append(nf)
    l_feat[0] = xyz.transpose(1, 2).contiguous()
    for i in (2, 1, 0):
        l_feat[i] = fp_sa(_sub(p, f"FP_modules.{i}.interpolation."), l_feat[i], l_xyz[i],
                          l_feat[i + 1], l_xyz[i + 1])
        if stages is not None:
            stages[f"fp{i}_out"] = l_feat[i]
    h = F.conv1d(l_feat[0], p["cov_final.weight"], p["cov_final.bias"])
    return xyz, h


def linear_res(p, x):
    """LinearRes (lanegcn_nets.py:228-241); GroupNorm group count = gcd(ng, n_out) is implied
    by the caller through p['__groups__']."""
    g = p["__groups__"]
    out = F.relu(F.group_norm(F.linear(x, p["linear1.weight"]), g, p["norm1.weight"], p["norm1.bias"]))
    out = F.group_norm(F.linear(out, p["linear2.weight"]), g, p["norm2.weight"], p["norm2.bias"])
    if "transform.0.weight" in p:
        out = out + F.group_norm(F.linear(x, p["transform.0.weight"]), g,
                                 p["transform.1.weight"], p["transform.1.bias"])
    else:
        out = out + x
    return F.relu(out)


def _linres_params(sd, prefix, ng):
    p = _sub(sd, prefix)
    p["__groups__"] = math.gcd(ng, p["linear1.weight"].shape[0])
    return p


def _stn(p, x, k):
    """STN3d / STNkd (pointnet.py:27-45, :67-85).  x [B,k,N] -> [B,k,k]"""
    B = x.shape[0]
    for i, n in ((1, "conv1"), (2, "conv2"), (3, "conv3")):
        x = F.relu(_bn(F.conv1d(x, p[n + ".weight"], p[n + ".bias"]), p, f"bn{i}", 3))
    x = x.max(dim=2)[0]
    x = F.relu(_bn(_lin(x, p, "fc1"), p, "bn4", 2))
    x = F.relu(_bn(_lin(x, p, "fc2"), p, "bn5", 2))
    x = _lin(x, p, "fc3") + torch.eye(k, dtype=x.dtype).flatten().unsqueeze(0)
    return x.view(B, k, k)


def pointnet_encoder(p, x):
    """PointNetEncoder.forward with feature_transform=True, channel=3 (pointnet.py:103-127).
    x [B,3,N] -> (xyz [B,3,N], feat [B,1024,N])"""
    xyz = x
    trans = _stn(_sub(p, "stn."), x, 3)
    x = torch.bmm(x.transpose(2, 1), trans).transpose(2, 1)
    x = F.relu(_bn(F.conv1d(x, p["conv1.weight"], p["conv1.bias"]), p, "bn1", 3))
    tf = _stn(_sub(p, "fstn."), x, 64)
    x = torch.bmm(x.transpose(2, 1), tf).transpose(2, 1)
    x = F.relu(_bn(F.conv1d(x, p["conv2.weight"], p["conv2.bias"]), p, "bn2", 3))
    x = _bn(F.conv1d(x, p["conv3.weight"], p["conv3.bias"]), p, "bn3", 3)
    return xyz, x


def pool_both(x):
    """get_pooled_feats, pool_type='both' (ReIDNet.py:529-532). x [B,C,L] -> [B,2C]"""
    return torch.cat([x.max(dim=2)[0], x.mean(dim=2)], dim=1)


def pool_channel_max(x, window):
    """get_pooled_feats, pool_type='max' (ReIDNet.py:145,526-528): nn.MaxPool1d(window) on the permuted [B,N,C]
    tensor, i.e. a max over windows of `window` channels of every point.  x [B,C,N] -> [B,N] when C == window"""
    return F.max_pool1d(x.permute(0, 2, 1), window).squeeze(-1)


def pt_pairs_concat(sd, s1, s2, backbone_list, window=64, head_ng=32, stages=None):
    """reid_pts_point-transformer_baseline.py: match_type='concat' + pool_type='max' (ReIDNet.py:415-419, 455-457)"""
    b = s1.shape[0]
    xyz, h = pt_backbone(_sub(sd, "backbone."), torch.cat([s1, s2], 0), backbone_list)
    cat = torch.cat([pool_channel_max(h[:b], window), pool_channel_max(h[b:], window)], dim=1)
    x = linear_res(_linres_params(sd, "match_head.0.", head_ng), cat)
    logits = F.linear(x, sd["match_head.1.weight"], sd["match_head.1.bias"]).squeeze(1)
    if stages is not None:
        stages.update(h1=h[:b], h2=h[b:], pooled1=cat[:, :cat.shape[1] // 2], logits=logits)
    return logits


def match(sd, h1, xyz1, h2, xyz2, stages=None, head_ng=8):
    """xcorr_eff + point-cat + pool 'both' + match head (ReIDNet.py:231-247, 444-462); head_ng = the match head's
    GroupNorm `ng` (8 in the Point-Transformer / PointNet configs, 16 in reid_pts_dgcnn_point-cat.py:33)"""
    c1, c2 = _sub(sd, "cross_stage1."), _sub(sd, "cross_stage2.")
    a1 = cross_attention(c1, h1, xyz1, h2, xyz2)
    a2 = cross_attention(c1, h2, xyz2, h1, xyz1)
    o1 = cross_attention(c2, a1, xyz1, a2, xyz2)
    o2 = cross_attention(c2, a2, xyz2, a1, xyz1)
    pooled = pool_both(torch.cat([o1, o2], dim=2))
    x = linear_res(_linres_params(sd, "match_head.0.", head_ng), pooled)
    logits = F.linear(x, sd["match_head.1.weight"], sd["match_head.1.bias"]).squeeze(1)
    if stages is not None:
        stages.update(x1_o1=a1, x1_o2=a2, x2_o1=o1, x2_o2=o2, pooled=pooled, logits=logits)
    return logits


def pt_pairs(sd, s1, s2, backbone_list, nsample=(32, 48, 48), stages=None):
    """Point-Transformer ReIDNet: siamese_forward + match_forward_inference."""
    b = s1.shape[0]
    xyz, h = pt_backbone(_sub(sd, "backbone."), torch.cat([s1, s2], 0), backbone_list, nsample, stages)
    if stages is not None:
        stages.update(h1=h[:b], h2=h[b:])
    return match(sd, h[:b], xyz[:b], h[b:], xyz[b:], stages)


def pointnet_pairs(sd, s1, s2, stages=None):
    """PointNet ReIDNet (use_dgcnn=True branch of siamese_forward, ReIDNet.py:316-324)."""
    b, n, _ = s1.shape
    x = torch.cat([s1, s2], 0).permute(0, 2, 1)
    xyz, f = pointnet_encoder(_sub(sd, "backbone.feat."), x)
    t = f.permute(0, 2, 1).reshape(-1, f.shape[1])
    t = linear_res(_linres_params(sd, "downsample.0.", 64), t)
    t = linear_res(_linres_params(sd, "downsample.1.", 16), t)
    t = F.linear(t, sd["downsample.2.weight"], sd["downsample.2.bias"])
    h = t.reshape(2 * b, n, -1).permute(0, 2, 1)
    xyz = xyz.permute(0, 2, 1)
    if stages is not None:
        stages.update(h1=h[:b], h2=h[b:], enc_max=f.max(dim=2)[0], enc_mean=f.mean(dim=2))
    return match(sd, h[:b], xyz[:b], h[b:], xyz[b:], stages)


# ---- DGCNN backbone (mmdet3d/models/dgcnn_orig.py: knn :22-29, get_graph_feature :32-56, DGCNN :89-152) ------
# The graph feature is materialised exactly as the reference does: cat[f_j - f_i, f_i] -> 1x1 Conv2d (no bias) ->
# BatchNorm2d -> LeakyReLU(0.2) -> max over the k neighbours; x1..x4 concatenated -> Conv1d 512 -> emb_dims + BN +
# LeakyReLU.  bn{i} and conv{i}.1 are ONE module registered twice (dgcnn_orig.py:94-104): a state_dict carries both
# names, load_state_dict writes conv{i}.1 last, so that is the name read here.  The neighbour SETS come from oracle/pcr_oracle.c:pcr_oracle_knn_feat, which fixes the summation order
# the reference leaves to its BLAS (`knn_fn` lets the tests substitute the reference's own torch formula).
def knn_feat_torch(x, k):
    """the reference's formula verbatim in torch (order of summation = whatever ATen picks)"""
    inner = -2 * torch.matmul(x.transpose(2, 1), x)
    xx = torch.sum(x ** 2, dim=1, keepdim=True)
    pd = -xx - inner - xx.transpose(2, 1)
    return pd.topk(k=k, dim=-1)[1]


def knn_feat_oracle(x, k):
    import point_ops
    return torch.from_numpy(point_ops.knn_feat(x.detach().cpu().numpy(), k)).long()


def graph_feature(x, idx):
    """x (B,C,N), idx (B,N,k) -> (B,2C,N,k) = cat[f_j - f_i, f_i]"""
    B, C, N = x.shape
    k = idx.shape[-1]
    xt = x.transpose(2, 1)                                          # (B,N,C)
    nb = torch.gather(xt.unsqueeze(1).expand(B, N, N, C), 2, idx.unsqueeze(-1).expand(B, N, k, C))
    ctr = xt.unsqueeze(2).expand(B, N, k, C)
    return torch.cat((nb - ctr, ctr), dim=3).permute(0, 3, 1, 2).contiguous()


def dgcnn_edge_layer(p, f, i, k=20, knn_fn=None):
    """EdgeConv layer i (1..4) on its own: f (B,C,N) -> (B,Co,N)"""
    idx = (knn_fn or knn_feat_oracle)(f, k)
    e = F.conv2d(graph_feature(f, idx), p["conv%d.0.weight" % i])
    return F.leaky_relu(_bn(e, p, "conv%d.1" % i, 4), 0.2).max(dim=-1)[0]


def dgcnn_backbone(p, x, k=20, stages=None, knn_fn=None):
    """x (B,3,N) -> (xyz (B,3,N), feats (B,emb_dims,N))"""
    knn_fn = knn_fn or knn_feat_oracle
    outs = []
    f = x
    for i in (1, 2, 3, 4):
        idx = knn_fn(f, k)
        if stages is not None:
            stages["knn%d" % i] = idx
        e = graph_feature(f, idx)
        e = F.conv2d(e, p["conv%d.0.weight" % i])
        e = F.leaky_relu(_bn(e, p, "conv%d.1" % i, 4), 0.2)
        f = e.max(dim=-1)[0]
        if stages is not None:
            stages["x%d" % i] = f
        outs.append(f)
    c = torch.cat(outs, dim=1)
    y = F.conv1d(c, p["conv5.0.weight"])
    y = F.leaky_relu(_bn(y, p, "conv5.1", 3), 0.2)
    return x, y


def dgcnn_pairs(sd, s1, s2, k=20, stages=None, knn_fn=None):
    """DGCNN ReIDNet (use_dgcnn branch of siamese_forward, ReIDNet.py:316-324; downsample ng = 64, 16)."""
    b, n, _ = s1.shape
    x = torch.cat([s1, s2], 0).permute(0, 2, 1).contiguous()
    xyz, f = dgcnn_backbone(_sub(sd, "backbone."), x, k, stages, knn_fn)
    t = f.permute(0, 2, 1).reshape(-1, f.shape[1])
    t = linear_res(_linres_params(sd, "downsample.0.", 64), t)
    t = linear_res(_linres_params(sd, "downsample.1.", 16), t)
    t = F.linear(t, sd["downsample.2.weight"], sd["downsample.2.bias"])
    h = t.reshape(2 * b, n, -1).permute(0, 2, 1)
    xyz = xyz.permute(0, 2, 1)
    if stages is not None:
        stages.update(h1=h[:b], h2=h[b:], enc_max=f.max(dim=2)[0], enc_mean=f.mean(dim=2))
    return match(sd, h[:b], xyz[:b], h[b:], xyz[b:], stages, head_ng=16)


# ---- PointNet++ SSG encoder (BASELINE config 2; build-defined composition) ---------
# Semantics restated from the reference's mmdet3d ops: Points_Sampler D-FPS (points_sampler.py:107-119),
# gather_points, QueryAndGroup with ball query and use_xyz (group_points.py:94-118), ConvModule stack
# (1x1 Conv2d bias-free + BN2d + ReLU, point_sa_module.py:289-299), max pool (:166-182).  FPS and ball
# query indices come from the C oracle (oracle/pcr_oracle.c), which replays the CUDA kernels' rules.
def ssg_sa_layer(p, xyz, feats, npoint, radius, nsample, stages=None, tag=""):
    import numpy as np
    import point_ops as P
    x = xyz.numpy()
    fps = P.fps(x, npoint)
    new_xyz = torch.from_numpy(np.take_along_axis(x, fps[..., None].astype(np.int64).repeat(3, -1), 1))
    idx = torch.from_numpy(P.ball_query(0.0, radius, nsample, x, new_xyz.numpy())).long()
    g = _gather_rows(xyz, idx) - new_xyz.unsqueeze(2)
    if feats is not None:
        g = torch.cat([g, _gather_rows(feats.permute(0, 2, 1), idx)], dim=-1)
    h = g.permute(0, 3, 1, 2)
    for i in range(3):
        h = F.conv2d(h, p[f"mlps.0.layer{i}.conv.weight"])
        h = F.relu(_bn(h, p, f"mlps.0.layer{i}.bn", 4))
    out = h.max(dim=3)[0]
    if stages is not None:
        stages[tag + "_fps"] = torch.from_numpy(fps)
        stages[tag + "_ball"] = idx
        stages[tag + "_out"] = out
    return new_xyz, out


def ssg_backbone(p, pc, num_points=(512, 128), radii=(0.2, 0.4), num_samples=(32, 64), stages=None):
    xyz = pc[..., :3].contiguous()
    feats = None
    for i in range(len(num_points)):
        xyz, feats = ssg_sa_layer(_sub(p, f"SA_modules.{i}."), xyz, feats, num_points[i], radii[i], num_samples[i],
                                  stages, f"sa{i}")
    return xyz, F.conv1d(feats, p["cov_final.weight"], p["cov_final.bias"])


def ssg_pairs(sd, s1, s2, stages=None, **kw):
    b = s1.shape[0]
    xyz, h = ssg_backbone(_sub(sd, "backbone."), torch.cat([s1, s2], 0), stages=stages, **kw)
    if stages is not None:
        stages.update(h1=h[:b], h2=h[b:])
    return match(sd, h[:b], xyz[:b], h[b:], xyz[b:], stages)
