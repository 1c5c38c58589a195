"""Training-mode forward of the Point-Transformer ReID path as a differentiable graph whose every node is a HIP launch.

`ReIDNet.train_step` in training mode runs through here: neighbour search (pcr_knn_prefix_f32, no gradient), the
grouped set-abstraction MLPs with BatchNorm in batch-statistics mode (train_ops.SaEdgeTrain), the linear-attention
blocks (dense projections + LinAttn + LayerNorm), the matching stages, pair pooling and the LinearRes + Linear head.
torch.autograd only strings the Functions of pcr_amd/train_ops.py together and owns the scalar loss
(BCEWithLogits over B logits); no torch matmul / conv / norm kernel is on this path.  It follows the reference's graph
(mmdet3d/models/pointnet2_utils.py:55-114, 242-288, 333-437; attention.py:192-219; lanegcn_nets.py:228-241;
ReIDNet.py:231-247, 526-534) and is pinned to the reference's loss and gradients by tests/golden/pt_train_step_n128.npz.
Tensors are (B, C, L) channel-major throughout, as in the inference path.  Nothing here is used in eval mode.
"""
import torch

from . import engine
from . import train_ops as TO

ATTN_EPS = 1e-6


def _cm(xyz):
    """(B,L,3) -> (B,3,L) channel-major coordinates"""
    return xyz.transpose(1, 2).contiguous()


def _tail(m, msg, res_in, residual):
    """merge, LayerNorm, feed-forward on [res_in ; msg], LayerNorm (+ residual): ONE launch each way where the chain
    kernel is instantiated (train_ops.attn_tail, round 5), else one launch per layer"""
    fused = TO.attn_tail(m, msg, res_in, residual)
    if fused is not None:
        return fused
    n1 = TO.tnorm(TO.dense(msg, m.merge.weight), m.norm1)
    f0 = TO.dense(res_in, m.mlp[0].weight, x2=n1, relu=True)
    return TO.tnorm(TO.dense(f0, m.mlp[2].weight), m.norm2, res=res_in if residual else None)


def _pos(pos_mlp, xyz_cm, add_to):
    """add_to + Linear(ReLU(Linear(xyz)))  (the residual add rides in the second dense launch)"""
    h = TO.dense(xyz_cm, pos_mlp[0].weight, pos_mlp[0].bias, relu=True)
    return TO.dense(h, pos_mlp[2].weight, pos_mlp[2].bias, res=add_to)


def _flat(x):
    """(B,C,L) -> (1,C,B L): the batch as ONE token axis (a copy)"""
    B, C, Ln = x.shape
    return x.permute(1, 0, 2).reshape(1, C, B * Ln)


def _unflat(x, B):
    """(1,C,B L) -> (B,C,L)"""
    C = x.shape[1]
    return x.reshape(C, B, -1).permute(1, 0, 2).contiguous()


# clouds of at most this many tokens run the per-token layers of an UNFUSED attention block on the flattened batch
FLAT_TOKENS = 32


def self_attention(m, feat, xyz_cm):
    """Self_Attention (pointnet2_utils.py:90-114): q, k, v all project feat + position code"""
    ws = (m.q_proj.weight, m.k_proj.weight, m.v_proj.weight)
    qkv = TO.attn_head(m.pos_mlp, feat, xyz_cm, ws, 7)      # position MLP + the three projections: ONE launch each way
    if qkv is not None:
        return _tail(m, TO.LinAttnQKV.apply(qkv, m.nhead, ATTN_EPS, 0), feat, True)
    B, _, Ln = feat.shape
    if B > 1 and Ln <= FLAT_TOKENS:
        # SA3's block (d_model 128: no fused chain) on 32-token clouds: every train-dense launch works on 64-token tiles of
        # ONE cloud, so half of every tile is padding and every cloud is a workgroup with its own dW partial record (134 MB
        # of partials for the 256 x 256 layer at 512 clouds).  Everything except the attention core maps a token to a
        # token: those layers run on the batch flattened to one token axis (full tiles, half the workgroups and partials);
        # the core gets its (B, 3 d, L) view back.  Five extra copies of 8-25 MB each way.
        ff, xf = _flat(feat), _flat(xyz_cm)
        qkv = _unflat(TO.dense(_pos(m.pos_mlp, xf, ff), torch.cat(ws, dim=0)), B)
        msg = _flat(TO.LinAttnQKV.apply(qkv, m.nhead, ATTN_EPS, 0))
        return _unflat(_tail(m, msg, ff, True), B)
    qkv = TO.dense(_pos(m.pos_mlp, xyz_cm, feat), torch.cat(ws, dim=0))
    return _tail(m, TO.LinAttnQKV.apply(qkv, m.nhead, ATTN_EPS, 0), feat, True)


def _cross(m, pos_mlp, q_in, kv_in, kv_xyz_cm, residual):
    """q from q_in; k from kv_in, v from kv_in + position code (FP_SA :407-437, corss_attention attention.py:192-219)"""
    q = TO.dense(q_in, m.q_proj.weight)
    kv = TO.attn_head(pos_mlp, kv_in, kv_xyz_cm, (m.k_proj.weight, m.v_proj.weight), 2)
    if kv is not None:
        msg = TO.LinAttnKV.apply(q, kv, m.nhead, ATTN_EPS)
    else:
        k, v = TO.dense(kv_in, m.k_proj.weight), TO.dense(_pos(pos_mlp, kv_xyz_cm, kv_in), m.v_proj.weight)
        msg = TO.LinAttn.apply(q, k, v, m.nhead, ATTN_EPS)
    return _tail(m, msg, q_in, residual)


def cross_attention_pairs(m, feats, xyz_cm, b):
    """corss_attention of every cloud of a (2b, C, N) batch against its pair partner (cloud i <-> cloud i +- b): q | k | v
    of every cloud from ONE head launch (q, k of x; v of x + position code), the partner's keys / values addressed by the
    attention core (kv_roll) -- no swapped copies of the batch, no separate q projection.  None: no fused instantiation."""
    qkv = TO.attn_head(m.pos_mlp, feats, xyz_cm, (m.q_proj.weight, m.k_proj.weight, m.v_proj.weight), 4)
    if qkv is None:
        return None
    return _tail(m, TO.LinAttnQKV.apply(qkv, m.nhead, ATTN_EPS, b), feats, True)


def fp_sa(m, feat1, feat2, xyz2_cm):
    return _cross(m, m.pos_mlp2, feat1, feat2, xyz2_cm, False)


def cross_attention(m, search, template, template_xyz_cm):
    return _cross(m, m.pos_mlp, search, template, template_xyz_cm, True)


def local_self_attention(m, feat, xyz_cm):
    """local_self_attention (attention.py:262-296) in training mode: feat (B,C,N), xyz (B,3,N) -> (B,C,N).  The
    neighbour graph is the feature-space kNN of the block's INPUT (no gradient, as torch.topk's indices); key and value
    of an edge are the neighbour point's own projections, so q | k | v come from one dense launch over the points"""
    from . import dgcnn_engine
    from . import _lib as L
    if m.pos_mlp_knn[2].weight.shape[0] != m.d_model:
        raise L.PcrError("local_self_attention: pos_size must equal d_model (the position code is added to the features)")
    idx = dgcnn_engine.knn_feat(feat.detach().contiguous(), int(m.knum))                 # (B,N,K) int32
    fp = _pos(m.pos_mlp_knn, xyz_cm, feat)
    qkv = TO.dense(fp, torch.cat([m.q_proj_knn.weight, m.k_proj_knn.weight, m.v_proj_knn.weight], dim=0))
    msg = TO.LocalAttn.apply(qkv, idx, int(m.nhead), ATTN_EPS)
    n1 = TO.tnorm(TO.dense(msg, m.merge_knn.weight), m.norm1_knn)
    f0 = TO.dense(feat, m.mlp_knn[0].weight, x2=n1, relu=True)
    return TO.tnorm(TO.dense(f0, m.mlp_knn[2].weight), m.norm2_knn, res=feat)


def sa_edge_layer(sa, xyz, feats, s, knn_idx=None):
    """PointNetSetAbstractionEdgeSA in training mode: (B,N,3), (B,D,N)|None -> (B,S,3), (B,D',S)
    knn_idx: the level's neighbours when the caller searched already (engine.knn_prefix2)"""
    xyz = xyz.contiguous()
    idx = knn_idx if knn_idx is not None else engine.knn_prefix(xyz, s, sa.nsample)   # HIP, (B,S,K) int32
    new_xyz = xyz[:, :s].contiguous()
    x = TO.sa_edge_train(sa, xyz, None if feats is None else feats.contiguous(), idx)
    return new_xyz, self_attention(sa.self_attention, x, _cm(new_xyz))


def backbone(bb, pointcloud, numpoints):
    xyz = pointcloud[..., 0:3].detach().contiguous()
    l_xyz, l_feat = [xyz], [None]
    from mmdet3d.models import backbone_net as BN
    shared = None
    for i, sa in enumerate(bb.SA_modules):
        knn_idx = None
        if i == 0 and len(bb.SA_modules) > 1 and hasattr(sa, "shares_knn_with") and not BN._NO_KNN2 and \
                sa.shares_knn_with(bb.SA_modules[1], xyz.shape[1], numpoints[0], numpoints[1]):
            # the first level keeps every point: the second one searches the same cloud (one launch ranks both)
            knn_idx, shared = engine.knn_prefix2(xyz, numpoints[0], sa.nsample, numpoints[1], bb.SA_modules[1].nsample)
        elif i == 1:
            knn_idx = shared
        nx, nf = sa_edge_layer(sa, l_xyz[i], l_feat[i], numpoints[i], knn_idx=knn_idx)
        l_xyz.append(nx)
        l_feat.append(nf)
    l_feat[0] = _cm(xyz)
    for i in (2, 1, 0):
        l_feat[i] = fp_sa(bb.FP_modules[i].interpolation, l_feat[i], l_feat[i + 1], _cm(l_xyz[i + 1]))
    cf = bb.cov_final
    return xyz, TO.dense(l_feat[0], cf.weight.view(cf.weight.shape[0], -1), cf.bias)


def _conv_bn(x, conv, bn, relu=True):
    return TO.bn_act(TO.dense(x, conv.weight.view(conv.weight.shape[0], -1), conv.bias), bn, relu)


def _stn(m, x):
    """STN3d / STNkd (models/pointnet.py:27-45, 67-85) in training mode: x (B,k,N) -> per-cloud (B,k,k) transforms"""
    B = x.shape[0]
    h = _conv_bn(_conv_bn(_conv_bn(x, m.conv1, m.bn1), m.conv2, m.bn2), m.conv3, m.bn3)
    g = TO.PoolBoth.apply(h)[:, :h.shape[1]]                 # max over the points: (B, 1024)
    f = g.t().contiguous().unsqueeze(0)                      # (1, 1024, B): clouds as tokens, BatchNorm1d over the batch
    f = _conv_bn(_conv_bn(f, m.fc1, m.bn4), m.fc2, m.bn5)
    t = TO.dense(f, m.fc3.weight, m.fc3.bias)                # (1, k*k, B)
    iden = torch.eye(m.k, dtype=t.dtype, device=t.device).flatten().unsqueeze(0)
    return (t.squeeze(0).t() + iden).reshape(B, m.k, m.k)


def pointnet_encoder(enc, xyz):
    """PointNetEncoder.forward with feature_transform (models/pointnet.py:103-127) in training mode:
    xyz (B,3,N) -> (xyz, per-point features (B,1024,N)); every BatchNorm uses batch statistics"""
    x = xyz.contiguous()
    x = TO.Bmm.apply(x, _stn(enc.stn, x))
    x = _conv_bn(x, enc.conv1, enc.bn1)
    x = TO.Bmm.apply(x, _stn(enc.fstn, x))
    x = _conv_bn(x, enc.conv2, enc.bn2)
    return xyz, _conv_bn(x, enc.conv3, enc.bn3, relu=False)


def dgcnn(net, xyz):
    """DGCNN.forward (models/dgcnn_orig.py:127-152) in training mode: xyz (B,3,N) -> (xyz, per-point features
    (B,emb_dims,N)); feature-space kNN (no gradient, as torch.topk's indices) + EdgeConv with batch statistics"""
    from . import dgcnn_engine
    x = xyz.contiguous()
    outs = []
    f = x
    for conv in (net.conv1, net.conv2, net.conv3, net.conv4):
        idx = dgcnn_engine.knn_feat(f.detach().contiguous(), int(net.k))                 # (B,N,k) int32
        w = conv[0].weight.view(conv[0].weight.shape[0], -1)                             # (Co, 2C): [W1 | W2]
        c = w.shape[1] // 2
        tab = TO.dense(f, torch.cat([w[:, :c], w[:, c:] - w[:, :c]], dim=0))              # (B, 2Co, N)
        f = TO.EdgeConvTrain.apply(tab, idx, conv[1].weight, conv[1].bias, conv[1], float(conv[2].negative_slope))
        outs.append(f)
    cat = torch.cat(outs, dim=1)
    c5 = net.conv5
    y = TO.bn_act(TO.dense(cat, c5[0].weight.view(c5[0].weight.shape[0], -1)), c5[1], True, float(c5[2].negative_slope))
    return xyz, y


def downsample_points(mods, x):
    """ReIDNet.downsample ([LinearRes, LinearRes, Linear], ReIDNet.py:316-324) per point on (B,C,N) in training mode"""
    from mmdet3d.models.lanegcn_nets import LinearRes
    for m in mods:
        if isinstance(m, LinearRes):
            x = linear_res_rows(m, x)
        elif isinstance(m, torch.nn.Linear):
            x = TO.dense(x, m.weight, m.bias)
        else:
            from . import _lib as L
            raise L.PcrError("downsample: %s has no HIP training form" % type(m).__name__)
    return x


def linear_res_rows(m, x):
    """LinearRes (lanegcn_nets.py:228-241) on (M, n) rows, evaluated channel-major with the rows as tokens (1, n, M):
    relu(GN(W1 x)) -> GN(W2 .) + shortcut -> relu, the ReLUs and the shortcut add inside the norm launches"""
    out = TO.tnorm(TO.dense(x, m.linear1.weight), m.norm1, relu=True)
    short = x if m.transform is None else TO.tnorm(TO.dense(x, m.transform[0].weight), m.transform[1])
    return TO.tnorm(TO.dense(out, m.linear2.weight), m.norm2, res=short, relu=True)


def _joined(a1, a2):
    """[a1 ; a2] along the batch -- the tensor they were sliced from when they are its two halves (siamese_forward
    returns h[:b], h[b:]): no copy, and no slice / cat nodes between the backbone and the matching stages"""
    base = a1._base
    if (base is not None and base is a2._base and base.is_contiguous() and base.shape[0] == 2 * a1.shape[0]
            and base.shape[1:] == a1.shape[1:] == a2.shape[1:] and a1.is_contiguous() and a2.is_contiguous()
            and a1.data_ptr() == base.data_ptr() and a2.data_ptr() == base.data_ptr() + a1.numel() * a1.element_size()):
        return base
    return torch.cat([a1, a2], dim=0)


def _head(model, pooled):
    """match_head = [LinearRes, Linear] on pooled rows (B,F), samples as tokens"""
    x = pooled.t().contiguous().unsqueeze(0)                               # (1, F, B)
    x = linear_res_rows(model.match_head[0], x)
    return TO.dense(x, model.match_head[1].weight, model.match_head[1].bias).reshape(-1)   # (1, 1, B)


def supports(model):
    """the training graph covers: xcorr_eff + point-cat + pool 'both' (every point-cat config), xcorr-baseline + pool
    'both' (reid_pts_point-transformer_baseline_stnet.py), xcorr + pool 'both' (..._baseline_orig.py) and concat + pool 'max'
    (reid_pts_point-transformer_baseline.py); -> None or the reason it does not"""
    head = model.match_head
    if not (isinstance(head, torch.nn.Sequential) and len(head) == 2):
        return "match_head must be [LinearRes, Linear]"
    mt, pool = model.match_type, model.pool_type
    if mt == "xcorr_eff" and model.combine == "point-cat" and pool == "both":
        return None
    if mt == "xcorr-baseline" and pool == "both":
        return None
    if mt == "xcorr" and pool == "both":
        if model.local_stage1 is None or model.local_stage2 is None:
            return "match_type='xcorr' needs local_stage1 / local_stage2 (local_self_attention)"
        return None
    if mt == "concat" and pool == "max":
        return None
    return ("match_type=%r / combine=%r / pool_type=%r has no HIP training graph (covered: xcorr_eff + point-cat + both, "
            "xcorr-baseline + both, xcorr + both, concat + max)" % (mt, model.combine, pool))


def match_logits(model, h1, xyz1, h2, xyz2):
    """matching + pooling + LinearRes + Linear in training mode -> (logits (B), stage-2 features or None)
    * xcorr_eff + point-cat + pool 'both' (ReIDNet.py:231-247, 526-534, 455-457): the four cross-attention calls of the
      reference run as two, over all 2B clouds with the halves swapped as templates;
    * xcorr-baseline (:258-264): the search branch only, two cross-attention stages against the template, pool 'both';
    * concat + pool 'max' (:415-419, 526-528): channel-window max of each encoding, concatenated."""
    why = supports(model)
    if why is not None:
        from . import _lib as L
        raise L.PcrError(why)
    b = h1.shape[0]
    if model.match_type == "concat":
        w = model.output_sequence_size
        if h1.shape[1] != w:
            from . import _lib as L
            raise L.PcrError("concat + pool 'max' pools one window of %d channels per point; the encoder gives %d"
                             % (w, h1.shape[1]))
        pooled = TO.ChannelMax.apply(_joined(h1, h2), w)                   # (2B, 1, N)
        pooled = pooled.reshape(2 * b, -1)
        return _head(model, torch.cat([pooled[:b], pooled[b:]], dim=1)), None
    if model.match_type == "xcorr-baseline":
        a = cross_attention(model.cross_stage1, h1, h2, _cm(xyz2))
        o = cross_attention(model.cross_stage2, a, h2, _cm(xyz2))
        return _head(model, TO.PoolBoth.apply(o)), None
    if model.match_type == "xcorr":          # baseline-orig (ReIDNet.py:250-256): cross -> local -> cross -> local
        x1, x2 = _cm(xyz1), _cm(xyz2)
        a = cross_attention(model.cross_stage1, h1, h2, x2)
        bb = local_self_attention(model.local_stage1, a, x1)
        c = cross_attention(model.cross_stage2, bb, h2, x2)
        o = local_self_attention(model.local_stage2, c, x1)
        return _head(model, TO.PoolBoth.apply(o)), None
    feats = _joined(h1, h2)
    xyz_cm = _cm(_joined(xyz1, xyz2))
    a = cross_attention_pairs(model.cross_stage1, feats, xyz_cm, b)
    o = None if a is None else cross_attention_pairs(model.cross_stage2, a, xyz_cm, b)
    if o is None:
        swap = lambda t: torch.roll(t, b, 0)                # noqa: E731  (halves exchanged: one launch each way)
        a = cross_attention(model.cross_stage1, feats, swap(feats), swap(xyz_cm))
        o = cross_attention(model.cross_stage2, a, swap(a), swap(xyz_cm))
    return _head(model, TO.PoolPair.apply(o)), o
