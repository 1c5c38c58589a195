"""Matching-head attention blocks (reference: mmdet3d/models/attention.py).  `corss_attention`
(sic, the reference's spelling is part of the config vocabulary) is the linear cross-attention
used by ReIDNet.xcorr_eff; the image-only / feature-kNN variants are outside the hot path."""
import torch
import torch.nn as nn

from pcr_amd import engine
from .pointnet2_utils import _Planned, _attn_holder


class corss_attention(_Planned):
    """search (B,C,Ns), template (B,C,Nt) -> (B,C,Ns); keys = template, values = template +
    pos_mlp(template_xyz), queries = search, residual on search (attention.py:192-219)."""

    def __init__(self, d_model, nhead, attention="linear"):
        super().__init__()
        self.dim = d_model // nhead
        self.nhead = nhead
        _attn_holder(self, "pos_mlp", d_model, d_model, d_model, d_model, d_model * 2, d_model)

    def plan(self, device):
        return self._plan(device, lambda dev: engine.AttnPlan(self, "pos_mlp", dev, self.nhead, q_pos=False,
                                                              k_pos=False, residual=True))

    def forward(self, search_feat, search_xyz, template_feat, template_xyz, mask=None):
        assert mask is None
        return self.plan(search_feat.device).run(search_feat.contiguous(), search_xyz.contiguous(),
                                                 template_feat.contiguous(), template_xyz.contiguous())

    def forward_paired(self, feats, xyz, partner):
        """all 2B clouds at once: cloud b attends to cloud partner[b] (no concatenation copies)"""
        return self.plan(feats.device).run(feats, xyz, feats, xyz, kv_index=partner)


class _OutOfScope(nn.Module):
    what = ""

    def __init__(self, *a, **k):
        super().__init__()
        raise NotImplementedError("%s is outside the siamese point-cloud hot path rebuilt here "
                                  "(SURVEY.md section 8f)" % self.what)


class local_self_attention(nn.Module):
    """Every point attends to its `knum` feature-space neighbours (attention.py:221-296; the `baseline_orig`
    matching, match_type='xcorr').  Parameter names are the reference's.  search_feat (B,C,N), search_xyz (B,N,3)
    -> (B,C,N)."""

    def __init__(self, d_model, nhead, attention="linear", knum=32, pos_size=16):
        super().__init__()
        self.d_model, self.dim, self.nhead, self.knum = d_model, d_model // nhead, nhead, knum
        self.pos_mlp_knn = nn.Sequential(nn.Linear(3, pos_size), nn.ReLU(True), nn.Linear(pos_size, pos_size))
        self.q_proj_knn = nn.Linear(d_model, d_model, bias=False)
        self.k_proj_knn = nn.Linear(d_model, d_model, bias=False)
        self.v_proj_knn = nn.Linear(d_model, d_model, bias=False)
        self.merge_knn = nn.Linear(d_model, d_model, bias=False)
        self.mlp_knn = nn.Sequential(nn.Linear(d_model * 2, d_model * 2, bias=False), nn.ReLU(True),
                                     nn.Linear(d_model * 2, d_model, bias=False))
        self.norm1_knn = nn.LayerNorm(d_model)
        self.norm2_knn = nn.LayerNorm(d_model)

    def forward(self, search_feat, search_xyz, mask=None):
        assert mask is None
        from pcr_amd import local_attn_engine
        return local_attn_engine.forward(self, search_feat, search_xyz)


class cross_lin_attn(_OutOfScope):
    what = "cross_lin_attn (image-token variant)"
