"""LinearRes: the residual two-layer GroupNorm MLP of the match head and of the PointNet
`downsample` stack (reference: mmdet3d/models/lanegcn_nets.py:193-241).  The rest of the
reference's lanegcn_nets.py (LaneGCN conv blocks, BEV RoI sampling) is outside the ReID path."""
from math import gcd

import torch.nn as nn


class LinearRes(nn.Module):
    """relu(GN(W2 relu(GN(W1 x))) + shortcut(x)); shortcut = Linear+GN when n_in != n_out.

    Parameter names match the reference (linear1, linear2, norm1, norm2, transform.{0,1}).  Inside
    ReIDNet.match_head the arithmetic runs in the fused head kernel (pcr_pool_head_f32); called on its
    own (or from the PointNet `downsample` stack) it runs as dense + GroupNorm launches (pcr_amd/rows.py)."""

    def __init__(self, n_in, n_out, norm="GN", ng=32, activation="ReLU"):
        super().__init__()
        assert norm in ("GN", "BN", "SyncBN")
        if norm != "GN":
            raise NotImplementedError("only norm='GN' is used by the ReID configs")
        if activation != "ReLU":
            raise NotImplementedError("only activation='ReLU' is used by the ReID configs")
        groups = gcd(ng, n_out)
        self.linear1 = nn.Linear(n_in, n_out, bias=False)
        self.linear2 = nn.Linear(n_out, n_out, bias=False)
        self.relu = nn.ReLU(inplace=True)
        self.norm1 = nn.GroupNorm(groups, n_out)
        self.norm2 = nn.GroupNorm(groups, n_out)
        if n_in != n_out:
            self.transform = nn.Sequential(nn.Linear(n_in, n_out, bias=False), nn.GroupNorm(groups, n_out))
        else:
            self.transform = None

    def forward(self, x):
        from pcr_amd import rows
        return rows.linear_res(self, x)
