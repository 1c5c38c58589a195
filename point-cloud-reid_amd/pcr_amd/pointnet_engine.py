"""PointNet encoder forward (reference: mmdet3d/models/pointnet.py:27-45, 67-85, 103-127) as a
sequence of libpcr_hip.so launches: 1x1 convs / fc layers with folded eval-mode BatchNorm are
pcr_dense_f32, the global max pool is pcr_max_over_l_f32 (written channel-major so the fc stack needs
no transposes), the learned 3x3 / 64x64 input and feature transforms are applied with per-cloud
packed weights (pcr_pack_bmm_f32 + pcr_dense_bmm_f32)."""
import torch

import os

from . import _lib as L
from . import engine as E

_NO_DENSE_MAX = bool(os.environ.get("PCR_NO_DENSE_MAX"))     # diagnostics: conv3 and the max over the points as two launches


# (The encoder's convs and both transform nets stay on the f32-input MFMA in every mode: measured in split bf16, the
# 1024-wide features (|x| ~ 90) come out 1.2e-3 off and the per-point embeddings behind them 1.6e-4 -- outside the 1e-4
# parity bound -- because an error in the learned 3 x 3 / 64 x 64 transforms multiplies every feature.  The LinearRes
# downsample rows, which hold most of the encoder's flops, do run on the bf16 matrix core: rows.py.)
class _StnPlan:
    def __init__(self, stn, device):
        self.k = stn.k
        self.convs = []
        for conv, bn in ((stn.conv1, stn.bn1), (stn.conv2, stn.bn2), (stn.conv3, stn.bn3)):
            sc, sh = E.fold_bn(bn, conv.bias, device)
            self.convs.append((E.pack_weight(conv.weight, device), conv.weight.shape[0], sc, sh))
        self.fcs = []
        for fc, bn in ((stn.fc1, stn.bn4), (stn.fc2, stn.bn5)):
            sc, sh = E.fold_bn(bn, fc.bias, device)
            self.fcs.append((E.pack_weight(fc.weight, device), fc.weight.shape[0], sc, sh, 1))
        iden = torch.eye(self.k, dtype=torch.float64).flatten()
        shift = (stn.fc3.bias.detach().double().cpu() + iden).float().to(device).contiguous()
        self.fcs.append((E.pack_weight(stn.fc3.weight, device), self.k * self.k, None, shift, 0))

    def run(self, x):
        """x (B,k,N) -> per-cloud packed transform images"""
        B = x.shape[0]
        lib = L.load()
        for wp, cout, sc, sh in self.convs[:2]:
            x = E.dense(x, wp, cout, sc, sh, act=1)
        wp, C, sc, sh = self.convs[2]
        g = torch.empty((1, C, B), dtype=torch.float32, device=x.device)
        cin, Ln = x.shape[1], x.shape[2]
        if not _NO_DENSE_MAX and x.is_contiguous() and lib.pcr_dense_max_ok(cin, C, Ln):
            # conv3 + BN + ReLU + max over the points in one launch: the (B,1024,N) tensor is never written
            with E._prof("dense_max[cin=%d,cout=%d,L=%d]" % (cin, C, Ln), 2.0 * B * Ln * cin * C, 4.0 * B * (Ln * cin + C),
                         arith="f32"):
                L.check(lib.pcr_dense_max_f32(L.ptr(x), L.ptr(wp), L.ptr(sc), L.ptr(sh), L.ptr(g), B, cin, C, Ln, 1,
                                              L.stream_ptr()), "pcr_dense_max_f32")
        else:
            x = E.dense(x, wp, C, sc, sh, act=1)
            L.check(lib.pcr_max_over_l_f32(L.ptr(x), L.ptr(g), B, C, x.shape[2], L.stream_ptr()), "pcr_max_over_l_f32")
        for wp, cout, sc, sh, act in self.fcs:
            g = E.dense(g, wp, cout, sc, sh, act=act)
        lib = L.load()
        img = torch.empty((B, lib.pcr_packed_weight_floats(self.k, self.k)), dtype=torch.float32, device=x.device)
        L.check(lib.pcr_pack_bmm_f32(L.ptr(g), L.ptr(img), B, self.k, L.stream_ptr()), "pcr_pack_bmm_f32")
        return img


def _bmm(x, img, k):
    B, _, N = x.shape
    y = torch.empty((B, k, N), dtype=torch.float32, device=x.device)
    L.check(L.load().pcr_dense_bmm_f32(L.ptr(x), L.ptr(img), L.ptr(y), B, k, k, N, L.stream_ptr()), "pcr_dense_bmm_f32")
    return y


class _EncoderPlan:
    def __init__(self, enc, device):
        self.stn = _StnPlan(enc.stn, device)
        self.fstn = _StnPlan(enc.fstn, device)
        self.layers = []
        for conv, bn, act in ((enc.conv1, enc.bn1, 1), (enc.conv2, enc.bn2, 1), (enc.conv3, enc.bn3, 0)):
            sc, sh = E.fold_bn(bn, conv.bias, device)
            self.layers.append((E.pack_weight(conv.weight, device), conv.weight.shape[0], sc, sh, act))


def encoder_forward(enc, xyz):
    """xyz (B,3,N) -> (xyz, per-point features (B,1024,N)); eval mode only"""
    L.require_cuda(xyz)
    if not enc.feature_transform:
        raise L.PcrError("PointNetEncoder: the ReID configs build feature_transform=True")
    if enc.training:
        # differentiable graph with BatchNorm batch statistics (pcr_amd/train_graph.py); eval mode below
        from . import train_graph
        return train_graph.pointnet_encoder(enc, xyz.contiguous().float())
    key = (str(xyz.device), E.param_version(enc))
    if getattr(enc, "_pcr_key", None) != key:
        object.__setattr__(enc, "_pcr_plan", _EncoderPlan(enc, xyz.device))
        object.__setattr__(enc, "_pcr_key", key)
    p = enc._pcr_plan
    x = xyz.contiguous().float()
    x = _bmm(x, p.stn.run(x), 3)
    wp, cout, sc, sh, act = p.layers[0]
    x = E.dense(x, wp, cout, sc, sh, act=act)
    x = _bmm(x, p.fstn.run(x), 64)
    for wp, cout, sc, sh, act in p.layers[1:]:
        x = E.dense(x, wp, cout, sc, sh, act=act)
    return xyz, x
