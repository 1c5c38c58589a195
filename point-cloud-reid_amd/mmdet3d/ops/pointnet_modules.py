"""mmdet3d-style PointNet++ building blocks with the reference's names and semantics, over the HIP
point ops and the fused SA kernel.

Reference (bentherien/point-cloud-reid, mmdet3d/ops/...):
  furthest_point_sample/points_sampler.py:34-157   Points_Sampler, DFPS/FFPS/FS samplers
  furthest_point_sample/utils.py:4-32              calc_square_dist
  group_points/group_points.py:11-166              QueryAndGroup, GroupAll
  pointnet_modules/point_sa_module.py:10-354       BasePointSAModule, PointSAModuleMSG, PointSAModule
  pointnet_modules/point_fp_module.py:10-79        PointFPModule
  pointnet_modules/builder.py:3-38                 SA_MODULES, build_sa_module
mmcv's ConvModule (1x1 Conv2d, bias off when a norm follows, BatchNorm2d, ReLU) is restated as
`ConvModule` below with the same sub-module names (`conv`, `bn`, `activate`) so checkpoints line up.

Eval-mode forward of a 3-layer SA scale is ONE fused launch (pcr_sa_mlp_f32, mode 1) after FPS and
ball query; nothing of shape (B,C,S,K) is materialised.  `QueryAndGroup` / `grouping_operation`
remain available as stand-alone ops for callers that want the grouped tensor.
"""
import os

import torch
from torch import nn as nn

from pcr_amd import engine
from pcr_amd import _lib as L
from ..models.builder import Registry
from .point_ops import (ball_query, ball_query_cnt, ball_query_rows, fps_ball_query_rows, fps_ball_query_rows_ok, furthest_point_sample, furthest_point_sample_with_dist, gather_points,
                        grouping_operation, knn, three_interpolate, three_nn)

SA_MODULES = Registry("point_sa_module")


class ConvModule(nn.Module):
    """1x1 conv -> BatchNorm -> ReLU (mmcv.cnn.ConvModule with conv_cfg=Conv2d, norm_cfg=BN2d)"""

    def __init__(self, in_channels, out_channels, kernel_size=(1, 1), stride=(1, 1), conv_cfg=None, norm_cfg=None,
                 bias="auto", act=True):
        super().__init__()
        norm_cfg = norm_cfg if norm_cfg is not None else dict(type="BN2d")
        if norm_cfg.get("type") not in ("BN2d", "BN"):
            raise NotImplementedError("norm_cfg %r: the point modules use BatchNorm2d" % (norm_cfg,))
        if bias == "auto":
            bias = False
        self.conv = nn.Conv2d(in_channels, out_channels, kernel_size=1, bias=bias)
        self.bn = nn.BatchNorm2d(out_channels)
        self.activate = nn.ReLU(inplace=True) if act else None


def calc_square_dist(point_feat_a, point_feat_b, norm=True):
    """(B,N,C),(B,M,C) -> (B,N,M) squared distances by the expanded form, as the reference's F-FPS does
    (furthest_point_sample/utils.py:4-32); one HIP launch with a fixed summation order (pcr_pairwise_sqdist_f32)"""
    L.require_cuda(point_feat_a, point_feat_b)
    L.require_f32(point_feat_a, point_feat_b)
    a, b = point_feat_a.contiguous(), point_feat_b.contiguous()
    B, N, C = a.shape
    M = b.shape[1]
    dist = torch.empty((B, N, M), dtype=torch.float32, device=a.device)
    L.check(L.load().pcr_pairwise_sqdist_f32(L.ptr(a), L.ptr(b), L.ptr(dist), B, N, M, C, int(bool(norm)), L.stream_ptr()),
            "pcr_pairwise_sqdist_f32")
    return dist


class DFPS_Sampler(nn.Module):
    def forward(self, points, features, npoint):
        return furthest_point_sample(points.contiguous(), npoint)


class FFPS_Sampler(nn.Module):
    def forward(self, points, features, npoint):
        assert features is not None, "feature input to FFPS_Sampler should not be None"
        f = torch.cat([points, features.transpose(1, 2)], dim=2)
        return furthest_point_sample_with_dist(calc_square_dist(f, f, norm=False).contiguous(), npoint)


class FS_Sampler(nn.Module):
    def forward(self, points, features, npoint):
        assert features is not None, "feature input to FS_Sampler should not be None"
        f = torch.cat([points, features.transpose(1, 2)], dim=2)
        ffps = furthest_point_sample_with_dist(calc_square_dist(f, f, norm=False).contiguous(), npoint)
        dfps = furthest_point_sample(points.contiguous(), npoint)
        return torch.cat([ffps, dfps], dim=1)


def get_sampler_type(sampler_type):
    try:
        return {"D-FPS": DFPS_Sampler, "F-FPS": FFPS_Sampler, "FS": FS_Sampler}[sampler_type]
    except KeyError:
        raise ValueError('Only "sampler_type" of "D-FPS", "F-FPS", or "FS" are supported, got %s' % sampler_type)


class Points_Sampler(nn.Module):
    def __init__(self, num_point, fps_mod_list=["D-FPS"], fps_sample_range_list=[-1]):
        super().__init__()
        assert len(num_point) == len(fps_mod_list) == len(fps_sample_range_list)
        self.num_point = num_point
        self.fps_sample_range_list = fps_sample_range_list
        self.samplers = nn.ModuleList([get_sampler_type(m)() for m in fps_mod_list])
        self.fp16_enabled = False

    def forward(self, points_xyz, features):
        indices = []
        last = 0
        for rng, sampler, npoint in zip(self.fps_sample_range_list, self.samplers, self.num_point):
            assert rng < points_xyz.shape[1]
            if rng == -1:
                xyz = points_xyz[:, last:]
                feat = features[:, :, last:] if features is not None else None
            else:
                xyz = points_xyz[:, last:rng]
                feat = features[:, :, last:rng] if features is not None else None
            idx = sampler(xyz.contiguous().float(), feat, npoint)
            indices.append(idx + last)
            last += rng
        return torch.cat(indices, dim=1)


class QueryAndGroup(nn.Module):
    """ball query (or kNN when max_radius is None) + grouping; returns the (B,3+C,npoint,K) tensor"""

    def __init__(self, max_radius, sample_num, min_radius=0, use_xyz=True, return_grouped_xyz=False,
                 normalize_xyz=False, uniform_sample=False, return_unique_cnt=False, return_grouped_idx=False):
        super().__init__()
        self.max_radius, self.min_radius, self.sample_num = max_radius, min_radius, sample_num
        self.use_xyz, self.return_grouped_xyz, self.normalize_xyz = use_xyz, return_grouped_xyz, normalize_xyz
        self.uniform_sample, self.return_unique_cnt, self.return_grouped_idx = (uniform_sample, return_unique_cnt,
                                                                               return_grouped_idx)
        if uniform_sample:
            raise NotImplementedError("uniform_sample draws host-side random replacements; unused by the ReID path")
        if self.max_radius is None:
            assert not self.normalize_xyz, "can not normalize grouped xyz when max_radius is None"

    def query(self, points_xyz, center_xyz):
        if self.max_radius is None:
            return knn(self.sample_num, points_xyz, center_xyz, False).transpose(1, 2).contiguous()
        return ball_query(self.min_radius, self.max_radius, self.sample_num, points_xyz, center_xyz)

    def query_cnt(self, points_xyz, center_xyz):
        """(idx, cnt): cnt = genuine hits per row for ball queries, None for kNN (no repeated rows)"""
        if self.max_radius is None:
            return self.query(points_xyz, center_xyz), None
        return ball_query_cnt(self.min_radius, self.max_radius, self.sample_num, points_xyz, center_xyz)

    def forward(self, points_xyz, center_xyz, features=None):
        idx = self.query(points_xyz, center_xyz)
        xyz_trans = points_xyz.transpose(1, 2).contiguous()
        grouped_xyz = grouping_operation(xyz_trans, idx)
        diff = grouped_xyz - center_xyz.transpose(1, 2).unsqueeze(-1)
        if self.normalize_xyz:
            diff = diff / self.max_radius
        if features is not None:
            grouped = grouping_operation(features, idx)
            new_features = torch.cat([diff, grouped], dim=1) if self.use_xyz else grouped
        else:
            assert self.use_xyz, "Cannot have not features and not use xyz as a feature!"
            new_features = diff
        ret = [new_features]
        if self.return_grouped_xyz:
            ret.append(grouped_xyz)
        if self.return_grouped_idx:
            ret.append(idx)
        return ret[0] if len(ret) == 1 else tuple(ret)


class GroupAll(nn.Module):
    def __init__(self, use_xyz=True):
        super().__init__()
        self.use_xyz = use_xyz

    def forward(self, xyz, new_xyz, features=None):
        grouped_xyz = xyz.transpose(1, 2).unsqueeze(2)
        if features is None:
            return grouped_xyz
        grouped = features.unsqueeze(2)
        return torch.cat([grouped_xyz, grouped], dim=1) if self.use_xyz else grouped


class BasePointSAModule(nn.Module):
    def __init__(self, num_point, radii, sample_nums, mlp_channels, fps_mod=["D-FPS"], fps_sample_range_list=[-1],
                 dilated_group=False, use_xyz=True, pool_mod="max", normalize_xyz=False,
                 grouper_return_grouped_xyz=False, grouper_return_grouped_idx=False):
        super().__init__()
        assert len(radii) == len(sample_nums) == len(mlp_channels)
        assert pool_mod in ["max", "avg"]
        assert isinstance(fps_mod, (list, tuple)) and isinstance(fps_sample_range_list, (list, tuple))
        assert len(fps_mod) == len(fps_sample_range_list)
        if isinstance(mlp_channels, tuple):
            mlp_channels = list(map(list, mlp_channels))
        self.mlp_channels = mlp_channels
        if isinstance(num_point, int):
            self.num_point = [num_point]
        elif isinstance(num_point, (list, tuple)):
            self.num_point = num_point
        else:
            raise NotImplementedError("Error type of num_point!")
        self.pool_mod = pool_mod
        self.use_xyz = use_xyz
        self.normalize_xyz = normalize_xyz
        self.groupers = nn.ModuleList()
        self.mlps = nn.ModuleList()
        self.fps_mod_list = fps_mod
        self.fps_sample_range_list = fps_sample_range_list
        self.points_sampler = Points_Sampler(self.num_point, self.fps_mod_list, self.fps_sample_range_list)
        # evaluate the shared MLP only on the distinct rows of a ball-query group (identical output)
        self.skip_repeats = True
        for i in range(len(radii)):
            if num_point is not None:
                min_radius = radii[i - 1] if dilated_group and i != 0 else 0
                grouper = QueryAndGroup(radii[i], sample_nums[i], min_radius=min_radius, use_xyz=use_xyz,
                                        normalize_xyz=normalize_xyz, return_grouped_xyz=grouper_return_grouped_xyz,
                                        return_grouped_idx=grouper_return_grouped_idx)
            else:
                grouper = GroupAll(use_xyz)
            self.groupers.append(grouper)
        self._plans = {}

    def _sample_points(self, points_xyz, features, indices, target_xyz):
        xyz_flipped = points_xyz.transpose(1, 2).contiguous()
        if indices is not None:
            assert indices.shape[1] == self.num_point[0]
            new_xyz = gather_points(xyz_flipped, indices).transpose(1, 2).contiguous()
        elif target_xyz is not None:
            new_xyz = target_xyz.contiguous()
        else:
            indices = self.points_sampler(points_xyz, features)
            new_xyz = gather_points(xyz_flipped, indices).transpose(1, 2).contiguous()
        return new_xyz, indices

    def _plan(self, i, device):
        mlp = self.mlps[i]
        if self.training:
            raise L.PcrError("PointSAModule: the fused HIP path implements eval-mode inference; call .eval()")
        key = (str(device), engine.param_version(mlp))
        if self._plans.get(i, (None,))[0] != key:
            layers = list(mlp.children())
            if len(layers) != 3 or self.pool_mod != "max" or self.normalize_xyz or not self.use_xyz:
                raise L.PcrError("fused SA launch covers 3-layer MLPs with use_xyz, max pooling and un-normalised "
                                 "xyz (got %d layers, pool %s)" % (len(layers), self.pool_mod))
            self._plans[i] = (key, engine.SaPlan([l.conv for l in layers], [l.bn for l in layers], device, mode=1))
        return self._plans[i][1]

    def _fused_sample_and_query(self, points_xyz, features, indices, target_xyz):
        """D-FPS over the whole cloud + ONE ball-query scale whose SA kernel reads the row table: sampling, the centres'
        coordinates, hit counts and the row table come from one launch (ops.fps_ball_query_rows) -- the sampler's
        distances ARE the query's.  None: not that configuration (the caller runs the separate ops)."""
        if (_NO_FPS_BQ or indices is not None or target_xyz is not None or len(self.groupers) != 1 or
                not self.skip_repeats or _NO_ROW_TABLE or engine.PRECISION == "f32" or engine._LEVEL >= 1 or
                len(self.points_sampler.samplers) != 1 or not isinstance(self.points_sampler.samplers[0], DFPS_Sampler) or
                self.points_sampler.fps_sample_range_list[0] != -1):
            return None
        grouper = self.groupers[0]
        if not isinstance(grouper, QueryAndGroup) or grouper.max_radius is None or grouper.min_radius:
            return None
        B, N, _ = points_xyz.shape
        M, K = self.num_point[0], grouper.sample_num
        plan = self._plan(0, points_xyz.device)
        if not (fps_ball_query_rows_ok(N, M, K) and plan.wants_row_table(N, K, 0.0, B, M)):
            return None
        indices, new_xyz, cnt, rows = fps_ball_query_rows(points_xyz, M, grouper.max_radius, K)
        out = plan.run(points_xyz, features, None, centre_idx=indices, cnt=cnt, rows=rows, K=K, out_point_major=True)
        return new_xyz, out, indices

    def _scale(self, plan, grouper, points_xyz, new_xyz, features, indices):
        """one grouping scale under the current engine.PRECISION: neighbour query + grouped MLP + max"""
        if (self.skip_repeats and not _NO_ROW_TABLE and grouper.max_radius is not None and not grouper.min_radius and
                plan.wants_row_table(points_xyz.shape[1], grouper.sample_num, 0.0, points_xyz.shape[0], new_xyz.shape[1])):
            # the ball query hands the SA kernel its rows ready-made ({neighbour, point - centre}); no index tensor
            _, cnt, rows = ball_query_rows(grouper.max_radius, grouper.sample_num, points_xyz, new_xyz)
            return plan.run(points_xyz, features, None, centre_idx=indices.contiguous(), cnt=cnt, rows=rows,
                            K=grouper.sample_num, out_point_major=len(self.groupers) == 1)
        idx, cnt = grouper.query_cnt(points_xyz, new_xyz)
        # single-scale modules hand out the (B,C,S) view of a point-major buffer (a centre's channels are
        # stored as one run); SaPlan / engine.dense read either layout, anyone else may call .contiguous()
        return plan.run(points_xyz, features, idx, centre_idx=indices.contiguous(),
                        cnt=cnt if self.skip_repeats else None, out_point_major=len(self.groupers) == 1)

    def forward(self, points_xyz, features=None, indices=None, target_xyz=None):
        """points_xyz (B,N,3), features (B,C,N) -> new_xyz (B,M,3), new_features (B,sum C',M), indices (B,M)"""
        points_xyz = points_xyz.contiguous()
        fused = self._fused_sample_and_query(points_xyz, features, indices, target_xyz)
        if fused is not None:
            return fused
        new_xyz, indices = self._sample_points(points_xyz, features, indices, target_xyz)
        if indices is None:
            raise L.PcrError("the fused SA launch gathers centres by index; pass `indices` or let the sampler run")
        outs = []
        for i, grouper in enumerate(self.groupers):
            if not isinstance(grouper, QueryAndGroup):
                raise L.PcrError("GroupAll scales are not on the ReID path")
            plan = self._plan(i, points_xyz.device)
            # the launches of one scale (ball query in the form this arithmetic's SA kernel reads + the grouped MLP): these
            # layers carry folded BatchNorm scales -- f32 from guard level 1 on (engine.guarded)
            outs.append(engine.guarded(lambda: self._scale(plan, grouper, points_xyz, new_xyz, features, indices)))
        return new_xyz, torch.cat(outs, dim=1) if len(outs) > 1 else outs[0], indices


_NO_ROW_TABLE = bool(os.environ.get("PCR_NO_ROW_TABLE"))   # diagnostics: the indexed ragged launch instead
_NO_FPS_BQ = bool(os.environ.get("PCR_NO_FPS_BQ"))         # diagnostics: sampling and ball query as separate launches


@SA_MODULES.register_module()
class PointSAModuleMSG(BasePointSAModule):
    def __init__(self, num_point, radii, sample_nums, mlp_channels, fps_mod=["D-FPS"], fps_sample_range_list=[-1],
                 dilated_group=False, norm_cfg=dict(type="BN2d"), use_xyz=True, pool_mod="max", normalize_xyz=False,
                 bias="auto"):
        super().__init__(num_point=num_point, radii=radii, sample_nums=sample_nums, mlp_channels=mlp_channels,
                         fps_mod=fps_mod, fps_sample_range_list=fps_sample_range_list, dilated_group=dilated_group,
                         use_xyz=use_xyz, pool_mod=pool_mod, normalize_xyz=normalize_xyz)
        for i in range(len(self.mlp_channels)):
            mlp_channel = self.mlp_channels[i]
            if use_xyz:
                mlp_channel[0] += 3
            mlp = nn.Sequential()
            for j in range(len(mlp_channel) - 1):
                mlp.add_module("layer%d" % j, ConvModule(mlp_channel[j], mlp_channel[j + 1], kernel_size=(1, 1),
                                                         stride=(1, 1), conv_cfg=dict(type="Conv2d"),
                                                         norm_cfg=norm_cfg, bias=bias))
            self.mlps.append(mlp)


@SA_MODULES.register_module()
class PointSAModule(PointSAModuleMSG):
    def __init__(self, mlp_channels, num_point=None, radius=None, num_sample=None, norm_cfg=dict(type="BN2d"),
                 use_xyz=True, pool_mod="max", fps_mod=["D-FPS"], fps_sample_range_list=[-1], normalize_xyz=False):
        super().__init__(mlp_channels=[mlp_channels], num_point=num_point, radii=[radius], sample_nums=[num_sample],
                         norm_cfg=norm_cfg, use_xyz=use_xyz, pool_mod=pool_mod, fps_mod=fps_mod,
                         fps_sample_range_list=fps_sample_range_list, normalize_xyz=normalize_xyz)


class PointFPModule(nn.Module):
    """3-NN inverse-distance interpolation + concat + ConvModule stack (point_fp_module.py:39-79)"""

    def __init__(self, mlp_channels, norm_cfg=dict(type="BN2d"), init_cfg=None):
        super().__init__()
        self.fp16_enabled = False
        self.mlps = nn.Sequential()
        for i in range(len(mlp_channels) - 1):
            self.mlps.add_module("layer%d" % i, ConvModule(mlp_channels[i], mlp_channels[i + 1], kernel_size=(1, 1),
                                                          stride=(1, 1), conv_cfg=dict(type="Conv2d"),
                                                          norm_cfg=norm_cfg))
        self._plan_key = None
        self._plan = None

    def _packed(self, device):
        if self.training:
            raise L.PcrError("PointFPModule: the HIP path implements eval-mode inference; call .eval()")
        key = (str(device), engine.param_version(self.mlps))
        if self._plan_key != key:
            plan = []
            for layer in self.mlps.children():
                scale, shift = engine.fold_bn(layer.bn, layer.conv.bias, device)
                plan.append((engine.pack_weight(layer.conv.weight, device), layer.conv.weight.shape[0], scale, shift))
            self._plan, self._plan_key = plan, key
        return self._plan

    def forward(self, target, source, target_feats, source_feats):
        """target (B,n,3), source (B,m,3), target_feats (B,C1,n) or None, source_feats (B,C2,m) -> (B,M,n)"""
        if source is not None:
            dist, idx = three_nn(target.contiguous().float(), source.contiguous().float())
            dist_reciprocal = 1.0 / (dist + 1e-8)
            norm = torch.sum(dist_reciprocal, dim=2, keepdim=True)
            weight = (dist_reciprocal / norm).contiguous()
            interpolated = three_interpolate(source_feats.contiguous().float(), idx, weight)
        else:
            interpolated = source_feats.expand(*source_feats.size()[0:2], target.size(1))
        x = torch.cat([interpolated, target_feats], dim=1) if target_feats is not None else interpolated
        x = x.contiguous()
        for wp, cout, scale, shift in self._packed(x.device):
            x = engine.dense(x, wp, cout, scale, shift, act=1)
        return x


def build_sa_module(cfg, *args, **kwargs):
    if cfg is None:
        cfg_ = dict(type="PointSAModule")
    else:
        if not isinstance(cfg, dict):
            raise TypeError("cfg must be a dict")
        if "type" not in cfg:
            raise KeyError('the cfg dict must contain the key "type"')
        cfg_ = cfg.copy()
    module_type = cfg_.pop("type")
    if module_type not in SA_MODULES:
        raise KeyError("Unrecognized module type %s" % module_type)
    return SA_MODULES.get(module_type)(*args, **kwargs, **cfg_)
