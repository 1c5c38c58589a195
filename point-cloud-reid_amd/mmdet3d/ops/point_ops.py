"""`mmdet3d.ops` point-op functions with the reference's names, argument meaning and error
behaviour, as torch.autograd.Functions over the C ABI in include/pcr.h.

Reference wrappers mirrored (bentherien/point-cloud-reid, mmdet3d/ops/...):
  furthest_point_sample/furthest_point_sample.py:7-78, ball_query/ball_query.py:7-54,
  knn/knn.py:7-71, gather_points/gather_points.py:7-50, group_points/group_points.py:169-220,
  interpolate/three_nn.py:8-45, interpolate/three_interpolate.py:8-59.
Index outputs are int32 and non-differentiable; index ops return None gradients.
"""
import ctypes

import torch
from torch.autograd import Function

from pcr_amd import _lib as L
from pcr_amd.engine import _prof


def _on_device(fn):
    """run a Function's forward / backward with the tensors' device current (a tensor may live on another GPU than
    the caller's current one; the reference's KNN wrapper does the same with torch.cuda.device_of, knn.py:47)"""
    def wrapped(ctx, *args):
        dev = next((a.device for a in args if isinstance(a, torch.Tensor) and a.is_cuda), None)
        if dev is None:
            return fn(ctx, *args)
        with torch.cuda.device(dev):
            return fn(ctx, *args)
    wrapped.__name__ = fn.__name__
    wrapped.__doc__ = fn.__doc__
    return staticmethod(wrapped)


def _i32(*shape, device):
    return torch.empty(shape, dtype=torch.int32, device=device)


def _f32(*shape, device):
    return torch.empty(shape, dtype=torch.float32, device=device)


class FurthestPointSampling(Function):
    @_on_device
    def forward(ctx, points_xyz, num_points):
        assert points_xyz.is_contiguous()
        L.require_cuda(points_xyz)
        L.require_f32(points_xyz)
        B, N = points_xyz.size()[:2]
        out = _i32(B, num_points, device=points_xyz.device)
        temp = torch.full((B, N), 1e10, dtype=torch.float32, device=points_xyz.device)
        with _prof("fps[N=%d,M=%d]" % (N, num_points), 8.0 * B * N * num_points, 4.0 * B * (3 * N + num_points)):
            L.check(L.load().pcr_fps_f32(L.ptr(points_xyz), L.ptr(temp), L.ptr(out), B, N, num_points,
                                         L.stream_ptr()), "pcr_fps_f32")
        ctx.mark_non_differentiable(out)
        return out

    @_on_device
    def backward(ctx, a=None):
        return None, None


class FurthestPointSamplingWithDist(Function):
    @_on_device
    def forward(ctx, points_dist, num_points):
        assert points_dist.is_contiguous()
        L.require_cuda(points_dist)
        L.require_f32(points_dist)
        B, N, _ = points_dist.size()
        out = _i32(B, num_points, device=points_dist.device)
        temp = torch.full((B, N), 1e10, dtype=torch.float32, device=points_dist.device)
        L.check(L.load().pcr_fps_dist_f32(L.ptr(points_dist), L.ptr(temp), L.ptr(out), B, N, num_points,
                                          L.stream_ptr()), "pcr_fps_dist_f32")
        ctx.mark_non_differentiable(out)
        return out

    @_on_device
    def backward(ctx, a=None):
        return None, None


class BallQuery(Function):
    @_on_device
    def forward(ctx, min_radius, max_radius, sample_num, xyz, center_xyz):
        assert center_xyz.is_contiguous()
        assert xyz.is_contiguous()
        assert min_radius < max_radius
        L.require_cuda(xyz, center_xyz)
        L.require_f32(xyz, center_xyz)
        B, N, _ = xyz.size()
        npoint = center_xyz.size(1)
        idx = _i32(B, npoint, sample_num, device=xyz.device)
        with _prof("ball_query[N=%d,M=%d,K=%d]" % (N, npoint, sample_num), 8.0 * B * N * npoint,
                   4.0 * B * (3 * N + 3 * npoint + npoint * sample_num)):
            L.check(L.load().pcr_ball_query_f32(L.ptr(center_xyz), L.ptr(xyz), L.ptr(idx), B, N, npoint,
                                                ctypes.c_float(min_radius), ctypes.c_float(max_radius),
                                                sample_num, L.stream_ptr()), "pcr_ball_query_f32")
        ctx.mark_non_differentiable(idx)
        return idx

    @_on_device
    def backward(ctx, a=None):
        return None, None, None, None, None


class BallQueryCnt(Function):
    """ball_query that also returns the number of genuine hits of every row (entries past it repeat the
    first hit, ball_query_cuda.cu:43-47); lets the fused SA kernel skip the repeated rows"""

    @_on_device
    def forward(ctx, min_radius, max_radius, sample_num, xyz, center_xyz):
        assert center_xyz.is_contiguous()
        assert xyz.is_contiguous()
        assert min_radius < max_radius
        L.require_cuda(xyz, center_xyz)
        L.require_f32(xyz, center_xyz)
        B, N, _ = xyz.size()
        npoint = center_xyz.size(1)
        idx = _i32(B, npoint, sample_num, device=xyz.device)
        cnt = _i32(B, npoint, device=xyz.device)
        with _prof("ball_query[N=%d,M=%d,K=%d]" % (N, npoint, sample_num), 8.0 * B * N * npoint,
                   4.0 * B * (3 * N + 3 * npoint + npoint * sample_num)):
            L.check(L.load().pcr_ball_query_cnt_f32(L.ptr(center_xyz), L.ptr(xyz), L.ptr(idx), L.ptr(cnt), B, N,
                                                    npoint, ctypes.c_float(min_radius), ctypes.c_float(max_radius),
                                                    sample_num, L.stream_ptr()), "pcr_ball_query_cnt_f32")
        ctx.mark_non_differentiable(idx, cnt)
        return idx, cnt

    @_on_device
    def backward(ctx, a=None, b=None):
        return None, None, None, None, None


def ball_query_rows(max_radius, sample_num, xyz, center_xyz, want_idx=False):
    """ball query (min_radius 0) -> (idx or None, cnt, rows): hit counts and the compact row table that the
    wave-autonomous ragged SA kernel reads (include/pcr.h pcr_ball_query_rows_f32); idx only on request"""
    assert center_xyz.is_contiguous() and xyz.is_contiguous()
    B, N, _ = xyz.size()
    npoint = center_xyz.size(1)
    with torch.cuda.device(xyz.device):
        L.require_cuda(xyz, center_xyz)
        L.require_f32(xyz, center_xyz)
        idx = _i32(B, npoint, sample_num, device=xyz.device) if want_idx else None
        cnt = _i32(B, npoint, device=xyz.device)
        rows = torch.empty((L.load().pcr_ball_query_rows_floats(B, npoint, sample_num),), dtype=torch.float32,
                           device=xyz.device)
        with _prof("ball_query[N=%d,M=%d,K=%d]" % (N, npoint, sample_num), 8.0 * B * N * npoint,
                   4.0 * B * (3 * N + 3 * npoint + npoint * sample_num)):
            L.check(L.load().pcr_ball_query_rows_f32(L.ptr(center_xyz), L.ptr(xyz), L.ptr(idx), L.ptr(cnt), L.ptr(rows),
                                                     B, N, npoint, ctypes.c_float(0.0), ctypes.c_float(max_radius),
                                                     sample_num, L.stream_ptr()), "pcr_ball_query_rows_f32")
    return idx, cnt, rows


def fps_ball_query_rows_ok(N, M, K):
    return bool(L.load().pcr_fps_ball_query_rows_ok(int(N), int(M), int(K)))


def fps_ball_query_rows(xyz, num_points, max_radius, sample_num):
    """D-FPS of `num_points` centres AND the ball query of those centres in one launch (pcr_fps_ball_query_rows_f32: a
    pick's distances to the cloud are the ball query's distances of that centre) -> (indices (B,M) int32, new_xyz (B,M,3),
    cnt (B,M), rows): what furthest_point_sample + gather_points + ball_query_rows return, entry for entry"""
    assert xyz.is_contiguous()
    B, N, _ = xyz.size()
    with torch.cuda.device(xyz.device):
        L.require_cuda(xyz)
        L.require_f32(xyz)
        dev = xyz.device
        idx, cnt = _i32(B, num_points, device=dev), _i32(B, num_points, device=dev)
        new_xyz = torch.empty((B, num_points, 3), dtype=torch.float32, device=dev)
        temp = torch.full((B, N), 1e10, dtype=torch.float32, device=dev)
        rows = torch.empty((L.load().pcr_ball_query_rows_floats(B, num_points, sample_num),), dtype=torch.float32, device=dev)
        with _prof("fps_ball_query[N=%d,M=%d,K=%d]" % (N, num_points, sample_num), 8.0 * B * N * num_points,
                   4.0 * B * (3 * N + 4 * num_points + num_points * sample_num)):
            L.check(L.load().pcr_fps_ball_query_rows_f32(L.ptr(xyz), L.ptr(temp), L.ptr(idx), L.ptr(new_xyz), L.ptr(cnt),
                                                         L.ptr(rows), B, N, num_points, ctypes.c_float(max_radius),
                                                         sample_num, L.stream_ptr()), "pcr_fps_ball_query_rows_f32")
    return idx, new_xyz, cnt, rows


class KNN(Function):
    @_on_device
    def forward(ctx, k, xyz, center_xyz=None, transposed=False):
        assert k > 0
        if center_xyz is None:
            center_xyz = xyz
        if transposed:
            xyz = xyz.transpose(2, 1).contiguous()
            center_xyz = center_xyz.transpose(2, 1).contiguous()
        assert xyz.is_contiguous()
        assert center_xyz.is_contiguous()
        assert center_xyz.device == xyz.device, "center_xyz and xyz should be put on the same device"
        L.require_cuda(xyz)
        L.require_f32(xyz, center_xyz)
        B, npoint, _ = center_xyz.shape
        N = xyz.shape[1]
        idx = _i32(B, npoint, k, device=xyz.device)
        dist2 = _f32(B, npoint, k, device=xyz.device)
        L.check(L.load().pcr_knn_f32(L.ptr(xyz), L.ptr(center_xyz), L.ptr(idx), L.ptr(dist2), B, N,
                                     npoint, k, L.stream_ptr()), "pcr_knn_f32")
        idx = idx.transpose(2, 1).contiguous()      # (B, k, npoint) as in knn.py:62
        ctx.mark_non_differentiable(idx)
        return idx

    @_on_device
    def backward(ctx, a=None):
        return None, None, None, None


class GatherPoints(Function):
    @_on_device
    def forward(ctx, features, indices):
        assert features.is_contiguous()
        assert indices.is_contiguous()
        L.require_cuda(features, indices)
        L.require_f32(features)
        L.require_i32(indices)
        B, npoint = indices.size()
        _, C, N = features.size()
        out = _f32(B, C, npoint, device=features.device)
        L.check(L.load().pcr_gather_fwd_f32(L.ptr(features), L.ptr(indices), L.ptr(out), B, C, N, npoint,
                                            L.stream_ptr()), "pcr_gather_fwd_f32")
        ctx.for_backwards = (indices, C, N)
        ctx.mark_non_differentiable(indices)
        return out

    @_on_device
    def backward(ctx, grad_out):
        idx, C, N = ctx.for_backwards
        B, npoint = idx.size()
        grad_features = torch.zeros(B, C, N, dtype=torch.float32, device=grad_out.device)
        g = grad_out.data.float().contiguous()
        L.check(L.load().pcr_gather_bwd_f32(L.ptr(g), L.ptr(idx), L.ptr(grad_features), B, C, N, npoint,
                                            L.stream_ptr()), "pcr_gather_bwd_f32")
        return grad_features, None


class GroupingOperation(Function):
    @_on_device
    def forward(ctx, features, indices):
        assert features.is_contiguous()
        assert indices.is_contiguous()
        L.require_cuda(features, indices)
        L.require_f32(features)
        L.require_i32(indices)
        B, nfeatures, nsample = indices.size()
        _, C, N = features.size()
        out = _f32(B, C, nfeatures, nsample, device=features.device)
        L.check(L.load().pcr_group_fwd_f32(L.ptr(features), L.ptr(indices), L.ptr(out), B, C, N, nfeatures,
                                           nsample, L.stream_ptr()), "pcr_group_fwd_f32")
        ctx.for_backwards = (indices, N)
        return out

    @_on_device
    def backward(ctx, grad_out):
        idx, N = ctx.for_backwards
        B, C, npoint, nsample = grad_out.size()
        grad_features = torch.zeros(B, C, N, dtype=torch.float32, device=grad_out.device)
        g = grad_out.data.float().contiguous()
        L.check(L.load().pcr_group_bwd_f32(L.ptr(g), L.ptr(idx), L.ptr(grad_features), B, C, N, npoint,
                                           nsample, L.stream_ptr()), "pcr_group_bwd_f32")
        return grad_features, None


class ThreeNN(Function):
    @_on_device
    def forward(ctx, target, source):
        assert target.is_contiguous()
        assert source.is_contiguous()
        L.require_cuda(target, source)
        L.require_f32(target, source)
        B, N, _ = target.size()
        m = source.size(1)
        dist2 = _f32(B, N, 3, device=target.device)
        idx = _i32(B, N, 3, device=target.device)
        L.check(L.load().pcr_three_nn_f32(L.ptr(target), L.ptr(source), L.ptr(dist2), L.ptr(idx), B, N, m,
                                          L.stream_ptr()), "pcr_three_nn_f32")
        ctx.mark_non_differentiable(idx)
        return torch.sqrt(dist2), idx

    @_on_device
    def backward(ctx, a=None, b=None):
        return None, None


class ThreeInterpolate(Function):
    @_on_device
    def forward(ctx, features, indices, weight):
        assert features.is_contiguous()
        assert indices.is_contiguous()
        assert weight.is_contiguous()
        L.require_cuda(features, indices, weight)
        L.require_f32(features, weight)
        L.require_i32(indices)
        B, c, m = features.size()
        n = indices.size(1)
        ctx.three_interpolate_for_backward = (indices, weight, m)
        out = _f32(B, c, n, device=features.device)
        L.check(L.load().pcr_three_interp_fwd_f32(L.ptr(features), L.ptr(indices), L.ptr(weight), L.ptr(out),
                                                  B, c, m, n, L.stream_ptr()), "pcr_three_interp_fwd_f32")
        return out

    @_on_device
    def backward(ctx, grad_out):
        idx, weight, m = ctx.three_interpolate_for_backward
        B, c, n = grad_out.size()
        grad_features = torch.zeros(B, c, m, dtype=torch.float32, device=grad_out.device)
        g = grad_out.data.float().contiguous()
        L.check(L.load().pcr_three_interp_bwd_f32(L.ptr(g), L.ptr(idx), L.ptr(weight), L.ptr(grad_features),
                                                  B, c, n, m, L.stream_ptr()), "pcr_three_interp_bwd_f32")
        return grad_features, None, None


furthest_point_sample = FurthestPointSampling.apply
furthest_point_sample_with_dist = FurthestPointSamplingWithDist.apply
ball_query = BallQuery.apply
ball_query_cnt = BallQueryCnt.apply
knn = KNN.apply
gather_points = GatherPoints.apply
grouping_operation = GroupingOperation.apply
three_nn = ThreeNN.apply
three_interpolate = ThreeInterpolate.apply
