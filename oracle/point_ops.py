"""TEST INFRASTRUCTURE ONLY -- numpy front-end of the C oracle (oracle/pcr_oracle.c).

Imported only by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libpcr_oracle.so")
_lib = None

_f = ctypes.POINTER(ctypes.c_float)
_i = ctypes.POINTER(ctypes.c_int)


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = ctypes.CDLL(_SO)
        _lib.pcr_oracle_fps_block.restype = ctypes.c_int
        _lib.pcr_oracle_knn.restype = ctypes.c_int
    return _lib


def _fp(a):
    return a.ctypes.data_as(_f)


def _ip(a):
    return a.ctypes.data_as(_i)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def fps_block(n):
    return lib().pcr_oracle_fps_block(int(n))


def fps(xyz, m):
    xyz = _f32(xyz)
    B, N, _ = xyz.shape
    temp = np.full((B, N), 1e10, np.float32)
    idx = np.zeros((B, m), np.int32)
    lib().pcr_oracle_fps(_fp(xyz), _fp(temp), _ip(idx), B, N, m)
    return idx


def fps_dist(dist, m):
    dist = _f32(dist)
    B, N, _ = dist.shape
    temp = np.full((B, N), 1e10, np.float32)
    idx = np.zeros((B, m), np.int32)
    lib().pcr_oracle_fps_dist(_fp(dist), _fp(temp), _ip(idx), B, N, m)
    return idx


def pairwise_sqdist(a, b, norm=False):
    a, b = _f32(a), _f32(b)
    B, N, C = a.shape
    M = b.shape[1]
    out = np.zeros((B, N, M), np.float32)
    lib().pcr_oracle_pairwise_sqdist(_fp(a), _fp(b), _fp(out), B, N, M, C, int(bool(norm)))
    return out


def ball_query(min_r, max_r, k, xyz, centres):
    xyz, centres = _f32(xyz), _f32(centres)
    B, N, _ = xyz.shape
    M = centres.shape[1]
    idx = np.zeros((B, M, k), np.int32)
    lib().pcr_oracle_ball_query(_fp(centres), _fp(xyz), _ip(idx), B, N, M,
                                ctypes.c_float(min_r), ctypes.c_float(max_r), k)
    return idx


def knn(k, xyz, centres):
    """returns (idx (B,M,K) int32, dist2 (B,M,K)) -- the kernel's native layout."""
    xyz, centres = _f32(xyz), _f32(centres)
    B, N, _ = xyz.shape
    M = centres.shape[1]
    idx = np.zeros((B, M, k), np.int32)
    d2 = np.zeros((B, M, k), np.float32)
    rc = lib().pcr_oracle_knn(_fp(xyz), _fp(centres), _ip(idx), _fp(d2), B, N, M, k)
    if rc:
        raise ValueError("knn: k must be in 1..100")
    return idx, d2


def knn_prefix(xyz, s, k):
    xyz = _f32(xyz)
    B, N, _ = xyz.shape
    idx = np.zeros((B, s, k), np.int32)
    lib().pcr_oracle_knn_prefix(_fp(xyz), _ip(idx), B, N, s, k)
    return idx


def knn_feat(x, k):
    """x (B,C,N) -> idx (B,N,k): DGCNN feature-space kNN in this build's pinned arithmetic order"""
    x = _f32(x)
    B, C, N = x.shape
    idx = np.zeros((B, N, k), np.int32)
    lib().pcr_oracle_knn_feat(_fp(x), _ip(idx), B, C, N, k)
    return idx


def gather_fwd(feat, idx):
    feat, idx = _f32(feat), _i32(idx)
    B, C, N = feat.shape
    M = idx.shape[1]
    out = np.zeros((B, C, M), np.float32)
    lib().pcr_oracle_gather_fwd(_fp(feat), _ip(idx), _fp(out), B, C, N, M)
    return out


def gather_bwd(grad_out, idx, n):
    grad_out, idx = _f32(grad_out), _i32(idx)
    B, C, M = grad_out.shape
    g = np.zeros((B, C, n), np.float32)
    lib().pcr_oracle_gather_bwd(_fp(grad_out), _ip(idx), _fp(g), B, C, n, M)
    return g


def group_fwd(feat, idx):
    feat, idx = _f32(feat), _i32(idx)
    B, C, N = feat.shape
    _, S, K = idx.shape
    out = np.zeros((B, C, S, K), np.float32)
    lib().pcr_oracle_group_fwd(_fp(feat), _ip(idx), _fp(out), B, C, N, S, K)
    return out


def group_bwd(grad_out, idx, n):
    grad_out, idx = _f32(grad_out), _i32(idx)
    B, C, S, K = grad_out.shape
    g = np.zeros((B, C, n), np.float32)
    lib().pcr_oracle_group_bwd(_fp(grad_out), _ip(idx), _fp(g), B, C, n, S, K)
    return g


def three_nn(unknown, known):
    """returns (dist2 (B,N,3) -- NOT sqrt'ed, idx (B,N,3))"""
    unknown, known = _f32(unknown), _f32(known)
    B, N, _ = unknown.shape
    M = known.shape[1]
    d2 = np.zeros((B, N, 3), np.float32)
    idx = np.zeros((B, N, 3), np.int32)
    lib().pcr_oracle_three_nn(_fp(unknown), _fp(known), _fp(d2), _ip(idx), B, N, M)
    return d2, idx


def three_interp_fwd(feat, idx, w):
    feat, idx, w = _f32(feat), _i32(idx), _f32(w)
    B, C, M = feat.shape
    N = idx.shape[1]
    out = np.zeros((B, C, N), np.float32)
    lib().pcr_oracle_three_interp_fwd(_fp(feat), _ip(idx), _fp(w), _fp(out), B, C, M, N)
    return out


def three_interp_bwd(grad_out, idx, w, m):
    grad_out, idx, w = _f32(grad_out), _i32(idx), _f32(w)
    B, C, N = grad_out.shape
    g = np.zeros((B, C, m), np.float32)
    lib().pcr_oracle_three_interp_bwd(_fp(grad_out), _ip(idx), _fp(w), _fp(g), B, C, N, m)
    return g
