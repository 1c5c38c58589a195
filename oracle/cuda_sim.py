"""TEST INFRASTRUCTURE ONLY -- a SECOND, independent transliteration of the reference's dormant CUDA point ops, used only
to generate tests/golden/ops_cuda_semantics.npz (oracle/make_golden.py gen_cuda_semantics).

oracle/pcr_oracle.c restates the *semantics* of the kernels (a per-lane best followed by a merge rule, a first-K scan,
a heap).  This file instead simulates their *execution*: one Python object per CUDA thread, the `__shared__` arrays as
arrays, every `__syncthreads()` as a phase boundary, the block / grid shape of the launchers -- so that the tie and
boundary rules the Python twins of the reference cannot pin (they differ by design: VERDICT r5 weak 4) come out of the
kernels' own control flow, not out of a reading of it.  The two transliterations share no code; where they agree on the
hand-built tie / boundary cases below, the C oracle's rules are pinned by a vector and not by one reviewer.

Arithmetic: binary32 throughout (numpy float32 scalars), the distance expression evaluated left to right without fma
contraction -- the floating-point contract of the C oracle and of the HIP kernels (oracle/pcr_oracle.c:16-20); the tie
cases use small integer / dyadic coordinates, for which that choice is immaterial (every product and sum is exact).

Reference sources simulated:
  mmdet3d/ops/furthest_point_sample/src/furthest_point_sample_cuda.cu:11-141 (xyz), :213-331 (distance matrix)
  mmdet3d/ops/ball_query/src/ball_query_cuda.cu:11-54
  mmdet3d/ops/knn/src/knn_cuda.cu:10-94
  mmdet3d/ops/interpolate/src/three_nn_cuda.cu:11-65
"""
import math

import numpy as np

F = np.float32


def launch_block_size(n):
    """opt_n_threads: 1 << int(log(n) / log(2)) clamped to [1, 1024] (the launcher's switch has a case for every power of
    two up to 1024, so the template block size equals the launched one)"""
    p = int(math.log(float(n)) / math.log(2.0))
    return max(min(1 << p, 1024), 1)


def _d2(a, b):
    dx, dy, dz = F(b[0] - a[0]), F(b[1] - a[1]), F(b[2] - a[2])
    return F(F(F(dx * dx) + F(dy * dy)) + F(dz * dz))


def fps_block(dataset, m, with_dist=False):
    """one thread block of furthest_point_sampling[_with_dist]_kernel: dataset (n,3) (or (n,n) distances), temp starts at
    1e10 (the Python wrapper's fill), -> idxs (m,) int32.  Threads run phase by phase; within a phase their order cannot
    matter (each touches only its own k's / its own shared slot), which the simulation checks by running them in a
    scrambled order."""
    n = dataset.shape[0]
    bs = launch_block_size(n)
    temp = np.full(n, F(1e10), dtype=F)
    idxs = np.zeros(m, dtype=np.int32)
    if m <= 0:
        return idxs
    sh_d = np.zeros(bs, dtype=F)            # __shared__ float dists[block_size]
    sh_i = np.zeros(bs, dtype=np.int32)     # __shared__ int dists_i[block_size]
    old = 0
    idxs[0] = old
    order = list(range(bs))
    order = order[1::2] + order[0::2]       # (any order: the phase is race free)
    for j in range(1, m):
        # ---- phase 1: every thread scans its strided candidates ----
        for tid in order:
            besti, best = 0, F(-1)
            for k in range(tid, n, bs):
                d = F(dataset[old, k]) if with_dist else _d2(dataset[old], dataset[k])
                d2 = d if d < temp[k] else temp[k]              # min(d, temp[k])
                temp[k] = d2
                if d2 > best:
                    besti, best = k, d2
            sh_d[tid], sh_i[tid] = best, besti
        # ---- phases 2..: the halving tree, one __syncthreads() per level ----
        s = bs // 2
        while s >= 1:
            nd, ni = sh_d.copy(), sh_i.copy()
            for tid in range(s):                                # if (tid < s) __update(dists, dists_i, tid, tid + s)
                v1, v2 = sh_d[tid], sh_d[tid + s]
                i1, i2 = sh_i[tid], sh_i[tid + s]
                nd[tid] = v1 if v1 > v2 else v2                 # max(v1, v2)
                ni[tid] = i2 if v2 > v1 else i1
            sh_d, sh_i = nd, ni
            s //= 2
        old = int(sh_i[0])
        idxs[j] = old
    return idxs


def fps(xyz, m):
    """grid = B blocks: (B,N,3) -> (B,m) int32"""
    return np.stack([fps_block(np.asarray(c, dtype=F), m) for c in xyz])


def fps_with_dist(dist, m):
    return np.stack([fps_block(np.asarray(c, dtype=F), m, with_dist=True) for c in dist])


def ball_query(min_r, max_r, nsample, xyz, new_xyz):
    """ball_query_kernel, one thread per (batch, centre); idx zero-initialised by the wrapper (ball_query.py:41)"""
    xyz, new_xyz = np.asarray(xyz, dtype=F), np.asarray(new_xyz, dtype=F)
    B, N, _ = xyz.shape
    M = new_xyz.shape[1]
    idx = np.zeros((B, M, nsample), dtype=np.int32)
    max2 = F(F(max_r) * F(max_r))
    min2 = F(F(min_r) * F(min_r))
    for b in range(B):
        for pt in range(M):
            c = new_xyz[b, pt]
            cnt = 0
            for k in range(N):
                p = xyz[b, k]
                dx, dy, dz = F(c[0] - p[0]), F(c[1] - p[1]), F(c[2] - p[2])
                d2 = F(F(F(dx * dx) + F(dy * dy)) + F(dz * dz))
                if d2 == 0 or (d2 >= min2 and d2 < max2):
                    if cnt == 0:
                        for l in range(nsample):
                            idx[b, pt, l] = k
                    idx[b, pt, cnt] = k
                    cnt += 1
                    if cnt >= nsample:
                        break
    return idx


def _reheap(dist, idx, k):
    root = 0
    child = root * 2 + 1
    while child < k:
        if child + 1 < k and dist[child + 1] > dist[child]:
            child += 1
        if dist[root] > dist[child]:
            return
        dist[root], dist[child] = dist[child], dist[root]
        idx[root], idx[child] = idx[child], idx[root]
        root = child
        child = root * 2 + 1


def knn(nsample, xyz, new_xyz):
    """knn_kernel, one thread per (batch, query): -> idx (B,M,nsample) int32, dist2 (B,M,nsample) float32"""
    xyz, new_xyz = np.asarray(xyz, dtype=F), np.asarray(new_xyz, dtype=F)
    B, N, _ = xyz.shape
    M = new_xyz.shape[1]
    out_i = np.zeros((B, M, nsample), dtype=np.int32)
    out_d = np.zeros((B, M, nsample), dtype=F)
    for b in range(B):
        for pt in range(M):
            q = new_xyz[b, pt]
            bd = [F(1e10)] * nsample
            bi = [0] * nsample
            for i in range(N):
                p = xyz[b, i]
                dx, dy, dz = F(q[0] - p[0]), F(q[1] - p[1]), F(q[2] - p[2])
                d2 = F(F(F(dx * dx) + F(dy * dy)) + F(dz * dz))
                if d2 < bd[0]:
                    bd[0], bi[0] = d2, i
                    _reheap(bd, bi, nsample)
            for i in range(nsample - 1, 0, -1):                 # heap_sort
                bd[0], bd[i] = bd[i], bd[0]
                bi[0], bi[i] = bi[i], bi[0]
                _reheap(bd, bi, i)
            out_i[b, pt], out_d[b, pt] = bi, bd
    return out_i, out_d


def three_nn(unknown, known):
    """three_nn_kernel: float distance compared against DOUBLE running bests (1e40 start), strict '<' cascade;
    -> dist2 (B,n,3) float32 (the kernel stores the squared distances; the wrapper takes the sqrt), idx (B,n,3) int32"""
    unknown, known = np.asarray(unknown, dtype=F), np.asarray(known, dtype=F)
    B, n, _ = unknown.shape
    m = known.shape[1]
    d_out = np.zeros((B, n, 3), dtype=F)
    i_out = np.zeros((B, n, 3), dtype=np.int32)
    for b in range(B):
        for pt in range(n):
            u = unknown[b, pt]
            best1 = best2 = best3 = float(1e40)
            i1 = i2 = i3 = 0
            for k in range(m):
                p = known[b, k]
                dx, dy, dz = F(u[0] - p[0]), F(u[1] - p[1]), F(u[2] - p[2])
                d = float(F(F(F(dx * dx) + F(dy * dy)) + F(dz * dz)))
                if d < best1:
                    best3, i3 = best2, i2
                    best2, i2 = best1, i1
                    best1, i1 = d, k
                elif d < best2:
                    best3, i3 = best2, i2
                    best2, i2 = d, k
                elif d < best3:
                    best3, i3 = d, k
            with np.errstate(over="ignore"):
                d_out[b, pt] = (F(best1), F(best2), F(best3))   # double 1e40 -> float: +inf where fewer than 3 points exist
            i_out[b, pt] = (i1, i2, i3)
    return d_out, i_out
