// DGCNN EdgeConv path for gfx950 (reference: mmdet3d/models/dgcnn_orig.py -- knn :22-29, get_graph_feature
// :32-56, DGCNN.forward :127-152).  Built with -ffp-contract=off: the neighbour indices are bit-exact against
// oracle/pcr_oracle.c:pcr_oracle_knn_feat.
//
// The reference materialises a (B,2C,N,k) edge tensor cat[f_j - f_i, f_i] and runs a 1x1 Conv2d + BatchNorm +
// LeakyReLU + max over k on it.  Here one EdgeConv layer is three launches and the edge tensor never exists:
//   1. pcr_knn_feat_f32   feature-space kNN: 32 queries x N distances per workgroup on the matrix core
//                         (v_mfma_f32_32x32x2_f32, exact fmaf chain over channels), staged in LDS as sortable keys,
//                         then one wave per query selects the k largest (same threshold + rank scheme as the xyz kNN);
//   2. pcr_dense_pm_f32   (twice) per-point tables  A = (s.W1) f  and  Bt = (s.(W2 - W1)) f,  point-major, with the
//                         BatchNorm scale s folded into the weights:  s.W [f_j - f_i; f_i] = A_j + Bt_i;
//   3. pcr_edge_max_f32   out_i = leaky(max_j A_j + Bt_i + shift): fl(a + b) is monotone in a and LeakyReLU is
//                         increasing, so the max over the k neighbours commutes with everything after the gather --
//                         k times fewer matrix flops than the reference's layout and a gather of k contiguous rows.
#include <math.h>
#include <stdlib.h>

#include "pcr_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kEdgeThreads = 256;
constexpr int kKnnThreads = 512;   // 8 waves: at N = 1024 the key tile leaves room for ONE workgroup per CU
constexpr int kSelCap = 128;       // candidates per query the fast ranking path takes (2 per lane)
constexpr int kKB = 16;            // B-operand values (k-blocks) fetched per batch, one batch ahead of the MFMAs

// xx[b][i] = ((x0^2 + x1^2) + x2^2) + ...  (separately rounded squares, channel order)
__global__ __launch_bounds__(256) void feat_sqnorm_kernel(const float *__restrict__ x, float *__restrict__ xx, int C,
                                                          int N, long bstride) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= N) return;
  const float *f = x + (size_t)blockIdx.y * bstride + i;
  float s = 0.f;
  for (int c = 0; c < C; c++) {
    const float v = f[(size_t)c * N];
    const float sq = v * v;
    s = c == 0 ? sq : s + sq;
  }
  xx[(size_t)blockIdx.y * N + i] = s;
}

// keys: smaller key = larger pd (so "k largest" is the same "k smallest keys" selection as the xyz kNN)
__device__ __forceinline__ uint32_t desc_key(float pd) { return ~pcr_orderable(pd + 0.f); }

// One wave: the K smallest (key, index) pairs of d[t] <-> index lane + 64 t, written in (key, index) order.
template <int T>
__device__ __forceinline__ void select_k(uint32_t (&d)[T], int lane, int K, unsigned long long *cand,
                                         int *__restrict__ out) {
  uint32_t m = 0xFFFFFFFFu;
#pragma unroll
  for (int t = 0; t < T; t++) m = d[t] < m ? d[t] : m;
  // 1. the lane whose (truncated, lane-tagged) minimum has rank K-1 gives a threshold with >= K keys under it
  const uint32_t mkey = (m & ~63u) | (uint32_t)lane;
  const uint32_t tau = (uint32_t)__builtin_amdgcn_readlane((int)pcr_wave_sort_u32(mkey, lane), K - 1) | 63u;   // (pcr_common.h)
  // 2. candidates d <= tau, compacted in index order
  int cnt = 0;
#pragma unroll
  for (int t = 0; t < T; t++) cnt += d[t] <= tau ? 1 : 0;
  int incl = cnt;
#pragma unroll
  for (int s2 = 1; s2 < 64; s2 <<= 1) {
    const int o = __shfl_up(incl, s2, 64);
    if (lane >= s2) incl += o;
  }
  const int total = __builtin_amdgcn_readlane(incl, 63);
  if (total <= kSelCap) {
    int off = incl - cnt;
#pragma unroll
    for (int t = 0; t < T; t++)
      if (d[t] <= tau) cand[off++] = ((unsigned long long)d[t] << 32) | (unsigned)(lane + 64 * t);
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    if (total <= 64) {
      const unsigned long long own = pcr_wave_sort_u64(lane < total ? cand[lane] : ~0ull, lane);
      if (lane < K) out[lane] = (int)(uint32_t)own;
    } else {
      unsigned long long own[kSelCap / 64];
      int rk[kSelCap / 64];
#pragma unroll
      for (int u = 0; u < kSelCap / 64; u++) {
        own[u] = lane + 64 * u < total ? cand[lane + 64 * u] : ~0ull;
        rk[u] = 0;
      }
      for (int j = 0; j < total; j++) {
        const unsigned long long cj = cand[j];
#pragma unroll
        for (int u = 0; u < kSelCap / 64; u++) rk[u] += cj < own[u] ? 1 : 0;
      }
#pragma unroll
      for (int u = 0; u < kSelCap / 64; u++)
        if (lane + 64 * u < total && rk[u] < K) out[rk[u]] = (int)(own[u] & 0xFFFFFFFFull);
    }
    __builtin_amdgcn_wave_barrier();
  } else {
    // heavily tied rows: K rounds of wave-wide argmin over (key, index)
    int mine = 0;
    for (int k = 0; k < K; k++) {
      uint32_t bd = d[0];
      int bt = 0;
#pragma unroll
      for (int t = 1; t < T; t++)
        if (d[t] < bd) { bd = d[t]; bt = t; }
      unsigned long long key = ((unsigned long long)bd << 32) | (unsigned)(lane + 64 * bt);
      key = pcr_wave_min_u64(key);
      const int win = (int)(key & 0xFFFFFFFFull);
      if (lane == k) mine = win;
      if (lane == (win & 63)) {
        const int wt = win >> 6;
#pragma unroll
        for (int t = 0; t < T; t++) d[t] = (t == wt) ? 0xFFFFFFFFu : d[t];
      }
    }
    if (lane < K) out[lane] = mine;
  }
}

// x (B,C,N) channel-major (batch stride bstride floats), xx (B,N) -> idx (B,N,K).
// Workgroup = QT queries (rows of a 32-row MFMA tile; QT = 16 wastes half of it but halves the key tile so that
// N = 2048 fits in LDS) x all N points.  LDS: sA [C2*2][32] query operands | sxq [32] | keys [QT][NP] | cand.
template <int T>   // NP = 64 T >= N
__global__ __launch_bounds__(kKnnThreads) void knn_feat_kernel(const float *__restrict__ x,
                                                                const float *__restrict__ xx,
                                                                int *__restrict__ idx, int C, int N, int K, int QT,
                                                                long bstride) {
  constexpr int NP = 64 * T;
  constexpr int NW = kKnnThreads / 64;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int C2 = (C + 1) >> 1;
  float *sA = smem;                         // [2*C2][32]
  float *sxq = sA + 2 * C2 * 32;            // [32]
  uint32_t *keys = reinterpret_cast<uint32_t *>(sxq + 32);   // [QT][NP]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, h = lane >> 5;
  unsigned long long *cand = reinterpret_cast<unsigned long long *>(keys + (size_t)QT * NP) + wave * kSelCap;
  const size_t b = blockIdx.y;
  const float *f = x + b * bstride;
  const float *xxb = xx + b * N;
  const int q0 = blockIdx.x * QT;
  for (int e = tid; e < 2 * C2 * 32; e += kKnnThreads) {
    const int c = e >> 5, r = e & 31;
    const int q = q0 + r < N ? q0 + r : N - 1;
    sA[e] = c < C ? f[(size_t)c * N + q] : 0.f;
  }
  if (tid < 32) sxq[tid] = xxb[q0 + tid < N ? q0 + tid : N - 1];
  const int nblk = (N + 31) >> 5;
  // columns [nblk*32, NP) are not touched by the MFMA loop
  for (int e = tid; e < QT * (NP - nblk * 32); e += kKnnThreads) {
    const int w = NP - nblk * 32;
    const int row = e / w, col = nblk * 32 + e - row * w;
    keys[(size_t)row * NP + col] = 0xFFFFFFFFu;
  }
  __syncthreads();
  // One flat sequence of batches (column block jb, k-blocks [kk, kk + kKB)): the B values of batch i+1 are
  // requested before the MFMAs of batch i, across column-block boundaries too.
  auto fetch = [&](float (&v)[kKB], int jb, int kk) {
    const int j = jb * 32 + l31;
    const float *bp = f + (j < N ? j : N - 1);
#pragma unroll
    for (int u = 0; u < kKB; u++) {
      const int c = 2 * (kk + u) + h;
      const int cc = c < C ? c : C - 1;
      const float t = bp[(size_t)cc * N];
      v[u] = c < C ? t : 0.f;
    }
  };
  float cur[kKB], nxt[kKB];
  int jb = wave, kk = 0;
  if (jb < nblk) fetch(cur, jb, 0);
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; r++) acc[r] = 0.f;
  while (jb < nblk) {
    int njb = jb, nkk = kk + kKB;
    if (nkk >= C2) { njb = jb + NW; nkk = 0; }
    if (njb < nblk) fetch(nxt, njb, nkk);
    const int lim = C2 - kk < kKB ? C2 - kk : kKB;
    const float *ap = sA + (2 * kk + h) * 32 + l31;
    if (lim == kKB) {
#pragma unroll
      for (int u = 0; u < kKB; u++) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[u * 64], cur[u], acc, 0, 0, 0);
    } else {
#pragma unroll
      for (int u = 0; u < kKB; u++)
        if (u < lim) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[u * 64], cur[u], acc, 0, 0, 0);
    }
    if (nkk == 0) {   // last batch of column block jb: keys out, accumulator reset
      const int j = jb * 32 + l31;
      const float xj = xxb[j < N ? j : N - 1];
#pragma unroll
      for (int r = 0; r < 16; r++) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
        if (row < QT) {
          const float t2 = 2.f * acc[r] - sxq[row];
          const float pd = t2 - xj;
          keys[(size_t)row * NP + j] = j < N ? desc_key(pd) : 0xFFFFFFFFu;
        }
        acc[r] = 0.f;
      }
    }
#pragma unroll
    for (int u = 0; u < kKB; u++) cur[u] = nxt[u];
    jb = njb;
    kk = nkk;
  }
  __syncthreads();
  for (int row = wave; row < QT; row += NW) {
    const int q = q0 + row;
    if (q >= N) break;
    uint32_t d[T];
#pragma unroll
    for (int t = 0; t < T; t++) d[t] = keys[(size_t)row * NP + lane + 64 * t];
    select_k<T>(d, lane, K, cand, idx + (b * N + q) * K);
  }
}

// out[b][c][i] = leaky(max_j ta[b][idx[b][i][j]][c] + tb[b][i][c] + shift[c]); ta, tb (B,N,Co) point-major;
// out / out2 channel-major with their own batch strides (out2 optional: the slice of the concatenated buffer).
struct EdgeMaxArgs {
  const float *ta, *tb, *shift;
  const int *idx;
  float *out, *out2;
  long out_bs, out2_bs;
  int N, Co, K;
  float slope;
};

__global__ __launch_bounds__(kEdgeThreads) void edge_max_kernel(const EdgeMaxArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int N = a.N, Co = a.Co, K = a.K;
  int *sidx = reinterpret_cast<int *>(smem);   // [32][K]
  float *so = smem + 32 * K;                   // [Co][33]
  const int tid = threadIdx.x;
  const size_t b = blockIdx.y;
  const int p0 = blockIdx.x * 32;
  const int np = N - p0 < 32 ? N - p0 : 32;
  const int *ib = a.idx + (b * N + p0) * K;
  for (int e = tid; e < np * K; e += kEdgeThreads) sidx[e] = ib[e];
  __syncthreads();
  const float *ta = a.ta + b * N * Co;
  const float *tb = a.tb + (b * N + p0) * Co;
  int p = tid / Co, c = tid - p * Co;
  const int dp = kEdgeThreads / Co, dc = kEdgeThreads - dp * Co;   // e += 256 without a division per element
  for (; p < np; ) {
    const int *nb = sidx + p * K;
    float m = -INFINITY;
    int k = 0;
    for (; k + 4 <= K; k += 4) {
      const float v0 = ta[(size_t)nb[k] * Co + c], v1 = ta[(size_t)nb[k + 1] * Co + c];
      const float v2 = ta[(size_t)nb[k + 2] * Co + c], v3 = ta[(size_t)nb[k + 3] * Co + c];
      m = fmaxf(fmaxf(m, fmaxf(v0, v1)), fmaxf(v2, v3));
    }
    for (; k < K; k++) m = fmaxf(m, ta[(size_t)nb[k] * Co + c]);
    const float s = m + tb[(size_t)p * Co + c];
    const float y = s + a.shift[c];
    so[c * 33 + p] = y > 0.f ? y : y * a.slope;
    p += dp;
    c += dc;
    if (c >= Co) { c -= Co; p++; }
  }
  __syncthreads();
  for (int e = tid; e < Co * 32; e += kEdgeThreads) {
    const int cc = e >> 5, pp = e & 31;
    if (pp < np) {
      const float v = so[cc * 33 + pp];
      a.out[b * a.out_bs + (size_t)cc * N + p0 + pp] = v;
      if (a.out2) a.out2[b * a.out2_bs + (size_t)cc * N + p0 + pp] = v;
    }
  }
}

// The same for small clouds: a workgroup owns (cloud, slice of CS channels) and keeps the slice of the A table,
// [N][CS] floats, in LDS -- every table row is read from L2 ONCE (coalesced) instead of k times through gathers, and
// the k-fold gather runs at LDS bandwidth.  N * CS * 4 <= 64 KB (CS = 64 up to 256 points, 32 up to 512).
template <int CS>
__global__ __launch_bounds__(kEdgeThreads) void edge_max_lds_kernel(const EdgeMaxArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int PG = kEdgeThreads / CS;   // points in flight per pass
  const int N = a.N, Co = a.Co, K = a.K;
  float *tab = smem;                 // [N][CS]
  float *so = smem + (size_t)N * CS;   // [CS][33]
  int *sidx = reinterpret_cast<int *>(so + CS * 33);   // [32][K]: neighbour indices of the current 32-point tile
  const int tid = threadIdx.x;
  const size_t b = blockIdx.y;
  const int c0 = blockIdx.x * CS;
  const int cw = Co - c0 < CS ? Co - c0 : CS;          // channels of this slice that exist
  const float *ta = a.ta + b * N * Co + c0;
  for (int e = tid; e < N * (CS / 4); e += kEdgeThreads) {
    const int p = e / (CS / 4), q = (e - p * (CS / 4)) * 4;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (q + 3 < cw && (Co & 3) == 0) v = *reinterpret_cast<const f32x4 *>(ta + (size_t)p * Co + q);
    else
      for (int u = 0; u < 4; u++)
        if (q + u < cw) v[u] = ta[(size_t)p * Co + q + u];
    *reinterpret_cast<f32x4 *>(tab + (size_t)p * CS + q) = v;
  }
  __syncthreads();
  const int c = tid % CS, pg = tid / CS;
  const bool live = c < cw;
  const float sh = live ? a.shift[c0 + c] : 0.f;
  const float *tb = a.tb + b * N * Co + c0 + (live ? c : 0);
  const int *ib = a.idx + b * N * K;
  for (int t0 = 0; t0 < N; t0 += 32) {
    {
      const int np = N - t0 < 32 ? N - t0 : 32;
      for (int e = tid; e < np * K; e += kEdgeThreads) sidx[e] = ib[(size_t)t0 * K + e];
    }
    __syncthreads();
    for (int pp = pg; pp < 32; pp += PG) {
      const int p = t0 + pp;
      if (p < N) {
        const int *nb = sidx + pp * K;
        const float tbv = tb[(size_t)p * Co];
        float m = -INFINITY;
        int k = 0;
        for (; k + 4 <= K; k += 4) {
          const int j0 = nb[k], j1 = nb[k + 1], j2 = nb[k + 2], j3 = nb[k + 3];
          const float v0 = tab[j0 * CS + c], v1 = tab[j1 * CS + c], v2 = tab[j2 * CS + c], v3 = tab[j3 * CS + c];
          m = fmaxf(fmaxf(m, fmaxf(v0, v1)), fmaxf(v2, v3));
        }
        for (; k < K; k++) m = fmaxf(m, tab[nb[k] * CS + c]);
        const float s = m + tbv;
        const float y = s + sh;
        so[c * 33 + pp] = y > 0.f ? y : y * a.slope;
      }
    }
    __syncthreads();
    for (int e = tid; e < CS * 32; e += kEdgeThreads) {
      const int cc = e >> 5, pp = e & 31;
      if (cc < cw && t0 + pp < N) {
        const float v = so[cc * 33 + pp];
        a.out[b * a.out_bs + (size_t)(c0 + cc) * N + t0 + pp] = v;
        if (a.out2) a.out2[b * a.out2_bs + (size_t)(c0 + cc) * N + t0 + pp] = v;
      }
    }
    __syncthreads();
  }
}

// local_self_attention's message (reference attention.py:262-289): for point i with neighbours j in idx[i],
//   a_j = <elu(q_i)+1, elu(k_j)+1>_head,  msg_i = sum_j a_j v_j / (sum_j a_j + eps)
// (the reference's LinearAttention with ONE query token: Q.(sum_j K_j (x) v_j / K) / (Q.sum_j K_j + eps) * K).
// qkv (B,N,3C) point-major rows [q | k | v]; one wave per point, lane = channel (C <= 64, heads of dh = C / nhead
// lanes, dh a power of two); the output tile is transposed through LDS into channel-major msg (B,C,N).
__device__ __forceinline__ float edge_elu1(float x) { return x > 0.f ? x + 1.0f : __expf(x); }

template <int CTRL>
__device__ __forceinline__ float dpp_f32(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
// sum over the DH lanes of a head, every lane gets the total.  DH = 16 / 32: DPP adds inside the 16-lane row
// (lane ^ 1, lane ^ 2, other quad of the half, other half of the row -- the same pairings as the xor butterfly, so
// the result is bit-identical) + one ds_swizzle across the two rows; the generic form is five ds_bpermute round
// trips per value, which is what this kernel spent its time on.
template <int DH>
__device__ __forceinline__ float head_sum(float a, int dh) {
  if constexpr (DH == 16 || DH == 32) {
    a += dpp_f32<0xB1>(a);
    a += dpp_f32<0x4E>(a);
    a += dpp_f32<0x141>(a);
    a += dpp_f32<0x140>(a);
    if constexpr (DH == 32) a += __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(a), 0x401F));   // lane ^ 16
  } else {
    for (int m = 1; m < dh; m <<= 1) a += __shfl_xor(a, m, 64);
  }
  return a;
}

template <int DH>
__global__ __launch_bounds__(kEdgeThreads) void local_attn_kernel(const float *__restrict__ qkv,
                                                                   const int *__restrict__ idx,
                                                                   float *__restrict__ msg, int N, int C, int K, int dh,
                                                                   float eps) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float *so = smem;   // [C][33]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const size_t b = blockIdx.y;
  const int p0 = blockIdx.x * 32;
  const int np = N - p0 < 32 ? N - p0 : 32;
  const float *rows = qkv + b * N * 3 * C;
  const int cl = lane < C ? lane : C - 1;
  for (int p = wave; p < np; p += kEdgeThreads / 64) {
    const int i = p0 + p;
    const float Q = edge_elu1(rows[(size_t)i * 3 * C + cl]);
    const int *nb = idx + (b * N + i) * K;
    float num = 0.f, den = 0.f;
    int k = 0;
    for (; k + 4 <= K; k += 4) {
      float kf[4], vv[4];
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const float *r = rows + (size_t)nb[k + u] * 3 * C;
        kf[u] = r[C + cl];
        vv[u] = r[2 * C + cl];
      }
      float a[4];
#pragma unroll
      for (int u = 0; u < 4; u++) a[u] = Q * edge_elu1(kf[u]);
#pragma unroll
      for (int u = 0; u < 4; u++) a[u] = head_sum<DH>(a[u], dh);
#pragma unroll
      for (int u = 0; u < 4; u++) {
        den += a[u];
        num = fmaf(a[u], vv[u], num);
      }
    }
    for (; k < K; k++) {
      const float *r = rows + (size_t)nb[k] * 3 * C;
      float a = Q * edge_elu1(r[C + cl]);
      const float v = r[2 * C + cl];
      a = head_sum<DH>(a, dh);
      den += a;
      num = fmaf(a, v, num);
    }
    if (lane < C) so[lane * 33 + p] = num / (den + eps);
  }
  __syncthreads();
  for (int e = tid; e < C * 32; e += kEdgeThreads) {
    const int c = e >> 5, pp = e & 31;
    if (pp < np) msg[(b * C + c) * N + p0 + pp] = so[c * 33 + pp];
  }
}

// The same with one workgroup per (cloud, head) and that head's elu(k)+1 | v columns of the whole cloud resident in
// LDS ([N][DH] each): the table is read from L2 once and the K-fold neighbour gather runs from LDS (with the DPP
// reduction above the LDS pipe is free for it).  Same operation order as local_attn_kernel: bit-identical output.
template <int DH>
__global__ __launch_bounds__(kEdgeThreads) void local_attn_lds_kernel(const float *__restrict__ qkv,
                                                                       const int *__restrict__ idx,
                                                                       float *__restrict__ msg, int N, int C, int K,
                                                                       float eps) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int PG = kEdgeThreads / DH;
  float *tk = smem;                       // [N][DH] elu(k)+1
  float *tv = tk + (size_t)N * DH;        // [N][DH]
  float *so = tv + (size_t)N * DH;        // [DH][33]
  int *sidx = reinterpret_cast<int *>(so + DH * 33);   // [32][K]
  const int tid = threadIdx.x;
  const size_t b = blockIdx.y;
  const int c0 = blockIdx.x * DH;
  const float *rows = qkv + b * N * 3 * C;
  for (int e = tid; e < N * DH; e += kEdgeThreads) {
    const int p = e / DH, c = e - p * DH;
    const float *r = rows + (size_t)p * 3 * C + c0 + c;
    tk[e] = edge_elu1(r[C]);
    tv[e] = r[2 * C];
  }
  const int c = tid % DH, pg = tid / DH;
  const int *ib = idx + b * N * K;
  for (int t0 = 0; t0 < N; t0 += 32) {
    const int np = N - t0 < 32 ? N - t0 : 32;
    for (int e = tid; e < np * K; e += kEdgeThreads) sidx[e] = ib[(size_t)t0 * K + e];
    __syncthreads();   // (also orders the table fill before its first use)
    for (int pp = pg; pp < 32; pp += PG) {
      const int pc = pp < np ? pp : np - 1;            // clamped: every lane takes part in the reductions
      const float Q = edge_elu1(rows[(size_t)(t0 + pc) * 3 * C + c0 + c]);
      const int *nb = sidx + pc * K;
      float num = 0.f, den = 0.f;
      int k = 0;
      for (; k + 4 <= K; k += 4) {
        float a[4], vv[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
          const int j = nb[k + u];
          a[u] = Q * tk[j * DH + c];
          vv[u] = tv[j * DH + c];
        }
#pragma unroll
        for (int u = 0; u < 4; u++) a[u] = head_sum<DH>(a[u], DH);
#pragma unroll
        for (int u = 0; u < 4; u++) {
          den += a[u];
          num = fmaf(a[u], vv[u], num);
        }
      }
      for (; k < K; k++) {
        const int j = nb[k];
        float a = head_sum<DH>(Q * tk[j * DH + c], DH);
        den += a;
        num = fmaf(a, tv[j * DH + c], num);
      }
      if (pp < np) so[c * 33 + pp] = num / (den + eps);
    }
    __syncthreads();
    for (int e = tid; e < DH * 32; e += kEdgeThreads) {
      const int cc = e >> 5, pp = e & 31;
      if (pp < np) msg[(b * C + c0 + cc) * N + t0 + pp] = so[cc * 33 + pp];
    }
    __syncthreads();
  }
}

}  // namespace

// ------------------------------------------------------------------------------ C ABI ----
PCR_EXPORT int pcr_local_attn_f32(const float *qkv, const int *idx, float *msg, int B, int N, int C, int K, int nhead,
                                  float eps, pcr_stream_t stream) {
  if (!qkv || !idx || !msg) return PCR_ERR_INVALID;
  if (B <= 0 || N <= 0 || K <= 0 || C <= 0 || C > 64 || nhead <= 0 || C % nhead || B > 65535) return PCR_ERR_INVALID;
  const int dh = C / nhead;
  if (dh & (dh - 1)) return PCR_ERR_INVALID;
  const size_t lds32 = ((size_t)N * 32 * 2 + 32 * 33 + 32 * K) * 4;
  if (dh == 32 && lds32 <= 72 * 1024 && !pcr_tune_str("PCR_LOCAL_NO_LDS")) {
    static bool big = hipFuncSetAttribute(reinterpret_cast<const void *>(local_attn_lds_kernel<32>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess;
    (void)big;
    hipLaunchKernelGGL(local_attn_lds_kernel<32>, dim3(nhead, B), dim3(kEdgeThreads), lds32, pcr_s(stream), qkv, idx,
                       msg, N, C, K, eps);
    PCR_CHECK_LAUNCH();
    return PCR_OK;
  }
  const dim3 g((N + 31) / 32, B), blk(kEdgeThreads);
  const size_t lds = (size_t)C * 33 * 4;
  if (dh == 32) hipLaunchKernelGGL(local_attn_kernel<32>, g, blk, lds, pcr_s(stream), qkv, idx, msg, N, C, K, dh, eps);
  else if (dh == 16) hipLaunchKernelGGL(local_attn_kernel<16>, g, blk, lds, pcr_s(stream), qkv, idx, msg, N, C, K, dh, eps);
  else hipLaunchKernelGGL(local_attn_kernel<0>, g, blk, lds, pcr_s(stream), qkv, idx, msg, N, C, K, dh, eps);
  PCR_CHECK_LAUNCH();
  return PCR_OK;
}

PCR_EXPORT int pcr_knn_feat_f32(const float *x, float *xx_ws, int *idx, int B, int C, int N, int K, long x_bstride,
                                pcr_stream_t stream) {
  if (!x || !xx_ws || !idx) return PCR_ERR_INVALID;
  if (B <= 0 || C <= 0 || N <= 0 || K <= 0 || K > N || K > 64 || N > 2048 || C > 512 || B > 65535)
    return PCR_ERR_INVALID;
  if (x_bstride <= 0) x_bstride = (long)C * N;
  hipStream_t st = pcr_s(stream);
  hipLaunchKernelGGL(feat_sqnorm_kernel, dim3((N + 255) / 256, B), dim3(256), 0, st, x, xx_ws, C, N, x_bstride);
  const int T = N <= 64 ? 1 : N <= 128 ? 2 : N <= 256 ? 4 : N <= 512 ? 8 : N <= 1024 ? 16 : 32;
  const int QT = N <= 1024 ? 32 : 16;
  const int C2 = (C + 1) / 2;
  const size_t lds = (size_t)(2 * C2 * 32 + 32) * 4 + (size_t)QT * 64 * T * 4 + (size_t)(kKnnThreads / 64) * kSelCap * 8;
  if (lds > 160 * 1024) return PCR_ERR_INVALID;
  const dim3 g((N + QT - 1) / QT, B), blk(kKnnThreads);
#define PCR_KNNF(TT)                                                                                              \
  do {                                                                                                            \
    static bool big = hipFuncSetAttribute(reinterpret_cast<const void *>(knn_feat_kernel<TT>),                    \
                                          hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess;  \
    (void)big;                                                                                                    \
    hipLaunchKernelGGL((knn_feat_kernel<TT>), g, blk, lds, st, x, (const float *)xx_ws, idx, C, N, K, QT,         \
                       x_bstride);                                                                                \
  } while (0)
  switch (T) {
    case 1: PCR_KNNF(1); break;
    case 2: PCR_KNNF(2); break;
    case 4: PCR_KNNF(4); break;
    case 8: PCR_KNNF(8); break;
    case 16: PCR_KNNF(16); break;
    default: PCR_KNNF(32); break;
  }
#undef PCR_KNNF
  PCR_CHECK_LAUNCH();
  return PCR_OK;
}

PCR_EXPORT int pcr_edge_max_f32(const float *ta, const float *tb, const int *idx, const float *shift, float slope,
                                float *out, long out_bstride, float *out2, long out2_bstride, int B, int N, int Co,
                                int K, pcr_stream_t stream) {
  if (!ta || !tb || !idx || !shift || !out) return PCR_ERR_INVALID;
  if (B <= 0 || N <= 0 || Co <= 0 || Co > 256 || K <= 0 || K > 64 || B > 65535) return PCR_ERR_INVALID;
  EdgeMaxArgs a{ta, tb, shift, idx, out, out2, out_bstride > 0 ? out_bstride : (long)Co * N,
                out2_bstride > 0 ? out2_bstride : (long)Co * N, N, Co, K, slope};
  if (N <= 512 && !pcr_tune_str("PCR_EDGE_NO_LDS")) {
    // channel-slice width: narrower slices = more, smaller workgroups.  Measured (512 pairs/step, ms for the four
    // layers, CS = 64 / 32 / 16): N = 128: 0.46 / 0.35 / 0.36; N = 256: 1.11 / 0.83 / 0.76; the L2-gather kernel
    // below: 0.72 and 1.55, but 1.52 against 1.69 at N = 1024, where it stays.
    static const int cs_env = pcr_tune_int("PCR_EDGE_CS");
    const int cs = cs_env ? cs_env : (N <= 128 ? 32 : 16);
#define PCR_EDGE_LDS(CS_)                                                                                         \
  do {                                                                                                            \
    static bool big = hipFuncSetAttribute(reinterpret_cast<const void *>(edge_max_lds_kernel<CS_>),               \
                                          hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess;  \
    (void)big;                                                                                                    \
    hipLaunchKernelGGL(edge_max_lds_kernel<CS_>, dim3((Co + CS_ - 1) / CS_, B), dim3(kEdgeThreads),               \
                       (size_t)(N * CS_ + CS_ * 33 + 32 * K) * 4, pcr_s(stream), a);                              \
  } while (0)
    if (cs == 64 && N <= 512) PCR_EDGE_LDS(64);
    else if (cs == 16 || N > 512) PCR_EDGE_LDS(16);
    else PCR_EDGE_LDS(32);
#undef PCR_EDGE_LDS
    PCR_CHECK_LAUNCH();
    return PCR_OK;
  }
  const size_t lds = (size_t)(32 * K + Co * 33) * 4;
  hipLaunchKernelGGL(edge_max_kernel, dim3((N + 31) / 32, B), dim3(kEdgeThreads), lds, pcr_s(stream), a);
  PCR_CHECK_LAUNCH();
  return PCR_OK;
}
