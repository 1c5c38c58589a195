// Training-mode kernels of the siamese ReID hot path (forward with BatchNorm batch statistics + backward).
//
// The reference trains this path with unfused autograd ops over materialised (B,C,S,K) tensors
// (models/ReIDNet.py:586-634,694-738; models/pointnet2_utils.py:333-360 with BatchNorm2d in batch-statistics mode).
// Here every dense layer is ONE launch forward and ONE launch backward, on the same (B, C, L) channel-major tensors
// the inference kernels use (L = S*K rows for the grouped MLPs):
//   forward  y = W f(x) + b            f = the previous layer's BatchNorm affine + ReLU, applied WHILE the tile is
//                                      loaded (the normalised activation never exists in HBM); per-channel sum and
//                                      sum of squares of y leave the kernel as per-workgroup partials;
//   backward dy = BN-backward(g, y)    formed while the tiles are loaded (BatchNorm's backward is linear in the
//                                      incoming gradient and the stored pre-activation: dy = ka g + kb y + kc);
//            dW += dy f(x)^T           on the matrix core (contraction over the tokens of the tile), accumulated
//                                      in registers over the workgroup's tiles;
//            dx  = (W^T dy) [f(x) > 0] on the matrix core, with the sums the NEXT BatchNorm backward needs.
// What a layer keeps for its backward is its raw output y (needed anyway: BatchNorm statistics are global, so the
// layer below cannot be normalised before the whole layer above it exists).  All reductions are two-stage with a
// fixed order (per-workgroup partials, then a reduce kernel): no float atomics, bit-reproducible gradients.
#include <type_traits>

#include "tile_dense.h"
#include "train_stream.h"

namespace {

constexpr int kTT = 64, kTRP = 65;   // tokens per tile (TB = 2), LDS row pitch

// dst[c][t] = f(c, t0 + t) for c < C, zero for rows [C, CP) and tokens beyond L.  F4(c, tg) -> four tokens
// tg .. tg+3 (an aligned piece inside the row; rows of L % 4 == 0 floats: `vec`), F1(c, tg) -> one.  Loads are issued in
// batches from clamped addresses.  A ragged last tile of a `vec` tensor is still filled by 16-byte pieces (the pieces
// beyond L are zeros): short clouds (L = 32: every tile is ragged) took 128 dependent scalar batches per thread on the
// 256-channel layers before, 0.5 ms of latency on a 0.07 ms launch.
template <int U = 4, class F4, class F1>
__device__ __forceinline__ void tile_fill(float *dst, int C, int CP, int L, int t0, bool vec, F4 f4, F1 f1) {
  constexpr int T = kTT, RP = kTRP, Q = T / 4;
  if (vec) {
    const int totq = CP * Q;
    const int nq = L - t0 >= T ? Q : (L - t0) >> 2;      // valid pieces of a row
    for (int e0 = threadIdx.x; e0 < totq; e0 += U * kThreads) {
      f32x4 v[U];
#pragma unroll
      for (int u = 0; u < U; u++) {
        const int e = e0 + u * kThreads;
        const int c = e / Q, q = e - c * Q;
        const bool ok = e < totq && c < C && q < nq;
        const f32x4 x = f4(ok ? c : 0, t0 + 4 * (ok ? q : 0));
        v[u] = ok ? x : f32x4{0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int u = 0; u < U; u++) {
        const int e = e0 + u * kThreads;
        if (e < totq) {
          const int c = e / Q, q = e - c * Q;
          float *d = dst + c * RP + 4 * q;
          d[0] = v[u][0];
          d[1] = v[u][1];
          d[2] = v[u][2];
          d[3] = v[u][3];
        }
      }
    }
  } else {
    const int total = CP * T;
    for (int e0 = threadIdx.x; e0 < total; e0 += 4 * kThreads) {
      float v[4];
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const int e = e0 + u * kThreads;
        const int c = e / T, t = e - c * T;
        const bool ok = e < total && c < C && t0 + t < L;
        const float x = f1(ok ? c : 0, ok ? t0 + t : 0);
        v[u] = ok ? x : 0.f;
      }
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const int e = e0 + u * kThreads;
        if (e < total) {
          const int c = e / T, t = e - c * T;
          dst[c * RP + t] = v[u];
        }
      }
    }
  }
}

__device__ __forceinline__ f32x4 ld4(const float *p) { return *reinterpret_cast<const f32x4 *>(p); }

// Row sums of an LDS tile use ALL 256 threads: thread t owns the 16-token quarter (t & 3) of rows (t >> 2) + 64 p and keeps
// its partial sums across the workgroup's tiles; the four quarters of a row meet once, at the end (two DPP adds inside
// the quad).  (One thread per row -- 64 dependent-ish iterations on a quarter of the threads -- was the longest serial
// stretch of a tile.)
constexpr int kRowPasses = 6;   // up to 384 rows
template <int CTRL>
__device__ __forceinline__ float tk_dpp(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float quad_sum(float v) {
  v += tk_dpp<0xB1>(v);
  v += tk_dpp<0x4E>(v);
  return v;
}

// ---------------------------------------------------------------------------------- forward ----
struct TFwd {
  const float *x, *x2;              // (B,cin1,L), (B,cin2,L) or null: the layer's input is [x ; x2] along the channels
  int cin1, cin2;
  const float *isc, *ish;           // (cin1) input affine of x (previous BatchNorm), null = none
  int in_relu;
  const float *wp, *bias;           // packed (cout, cin1+cin2); bias zero-padded to ceil32(cout), or null
  const float *res;                 // (B,cout,L) added to the output, or null
  int out_relu;
  float *y;
  int cout, L, tpw;                 // tiles per workgroup and cloud
  int B;                            // clouds: workgroup (x, y) takes clouds y, y + gridDim.y, ...
  float *stats;                     // partials [workgroups][2][ceil32(cout)] (sum, sum of squares of y), or null
};

// PFQ > 0: the NEXT tile's raw input (PFQ 16-byte pieces per thread) is requested right after the current tile has been
// committed to LDS, so its HBM round trip hides behind the matrix phase and the epilogue of the current tile
// WS / NR: the dense shape (tile_dense2): WS = 4 / 2 / 1 for one / two / >= 3 cout blocks, NR rounds of four blocks per wave
template <int WS, int NR, int PFQ>
__global__ __launch_bounds__(kThreads) void tdense_fwd_kernel(TFwd a) {
  constexpr int TB = 2, T = kTT, RP = kTRP;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int cin = a.cin1 + a.cin2, cinP = ceil8(cin), coutP = ceil32(a.cout), L = a.L;
  const int rows = cinP > coutP ? cinP : coutP;
  float *X = smem;
  float *s_isc = X + rows * RP, *s_ish = s_isc + a.cin1;
  const int tid = threadIdx.x;
  size_t b = blockIdx.y;
  const bool aff = a.isc != nullptr;
  if (aff)
    for (int e = tid; e < a.cin1; e += kThreads) {
      s_isc[e] = a.isc[e];
      s_ish[e] = a.ish[e];
    }
  // per-cloud bases (the fill / store lambdas below read the current ones)
  const float *xb = a.x + b * a.cin1 * L;
  const float *x2b = a.x2 ? a.x2 + b * a.cin2 * L : xb;
  const float *resb = a.res ? a.res + b * a.cout * L : nullptr;
  float *yb = a.y + b * a.cout * L;
  const bool vec = (L & 3) == 0;
  const int cin1 = a.cin1, in_relu = a.in_relu;
  float ssum[kRowPasses], ssq[kRowPasses];
#pragma unroll
  for (int p = 0; p < kRowPasses; p++) ssum[p] = ssq[p] = 0.f;
  __syncthreads();
  auto f4 = [&](int c, int tg) {
    const bool first = c < cin1;
    f32x4 v = ld4((first ? xb + (size_t)c * L : x2b + (size_t)(c - cin1) * L) + tg);
    if (aff && first) {
      const float sc = s_isc[c], sh = s_ish[c];
#pragma unroll
      for (int j = 0; j < 4; j++) {
        v[j] = fmaf(v[j], sc, sh);
        if (in_relu) v[j] = fmaxf(v[j], 0.f);
      }
    }
    return v;
  };
  auto f1 = [&](int c, int tg) {
    const bool first = c < cin1;
    float v = (first ? xb + (size_t)c * L : x2b + (size_t)(c - cin1) * L)[tg];
    if (aff && first) {
      v = fmaf(v, s_isc[c], s_ish[c]);
      if (in_relu) v = fmaxf(v, 0.f);
    }
    return v;
  };
  constexpr int NPQ = PFQ > 0 ? PFQ : 1;
  f32x4 pre[NPQ];
  const int totq = cinP * (T / 4);
  auto fetch = [&](int t0) {
#pragma unroll
    for (int u = 0; u < NPQ; u++) {
      const int e = tid + u * kThreads;
      const int c = e >> 4, q = e & 15;
      const bool ok = e < totq && c < cin;
      const int cc = ok ? c : 0;
      pre[u] = ld4((cc < cin1 ? xb + (size_t)cc * L : x2b + (size_t)(cc - cin1) * L) + t0 + 4 * (ok ? q : 0));
    }
  };
  auto commit = [&]() {
#pragma unroll
    for (int u = 0; u < NPQ; u++) {
      const int e = tid + u * kThreads;
      if (e < totq) {
        const int c = e >> 4, q = e & 15;
        f32x4 v = pre[u];
        if (c >= cin) v = f32x4{0.f, 0.f, 0.f, 0.f};
        else if (aff && c < cin1) {
          const float sc = s_isc[c], sh = s_ish[c];
#pragma unroll
          for (int j = 0; j < 4; j++) {
            v[j] = fmaf(v[j], sc, sh);
            if (in_relu) v[j] = fmaxf(v[j], 0.f);
          }
        }
        float *d = X + c * RP + 4 * q;
        d[0] = v[0];
        d[1] = v[1];
        d[2] = v[2];
        d[3] = v[3];
      }
    }
  };
  auto whole = [&](int t0) { return PFQ > 0 && vec && t0 + T <= L; };
  const int tfirst = blockIdx.x * a.tpw * T;
  for (bool first_cloud = true; b < (size_t)a.B; b += gridDim.y, first_cloud = false) {
  if (!first_cloud) {
    xb = a.x + b * a.cin1 * L;
    x2b = a.x2 ? a.x2 + b * a.cin2 * L : xb;
    resb = a.res ? a.res + b * a.cout * L : nullptr;
    yb = a.y + b * a.cout * L;
    __syncthreads();     // the previous cloud's last tile has been stored from X
  }
  bool have = false;
  if (tfirst < L && whole(tfirst)) {
    fetch(tfirst);
    have = true;
  }
  for (int ti = 0; ti < a.tpw; ti++) {
    const int t0 = (blockIdx.x * a.tpw + ti) * T;
    if (t0 >= L) break;
    if (ti) __syncthreads();
    if (have) commit();
    else tile_fill(X, cin, cinP, L, t0, vec, f4, f1);
    __syncthreads();
    have = ti + 1 < a.tpw && t0 + T < L && whole(t0 + T);
    if (have) fetch(t0 + T);
    tile_dense2<TB, NR, WS>(X, cinP, a.wp, coutP, true, [&](float v, int o, int t) { X[o * RP + t] = v; }, a.bias);
    __syncthreads();
    const int valid = L - t0 < T ? L - t0 : T;
    if (a.stats) {   // BatchNorm statistics of the raw output (bias included): quarter-row partial sums
      const int q16 = 16 * (tid & 3);
#pragma unroll
      for (int p = 0; p < kRowPasses; p++) {
        const int r = (tid >> 2) + 64 * p;
        if (r < coutP) {
          const float *row = X + r * RP + q16;
          float p0 = 0.f, p1 = 0.f, q0 = 0.f, q1 = 0.f;
#pragma unroll
          for (int t = 0; t < 16; t += 2) {
            const float v0 = q16 + t < valid ? row[t] : 0.f, v1 = q16 + t + 1 < valid ? row[t + 1] : 0.f;
            p0 += v0;
            p1 += v1;
            q0 += v0 * v0;
            q1 += v1 * v1;
          }
          ssum[p] += p0 + p1;
          ssq[p] += q0 + q1;
        }
      }
    }
    if (vec) {
      constexpr int Q = T / 4;
      const int nq = valid >> 2;
      for (int e = tid; e < a.cout * Q; e += kThreads) {
        const int o = e / Q, q = e - o * Q;
        if (q >= nq) continue;
        const float *xs = X + o * RP + 4 * q;
        f32x4 v = {xs[0], xs[1], xs[2], xs[3]};
        if (resb) {
          const f32x4 r = ld4(resb + (size_t)o * L + t0 + 4 * q);
          v += r;
        }
        if (a.out_relu) {
#pragma unroll
          for (int j = 0; j < 4; j++) v[j] = fmaxf(v[j], 0.f);
        }
        *reinterpret_cast<f32x4 *>(yb + (size_t)o * L + t0 + 4 * q) = v;
      }
    } else {
      for (int e = tid; e < a.cout * T; e += kThreads) {
        const int o = e / T, t = e - o * T;
        if (t < valid) {
          float v = X[o * RP + t];
          if (resb) v += resb[(size_t)o * L + t0 + t];
          if (a.out_relu) v = fmaxf(v, 0.f);
          yb[(size_t)o * L + t0 + t] = v;
        }
      }
    }
  }
  }   // clouds
  if (a.stats) {
    float *sp = a.stats + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 2 * coutP;
#pragma unroll
    for (int p = 0; p < kRowPasses; p++) {
      const int r = (tid >> 2) + 64 * p;
      const float s1 = quad_sum(ssum[p]), s2 = quad_sum(ssq[p]);
      if ((tid & 3) == 0 && r < coutP) {
        sp[r] = s1;
        sp[coutP + r] = s2;
      }
    }
  }
}

// --------------------------------------------------------------------------------- backward ----
// dy_mode: 0 dy = g | 1 dy = ka g + kb y + kc (BatchNorm backward) | 2 dy = g [y > 0] (ReLU output layer)
//          3 g is the gradient of the max-pooled output: g_eff[c][s K + k] = gp[c][s] if k == argmax[c][s] and
//            pooled[c][s] > 0, else 0; then as mode 1
struct TBwd {
  const float *g, *y;
  int dy_mode;
  const float *ka, *kb, *kc;
  const int *argmax;
  const float *pooled;
  int K, S;
  const float *x, *x2;              // the layer's forward input, as in TFwd
  int cin1, cin2;
  const float *isc, *ish, *iinv;    // input affine (+ 1 / isc: recovers the raw input for the next BN backward)
  int in_relu;
  const float *wpT;                 // packed (cin, cout) = W^T, or null: no input gradient wanted
  const float *wpT_bf;              // (BF kernels) the same matrix as a bf16 hi / lo image
  float *dx, *dx2;
  float *dstats;                    // partials [workgroups][2][ceil32(cin1)]: sum dxm, sum dxm * raw x (dxm = masked dx)
  float *dwp, *dbp;                 // partials [workgroups][ceil32(cout)][ceil32(cin)], [workgroups][ceil32(cout)]
  long dw_stride, db_stride;        // floats between the workgroups' partials (0 = dense arrays)
  int cout, L, tpw;
  int B;                            // clouds: workgroup (x, y, z) takes clouds y, y + gridDim.y, ...
  int dbg;                          // PCR_TD_DBG ablation mask (diagnostics only; 0 in production)
};

// QY / QX > 0: register prefetch of the NEXT tile (QY pieces of g and of y, QX pieces of the forward input per thread),
// requested once the current tile sits in LDS: the HBM round trip hides behind the two matrix phases
// BF (128 x 128 layers, round 4): both matrix phases on the bf16 matrix core as split bf16 (three MFMAs per product, f32
// accumulation) -- dx through tile_dense2p on the f32 tile (operands converted where they are consumed), dW by
// contracting the tile's 64 tokens in four steps of 16 (A = dy rows of a cout block, B = f(x) rows of the wave's cin
// block; token k of a step = 16 s + 8 h + j for both operands).
// (BF kernel) depth of the dx phase's weight ring in 16-channel steps.  tile_dense2p's default of four put the kernel 119
// registers into scratch -- in the dx k-loop, i.e. in every tile of the step's most expensive launch -- and nobody had
// looked (`tools/kres.py` lists registers / spills per kernel); two steps: 16 spilled, 0.708 -> 0.604 ms per launch.
#ifndef PCR_TDBF_DW_UNROLL
#define PCR_TDBF_DW_UNROLL 4
#endif
#ifndef PCR_TDBF_PF
#define PCR_TDBF_PF 2
#endif
template <int WSX, int NRX, int NTW, int QY, int QX, bool BF = false>
__device__ __forceinline__ void tdense_bwd_body(const TBwd &a) {
  constexpr int TB = 2, T = kTT, RP = kTRP;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int cin = a.cin1 + a.cin2, cinP = ceil32(cin), coutP = ceil32(a.cout), L = a.L;
  const int rowsY = coutP > cinP ? coutP : cinP;
  float *DY = smem;                 // [rowsY][RP]: dy, then (in place) W^T dy
  float *AT = DY + rowsY * RP;      // [cinP][RP]: the forward input f(x)
  float *s_k = AT + cinP * RP;      // ka | kb | kc (cout each), isc | ish | iinv (cin1 each)
  float *s_ka = s_k, *s_kb = s_k + a.cout, *s_kc = s_k + 2 * a.cout;
  float *s_isc = s_k + 3 * a.cout, *s_ish = s_isc + a.cin1, *s_iinv = s_ish + a.cin1;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, h = lane >> 5;
  size_t b = blockIdx.y;
  const int z = blockIdx.z;
  const bool bn = a.dy_mode == 1 || a.dy_mode == 3;
  const bool aff = a.isc != nullptr;
  for (int e = tid; e < a.cout; e += kThreads) {
    s_ka[e] = bn ? a.ka[e] : 1.f;
    s_kb[e] = bn ? a.kb[e] : 0.f;
    s_kc[e] = bn ? a.kc[e] : 0.f;
  }
  for (int e = tid; e < a.cin1; e += kThreads) {
    s_isc[e] = aff ? a.isc[e] : 1.f;
    s_ish[e] = aff ? a.ish[e] : 0.f;
    s_iinv[e] = (aff && a.iinv) ? a.iinv[e] : 1.f;
  }
  const int mode = a.dy_mode, K = a.K, S = a.S, cout = a.cout;
  const float *gb = mode == 3 ? a.g + b * a.cout * S : a.g + b * a.cout * L;
  const float *yb = a.y ? a.y + b * a.cout * L : gb;
  const int *amb = mode == 3 ? a.argmax + b * a.cout * S : nullptr;
  const bool has_pl = a.pooled != nullptr;      // (null: g is already zero where the pooled activation is)
  const float *plb = (mode == 3 && has_pl) ? a.pooled + b * a.cout * S : nullptr;
  const float *xb = a.x + b * a.cin1 * L;
  const float *x2b = a.x2 ? a.x2 + b * a.cin2 * L : xb;
  const bool vec = (L & 3) == 0;
  const int cin1 = a.cin1, in_relu = a.in_relu;
  // the gradient tile: one fetch per element whatever the mode (mode 3 reads the small pooled tensors instead of g)
  auto g1 = [&](int c, int tg) {
    if (mode != 3) return gb[(size_t)c * L + tg];
    const int s = tg / K, k = tg - s * K;
    const size_t o = (size_t)c * S + s;
    return (amb[o] == k && (!has_pl || plb[o] > 0.f)) ? gb[o] : 0.f;
  };
  auto dy4 = [&](int c, int tg) {
    f32x4 g;
    if (mode != 3) g = ld4(gb + (size_t)c * L + tg);
    else if ((K & 3) == 0) {   // the four tokens of an aligned piece belong to one centre
      const int s = tg / K, k = tg - s * K;
      const size_t o = (size_t)c * S + s;
      const int am = amb[o] - k;
      const float gv = (!has_pl || plb[o] > 0.f) ? gb[o] : 0.f;
#pragma unroll
      for (int j = 0; j < 4; j++) g[j] = am == j ? gv : 0.f;
    } else {
#pragma unroll
      for (int j = 0; j < 4; j++) g[j] = g1(c, tg + j);
    }
    if (mode == 0) return g;
    const f32x4 yv = ld4(yb + (size_t)c * L + tg);
    f32x4 r;
    if (mode == 2) {
#pragma unroll
      for (int j = 0; j < 4; j++) r[j] = yv[j] > 0.f ? g[j] : 0.f;
    } else {
      const float ka = s_ka[c], kb = s_kb[c], kc = s_kc[c];
#pragma unroll
      for (int j = 0; j < 4; j++) r[j] = fmaf(ka, g[j], fmaf(kb, yv[j], kc));
    }
    return r;
  };
  auto dy1 = [&](int c, int tg) {
    const float g = g1(c, tg);
    if (mode == 0) return g;
    const float yv = yb[(size_t)c * L + tg];
    if (mode == 2) return yv > 0.f ? g : 0.f;
    return fmaf(s_ka[c], g, fmaf(s_kb[c], yv, s_kc[c]));
  };
  auto a4 = [&](int c, int tg) {
    const bool first = c < cin1;
    f32x4 v = ld4((first ? xb + (size_t)c * L : x2b + (size_t)(c - cin1) * L) + tg);
    if (aff && first) {
      const float sc = s_isc[c], sh = s_ish[c];
#pragma unroll
      for (int j = 0; j < 4; j++) {
        v[j] = fmaf(v[j], sc, sh);
        if (in_relu) v[j] = fmaxf(v[j], 0.f);
      }
    }
    return v;
  };
  auto a1 = [&](int c, int tg) {
    const bool first = c < cin1;
    float v = (first ? xb + (size_t)c * L : x2b + (size_t)(c - cin1) * L)[tg];
    if (aff && first) {
      v = fmaf(v, s_isc[c], s_ish[c]);
      if (in_relu) v = fmaxf(v, 0.f);
    }
    return v;
  };
  // dW tiles of this workgroup: items [z 4 NTW, (z + 1) 4 NTW) of the (cout block, cin block) grid
  const int nIB = cinP >> 5, items = (coutP >> 5) * nIB;
  f32x16 acc[NTW];
#pragma unroll
  for (int i = 0; i < NTW; i++)
#pragma unroll
    for (int r = 0; r < 16; r++) acc[i][r] = 0.f;
  float dbsum[kRowPasses], s1[kRowPasses], s2[kRowPasses];
#pragma unroll
  for (int p = 0; p < kRowPasses; p++) dbsum[p] = s1[p] = s2[p] = 0.f;
  const bool want_dx = a.wpT != nullptr && z == 0;
  __syncthreads();
  constexpr bool kPF = QY > 0;
  constexpr int NQY = kPF ? QY : 1, NQX = kPF ? QX : 1;
  f32x4 pg[NQY], py[NQY], px[NQX];
  auto fetch = [&](int t0) {
#pragma unroll
    for (int u = 0; u < NQY; u++) {
      const int e = tid + u * kThreads;
      const int c = e >> 4, q = e & 15;
      const bool ok = c < cout;                 // (rows [cout, rowsY) stay zero)
      const int cc = ok ? c : 0, tg = t0 + 4 * q;
      if (mode == 3) {
        const int sc = tg / K, k = tg - sc * K;
        const size_t o = (size_t)cc * S + sc;
        pg[u] = f32x4{__int_as_float(amb[o] - k), has_pl ? plb[o] : 1.f, gb[o], 0.f};
      } else {
        pg[u] = ld4(gb + (size_t)cc * L + tg);
      }
      if (mode != 0) py[u] = ld4(yb + (size_t)cc * L + tg);
    }
#pragma unroll
    for (int u = 0; u < NQX; u++) {
      const int e = tid + u * kThreads;
      const int c = e >> 4, q = e & 15;
      const bool ok = c < cin;
      const int cc = ok ? c : 0;
      px[u] = ld4((cc < cin1 ? xb + (size_t)cc * L : x2b + (size_t)(cc - cin1) * L) + t0 + 4 * q);
    }
  };
  auto commit = [&]() {
#pragma unroll
    for (int u = 0; u < NQY; u++) {
      const int e = tid + u * kThreads;
      const int c = e >> 4, q = e & 15;
      f32x4 r = {0.f, 0.f, 0.f, 0.f};
      if (c < cout) {
        f32x4 g = pg[u];
        if (mode == 3) {
          const int am = __float_as_int(pg[u][0]);
          const float gv = pg[u][1] > 0.f ? pg[u][2] : 0.f;
#pragma unroll
          for (int j = 0; j < 4; j++) g[j] = am == j ? gv : 0.f;
        }
        if (mode == 0) r = g;
        else if (mode == 2) {
#pragma unroll
          for (int j = 0; j < 4; j++) r[j] = py[u][j] > 0.f ? g[j] : 0.f;
        } else {
          const float ka = s_ka[c], kb = s_kb[c], kc = s_kc[c];
#pragma unroll
          for (int j = 0; j < 4; j++) r[j] = fmaf(ka, g[j], fmaf(kb, py[u][j], kc));
        }
      }
      float *d = DY + c * RP + 4 * q;
      d[0] = r[0];
      d[1] = r[1];
      d[2] = r[2];
      d[3] = r[3];
    }
#pragma unroll
    for (int u = 0; u < NQX; u++) {
      const int e = tid + u * kThreads;
      const int c = e >> 4, q = e & 15;
      f32x4 v = px[u];
      if (c >= cin) v = f32x4{0.f, 0.f, 0.f, 0.f};
      else if (aff && c < cin1) {
        const float sc = s_isc[c], sh = s_ish[c];
#pragma unroll
        for (int j = 0; j < 4; j++) {
          v[j] = fmaf(v[j], sc, sh);
          if (in_relu) v[j] = fmaxf(v[j], 0.f);
        }
      }
      float *d = AT + c * RP + 4 * q;
      d[0] = v[0];
      d[1] = v[1];
      d[2] = v[2];
      d[3] = v[3];
    }
  };
  // The fill of the variants without prefetch: 16-byte pieces in batches of four per thread, RAW loads first (clamped
  // addresses, no data-dependent branch between them: all of a batch is in flight at once), transformation and LDS
  // writes after.  (The lambda-per-element form compiled to one `s_waitcnt vmcnt(0)` per piece inside divergent
  // branches: 16 serial HBM round trips per 128-channel tile.)
  auto fill_dy = [&](auto mtag, int t0, int nq) {
    constexpr int M = decltype(mtag)::value;
    const int totq = rowsY * 16;
    for (int e0 = tid; e0 < totq; e0 += 4 * kThreads) {
      f32x4 gq[4], yq[4];
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const int e = e0 + u * kThreads, c = e >> 4, q = e & 15;
        const bool ok = e < totq && c < cout && q < nq;
        const int cc = ok ? c : 0, tg = t0 + 4 * (ok ? q : 0);
        if constexpr (M == 3) {
          const int sc = tg / K, k = tg - sc * K;
          const size_t o = (size_t)cc * S + sc;
          gq[u] = f32x4{__int_as_float(amb[o] - k), has_pl ? plb[o] : 1.f, gb[o], 0.f};
        } else {
          gq[u] = ld4(gb + (size_t)cc * L + tg);
        }
        if constexpr (M != 0) yq[u] = ld4(yb + (size_t)cc * L + tg);
      }
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const int e = e0 + u * kThreads, c = e >> 4, q = e & 15;
        const bool ok = e < totq && c < cout && q < nq;
        const int cc = ok ? c : 0;
        f32x4 g = gq[u], r;
        if constexpr (M == 3) {
          const int am = __float_as_int(gq[u][0]);
          const float gv = gq[u][1] > 0.f ? gq[u][2] : 0.f;
#pragma unroll
          for (int j = 0; j < 4; j++) g[j] = am == j ? gv : 0.f;
        }
        if constexpr (M == 0) r = g;
        else if constexpr (M == 2) {
#pragma unroll
          for (int j = 0; j < 4; j++) r[j] = yq[u][j] > 0.f ? g[j] : 0.f;
        } else {
          const float ka = s_ka[cc], kb = s_kb[cc], kc = s_kc[cc];
#pragma unroll
          for (int j = 0; j < 4; j++) r[j] = fmaf(ka, g[j], fmaf(kb, yq[u][j], kc));
        }
#pragma unroll
        for (int j = 0; j < 4; j++) r[j] = ok ? r[j] : 0.f;
        if (e < totq) {
          float *d = DY + c * RP + 4 * q;
          d[0] = r[0];
          d[1] = r[1];
          d[2] = r[2];
          d[3] = r[3];
        }
      }
    }
  };
  auto fill_x = [&](int t0, int nq) {
    const int totq = cinP * 16;
    for (int e0 = tid; e0 < totq; e0 += 4 * kThreads) {
      f32x4 xq[4];
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const int e = e0 + u * kThreads, c = e >> 4, q = e & 15;
        const bool ok = e < totq && c < cin && q < nq;
        const int cc = ok ? c : 0, tg = t0 + 4 * (ok ? q : 0);
        const float *src = cc < cin1 ? xb + (size_t)cc * L : x2b + (size_t)(cc - cin1) * L;
        xq[u] = ld4(src + tg);
      }
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const int e = e0 + u * kThreads, c = e >> 4, q = e & 15;
        const bool ok = e < totq && c < cin && q < nq;
        const int cc = ok ? c : 0;
        const bool first = cc < cin1;
        const int ci = first ? cc : 0;
        const float sc = first ? s_isc[ci] : 1.f, sh = first ? s_ish[ci] : 0.f;
        const bool rl = in_relu && aff && first;
        f32x4 v = xq[u];
#pragma unroll
        for (int j = 0; j < 4; j++) {
          float t = (aff && first) ? fmaf(v[j], sc, sh) : v[j];
          t = rl ? fmaxf(t, 0.f) : t;
          v[j] = ok ? t : 0.f;
        }
        if (e < totq) {
          float *d = AT + c * RP + 4 * q;
          d[0] = v[0];
          d[1] = v[1];
          d[2] = v[2];
          d[3] = v[3];
        }
      }
    }
  };
  auto fill_vec = [&](int t0) {
    const int nq = L - t0 >= T ? 16 : (L - t0) >> 2;
    // (x first: its loads are independent of the mode switch and travel while the dy rows are formed)
    fill_x(t0, nq);
    switch (mode) {
      case 0: fill_dy(std::integral_constant<int, 0>(), t0, nq); break;
      case 1: fill_dy(std::integral_constant<int, 1>(), t0, nq); break;
      case 2: fill_dy(std::integral_constant<int, 2>(), t0, nq); break;
      default: fill_dy(std::integral_constant<int, 3>(), t0, nq); break;
    }
  };
  // (the prefetch path covers whole tiles of layers whose tiles are exactly QY / QX pieces per thread)
  const bool pf_ok = kPF && vec && (mode != 3 || (K & 3) == 0) && rowsY == 16 * QY && cinP == 16 * QX;
  auto whole = [&](int t0) { return pf_ok && t0 + T <= L; };
  const int tfirst = blockIdx.x * a.tpw * T;
  for (bool first_cloud = true; b < (size_t)a.B; b += gridDim.y, first_cloud = false) {
  if (!first_cloud) {
    gb = mode == 3 ? a.g + b * a.cout * S : a.g + b * a.cout * L;
    yb = a.y ? a.y + b * a.cout * L : gb;
    amb = mode == 3 ? a.argmax + b * a.cout * S : nullptr;
    plb = (mode == 3 && has_pl) ? a.pooled + b * a.cout * S : nullptr;
    xb = a.x + b * a.cin1 * L;
    x2b = a.x2 ? a.x2 + b * a.cin2 * L : xb;
    __syncthreads();     // the previous cloud's last tile has been consumed
  }
  bool have = false;
  if (tfirst < L && whole(tfirst)) {
    fetch(tfirst);
    have = true;
  }
  for (int ti = 0; ti < a.tpw; ti++) {
    const int t0 = (blockIdx.x * a.tpw + ti) * T;
    if (t0 >= L) break;
    if (ti) __syncthreads();
    if (a.dbg & 16) {
    } else if (have) commit();
    else if (vec && (mode != 3 || (K & 3) == 0)) fill_vec(t0);
    else
    // (rows that are not a multiple of four floats) ONE fill over both tiles (AT follows DY in LDS)
    tile_fill(DY, rowsY + cinP, rowsY + cinP, L, t0, vec,
              [&](int c, int tg) {
                if (c < rowsY) return c < cout ? dy4(c, tg) : f32x4{0.f, 0.f, 0.f, 0.f};
                return c - rowsY < cin ? a4(c - rowsY, tg) : f32x4{0.f, 0.f, 0.f, 0.f};
              },
              [&](int c, int tg) {
                if (c < rowsY) return c < cout ? dy1(c, tg) : 0.f;
                return c - rowsY < cin ? a1(c - rowsY, tg) : 0.f;
              });
    __syncthreads();
    have = ti + 1 < a.tpw && t0 + T < L && whole(t0 + T);
    if (have) fetch(t0 + T);
    const int valid = L - t0 < T ? L - t0 : T;
    // (tile_fill leaves the padding tokens of a ragged tile zero, also where BatchNorm's backward adds a constant)
    if (a.dbp && z == 0 && !(a.dbg & 4)) {   // sum of dy over the tile's tokens (padding tokens are zero), quarter rows
      const int q16 = 16 * (tid & 3);
#pragma unroll
      for (int p = 0; p < kRowPasses; p++) {
        const int r = (tid >> 2) + 64 * p;
        if (r < coutP) {
          const float *row = DY + r * RP + q16;
          float p0 = 0.f, p1 = 0.f;
#pragma unroll
          for (int t = 0; t < 16; t += 2) {
            p0 += row[t];
            p1 += row[t + 1];
          }
          dbsum[p] += p0 + p1;
        }
      }
    }
    if constexpr (BF) {
      // (128 x 128: sixteen dW tiles, item = wave + 4 it -> cout block it, cin block wave)
      if (a.dwp && !(a.dbg & 1)) {
#pragma unroll PCR_TDBF_DW_UNROLL
        for (int s = 0; s < T / 16; s++) {
          float xb8[8];
          const float *bp = AT + (wave * 32 + l31) * RP + 16 * s + 8 * h;
#pragma unroll
          for (int j = 0; j < 8; j++) xb8[j] = bp[j];
          bf16x8 bh, bl;
          bf_split8(xb8, bh, bl, true);
#pragma unroll
          for (int it = 0; it < NTW; it++) {
            float xa8[8];
            const float *ap = DY + (it * 32 + l31) * RP + 16 * s + 8 * h;
#pragma unroll
            for (int j = 0; j < 8; j++) xa8[j] = ap[j];
            bf16x8 ah, al;
            bf_split8(xa8, ah, al, true);
            acc[it] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc[it], 0, 0, 0);
            acc[it] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc[it], 0, 0, 0);
            acc[it] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[it], 0, 0, 0);
          }
        }
      }
    } else
    if (a.dwp && !(a.dbg & 1)) {
#pragma unroll
      for (int it = 0; it < NTW; it++) {
        const int item = z * 4 * NTW + wave + 4 * it;
        if (item < items) {
          const int ob = item / nIB, ib = item - ob * nIB;
          const float *ap = DY + (ob * 32 + l31) * RP + h;
          const float *bp = AT + (ib * 32 + l31) * RP + h;
#pragma unroll 8
          for (int ks = 0; ks < T / 2; ks++)
            acc[it] = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[2 * ks], bp[2 * ks], acc[it], 0, 0, 0);
        }
      }
    }
    if (want_dx && !(a.dbg & 2)) {
      // (the barrier between this call's k-loop and its epilogue also orders the dW reads of DY above before the
      // in-place overwrite)
      if constexpr (BF) {
        auto epi_bf = [&](float v, int o, int t) { DY[o * RP + t] = v; };
        tile_dense_bf_impl<TB, DenseShape<NRX, WSX>::nr, DenseShape<NRX, WSX>::ways, false, 3, decltype(epi_bf), PCR_TDBF_PF>(
            DY, a.cout, a.wpT_bf, cinP, true, epi_bf);
      }
      else
      tile_dense2<TB, NRX, WSX>(DY, ceil8(a.cout), a.wpT, cinP, true, [&](float v, int o, int t) { DY[o * RP + t] = v; });
      __syncthreads();
      if (a.dstats && !(a.dbg & 4)) {   // sums of the masked dx and of dx * (raw input) per input channel, quarter rows
        const int q16 = 16 * (tid & 3);
#pragma unroll
        for (int p = 0; p < kRowPasses; p++) {
          const int r = (tid >> 2) + 64 * p;
          if (r < cin1) {
            const float *dr = DY + r * RP + q16, *ar = AT + r * RP + q16;
            const float sh = s_ish[r], inv = s_iinv[r];
            float p0 = 0.f, q0 = 0.f;
#pragma unroll
            for (int t = 0; t < 16; t++) {
              const float av = ar[t];
              const float v = (q16 + t < valid && (!in_relu || av > 0.f)) ? dr[t] : 0.f;
              p0 += v;
              q0 += v * ((av - sh) * inv);
            }
            s1[p] += p0;
            s2[p] += q0;
          }
        }
      }
      if (a.dbg & 8) {
      } else if (vec) {
        constexpr int Q = T / 4;
        const int nq = valid >> 2;
        for (int e = tid; e < cin * Q; e += kThreads) {
          const int c = e / Q, q = e - c * Q;
          if (q >= nq) continue;
          const float *ds = DY + c * RP + 4 * q, *as = AT + c * RP + 4 * q;
          f32x4 v = {ds[0], ds[1], ds[2], ds[3]};
          if (c < cin1) {
            if (in_relu) {
#pragma unroll
              for (int j = 0; j < 4; j++) v[j] = as[j] > 0.f ? v[j] : 0.f;
            }
            *reinterpret_cast<f32x4 *>(a.dx + (b * cin1 + c) * L + t0 + 4 * q) = v;
          } else {
            *reinterpret_cast<f32x4 *>(a.dx2 + (b * a.cin2 + (c - cin1)) * L + t0 + 4 * q) = v;
          }
        }
      } else {
        for (int e = tid; e < cin * T; e += kThreads) {
          const int c = e / T, t = e - c * T;
          if (t < valid) {
            float v = DY[c * RP + t];
            if (c < cin1) {
              if (in_relu && !(AT[c * RP + t] > 0.f)) v = 0.f;
              a.dx[(b * cin1 + c) * L + t0 + t] = v;
            } else {
              a.dx2[(b * a.cin2 + (c - cin1)) * L + t0 + t] = v;
            }
          }
        }
      }
    }
  }
  }   // clouds
  const size_t wg = (size_t)blockIdx.y * gridDim.x + blockIdx.x;
  if (a.dwp) {
    float *dw = a.dwp + wg * (a.dw_stride ? (size_t)a.dw_stride : (size_t)coutP * cinP);
#pragma unroll
    for (int it = 0; it < NTW; it++) {
      const int item = z * 4 * NTW + wave + 4 * it;
      if (item < items) {
        const int ob = item / nIB, ib = item - ob * nIB;
#pragma unroll
        for (int r = 0; r < 16; r++)
          dw[(size_t)(ob * 32 + (r & 3) + 8 * (r >> 2) + 4 * h) * cinP + ib * 32 + l31] = acc[it][r];
      }
    }
  }
  if (a.dbp && z == 0) {
    float *db = a.dbp + wg * (a.db_stride ? (size_t)a.db_stride : (size_t)coutP);
#pragma unroll
    for (int p = 0; p < kRowPasses; p++) {
      const int r = (tid >> 2) + 64 * p;
      const float v = quad_sum(dbsum[p]);
      if ((tid & 3) == 0 && r < coutP) db[r] = v;
    }
  }
  if (a.dstats && want_dx) {
    const int c1P = ceil32(cin1);
    float *sp = a.dstats + wg * 2 * (size_t)c1P;
#pragma unroll
    for (int p = 0; p < kRowPasses; p++) {
      const int r = (tid >> 2) + 64 * p;
      const float v1 = quad_sum(s1[p]), v2 = quad_sum(s2[p]);
      if ((tid & 3) == 0 && r < c1P) {
        sp[r] = r < cin1 ? v1 : 0.f;
        sp[c1P + r] = r < cin1 ? v2 : 0.f;
      }
    }
  }
}

template <int WSX, int NRX, int NTW, int QY, int QX>
__global__ __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(2, 2))) void tdense_bwd_kernel(TBwd a) {
  tdense_bwd_body<WSX, NRX, NTW, QY, QX>(a);
}
__global__ __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(2, 2))) void tdense_bwd_bf_kernel(TBwd a) {
  tdense_bwd_body<1, 1, 4, 0, 0, true>(a);
}
// narrow layers (at most four dW tiles: one accumulator tile per wave): held to a quarter of the register file, so that
// four workgroups share a CU and the load / LDS-commit / matrix / store phases of different workgroups overlap -- the
// phases of ONE workgroup are strictly serial (ablation: 0.32 + 0.29 + 0.07 + 0.27 ms add up to the 0.93 ms launch)
template <int WSX, int NRX>
__global__ __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(4, 4))) void tdense_bwd_kernel_o4(TBwd a) {
  tdense_bwd_body<WSX, NRX, 1, 0, 0>(a);
}

// (prefetch variant with ONE dW accumulator tile per wave: layers of at most four dW tiles)
template <int WSX, int NRX, int QY, int QX>
__global__ __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(2, 2))) void tdense_bwd_kernel_p1(TBwd a) {
  tdense_bwd_body<WSX, NRX, 1, QY, QX>(a);
}
template <int WSX, int NRX>
__global__ __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(3, 3))) void tdense_bwd_kernel_o3(TBwd a) {
  tdense_bwd_body<WSX, NRX, 1, 0, 0>(a);
}

// ------------------------------------------------------------------- reductions / finalisers ----
// out[e] = sum over p of part[p][e] in a FIXED order (reproducible): block = 32 elements x 8 part lanes, lane pl adds
// parts pl, pl + 8, ... on four interleaved chains, the eight lane totals are combined in lane order.  Optional row
// gather: element e = (row r, col c) of a (rows x cols) result reads part[p][r * ld + c].
__global__ __launch_bounds__(256) void reduce_parts_kernel(const float *__restrict__ part, int nparts, size_t stride,
                                                           int rows, int cols, int ld, float *__restrict__ out) {
  __shared__ float red[8][32];
  const int el = threadIdx.x & 31, pl = threadIdx.x >> 5;
  const int e = blockIdx.x * 32 + el;
  const bool ok = e < rows * cols;
  const int r = ok ? e / cols : 0, c = ok ? e - r * cols : 0;
  const float *p = part + (size_t)r * ld + c;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  int q = pl;
  for (; q + 24 < nparts; q += 32) {
    const float v0 = p[(size_t)q * stride], v1 = p[(size_t)(q + 8) * stride], v2 = p[(size_t)(q + 16) * stride],
                v3 = p[(size_t)(q + 24) * stride];
    s0 += v0;
    s1 += v1;
    s2 += v2;
    s3 += v3;
  }
  for (; q < nparts; q += 8) s0 += p[(size_t)q * stride];
  red[pl][el] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (pl == 0 && ok) {
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < 8; i++) t += red[i][el];
    out[e] = t;
  }
}

// The same for a dense result (rows x cols contiguous, 16-byte aligned partial rows): block = 8 quads of elements x 32
// part lanes, 16-byte loads on eight interleaved chains per lane (~64 KB in flight per CU instead of ~12: the scalar
// form above is latency-bound on the large dW partial sets, 64 MB in 45 us).
__global__ __launch_bounds__(256) void reduce_parts4_kernel(const float *__restrict__ part, int nparts, size_t stride,
                                                            int total, float *__restrict__ out) {
  __shared__ f32x4 red[32][8];
  const int ql = threadIdx.x & 7, pl = threadIdx.x >> 3;
  const int e = (blockIdx.x * 8 + ql) * 4;
  const bool ok = e < total;
  const float *p = part + (ok ? e : 0);
  f32x4 s[8];
#pragma unroll
  for (int j = 0; j < 8; j++) s[j] = f32x4{0.f, 0.f, 0.f, 0.f};
  int q = pl;
  for (; q + 224 < nparts; q += 256) {
    f32x4 v[8];     // (all eight loads first: written as `s[j] += load` the compiler waits for each one in turn)
#pragma unroll
    for (int j = 0; j < 8; j++) v[j] = ld4(p + (size_t)(q + 32 * j) * stride);
#pragma unroll
    for (int j = 0; j < 8; j++) s[j] += v[j];
  }
  for (; q < nparts; q += 32) s[0] += ld4(p + (size_t)q * stride);
  red[pl][ql] = ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
  __syncthreads();
  if (pl == 0 && ok) {
    f32x4 t = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 32; i++) t += red[i][ql];
    if (e + 3 < total) *reinterpret_cast<f32x4 *>(out + e) = t;
    else
      for (int j = 0; e + j < total; j++) out[e + j] = t[j];
  }
}

// partial sums [nparts][2][CP] -> per-channel totals in double.  Block = 32 channels x 32 part lanes; the lane
// totals are combined in lane order (fixed => reproducible).
__device__ __forceinline__ void sum_parts2(const float *part, int nparts, int CP, int c, double &t0, double &t1,
                                           double (*red)[32][32]) {
  const int cl = threadIdx.x & 31, pg = threadIdx.x >> 5;
  double a0 = 0.0, a1 = 0.0;
  for (int p = pg; p < nparts; p += 32) {
    a0 += (double)part[((size_t)p * 2) * CP + c];
    a1 += (double)part[((size_t)p * 2 + 1) * CP + c];
  }
  red[0][pg][cl] = a0;
  red[1][pg][cl] = a1;
  __syncthreads();
  t0 = 0.0;
  t1 = 0.0;
  for (int i = 0; i < 32; i++) {
    t0 += red[0][i][cl];
    t1 += red[1][i][cl];
  }
}

struct BnFwdFin {
  const float *part;
  int nparts, CP, C;
  double R;
  const float *gamma, *beta;
  float eps, momentum;
  float *running_mean, *running_var;    // updated in place (may be null)
  float *scale, *shift, *inv_scale, *mean, *invstd;
  const float *shift0;                  // the sums are of (y - shift0[c * shift0_stride]) and its square (or null)
  int shift0_stride;
};

__global__ __launch_bounds__(1024) void bn_fwd_finalize_kernel(BnFwdFin a) {
  __shared__ double red[2][32][32];
  const int c = blockIdx.x * 32 + (threadIdx.x & 31);
  double s, sq;
  sum_parts2(a.part, a.nparts, a.CP, c < a.CP ? c : 0, s, sq, red);
  if ((threadIdx.x >> 5) != 0 || c >= a.C) return;
  const double mean0 = s / a.R;          // (of the shifted values: the variance does not see the shift)
  double var = sq / a.R - mean0 * mean0;
  if (var < 0.0) var = 0.0;
  const double mean = mean0 + (a.shift0 ? (double)a.shift0[(size_t)c * a.shift0_stride] : 0.0);
  const double invstd = 1.0 / sqrt(var + (double)a.eps);
  const double sc = (double)a.gamma[c] * invstd;
  a.scale[c] = (float)sc;
  a.shift[c] = (float)((double)a.beta[c] - mean * sc);
  a.inv_scale[c] = sc != 0.0 ? (float)(1.0 / sc) : 0.f;
  a.mean[c] = (float)mean;
  a.invstd[c] = (float)invstd;
  if (a.running_mean) {
    const double m = a.momentum;
    a.running_mean[c] = (float)((1.0 - m) * (double)a.running_mean[c] + m * mean);
    const double unb = a.R > 1.0 ? var * a.R / (a.R - 1.0) : var;
    a.running_var[c] = (float)((1.0 - m) * (double)a.running_var[c] + m * unb);
  }
}

// BatchNorm backward constants from S1 = sum dyhat, S2 = sum dyhat * y (y = the raw pre-activation):
//   dbeta = S1, dgamma = invstd (S2 - mean S1), dy = ka dyhat + kb y + kc with
//   ka = gamma invstd, kb = -ka dgamma invstd / R, kc = ka (dgamma invstd mean - dbeta) / R
struct BnBwdFin {
  const float *part;
  int nparts, CP, C;
  double R;
  const float *gamma, *mean, *invstd;
  float *ka, *kb, *kc, *dgamma, *dbeta;
  const float *centre;                  // S2 = sum dyhat * (y - centre[c]) (or null: centre 0)
};

__global__ __launch_bounds__(1024) void bn_bwd_finalize_kernel(BnBwdFin a) {
  __shared__ double red[2][32][32];
  const int c = blockIdx.x * 32 + (threadIdx.x & 31);
  double s1, s2;
  sum_parts2(a.part, a.nparts, a.CP, c < a.CP ? c : 0, s1, s2, red);
  if ((threadIdx.x >> 5) != 0 || c >= a.C) return;
  const double mean = a.mean[c], invstd = a.invstd[c], gamma = a.gamma[c];
  const double dgamma = invstd * (s2 - (mean - (a.centre ? (double)a.centre[c] : 0.0)) * s1);
  const double ka = gamma * invstd;
  a.dgamma[c] = (float)dgamma;
  a.dbeta[c] = (float)s1;
  a.ka[c] = (float)ka;
  a.kb[c] = (float)(-ka * dgamma * invstd / a.R);
  a.kc[c] = (float)(ka * (dgamma * invstd * mean - s1) / a.R);
}

// W (rows x cols, row-major, leading dimension ld) -> packed MFMA A-operand image of W (transpose = 0: cout = rows,
// cin = cols) or of W^T (transpose = 1: cout = cols, cin = rows): element (kb, o, h, j) = M[o][kb*8 + 2*j + h]
__global__ void pack_weight_kernel(const float *__restrict__ w, int rows, int cols, int ld, int transpose,
                                   float *__restrict__ packed) {
  // transpose = 2: both images, W first (ceil8(cols) * ceil32(rows) floats), then W^T
  const int n0 = ceil8(cols) * ceil32(rows), n1 = ceil8(rows) * ceil32(cols);
  const int total = transpose == 2 ? n0 + n1 : (transpose ? n1 : n0);
  for (int e0 = blockIdx.x * blockDim.x + threadIdx.x; e0 < total; e0 += gridDim.x * blockDim.x) {
    const bool tr = transpose == 1 || (transpose == 2 && e0 >= n0);
    const int e = (transpose == 2 && e0 >= n0) ? e0 - n0 : e0;
    const int cout = tr ? cols : rows, cin = tr ? rows : cols;
    const int OP = ceil32(cout);
    const int j = e & 3, hh = (e >> 2) & 1, o = (e >> 3) % OP, kb = (e >> 3) / OP;
    const int k = kb * 8 + 2 * j + hh;
    float v = 0.f;
    if (o < cout && k < cin) v = tr ? w[(size_t)k * ld + o] : w[(size_t)o * ld + k];
    packed[e0] = v;
  }
}

// the bf16 hi / lo image (pcr_pack_weight_bf16x2_f32's layout: [ceil16(cin) / 16 steps][ceil32(cout) / 32][hi, lo][64
// lanes][8 bf16], channel 16 s + bf_kpos(lane >> 5, e) in element e) of W (transpose = 0) or W^T (1), on the device
__global__ void pack_weight_bf_kernel(const float *__restrict__ w, int rows, int cols, int ld, int transpose,
                                      unsigned short *__restrict__ packed) {
  const int cout = transpose ? cols : rows, cin = transpose ? rows : cols;
  const int S = (cin + 15) >> 4, nCB = ceil32(cout) >> 5;
  const int total = S * nCB * 64 * 8;      // (hi, lo) pairs
  for (int e0 = blockIdx.x * blockDim.x + threadIdx.x; e0 < total; e0 += gridDim.x * blockDim.x) {
    const int e = e0 & 7, lane = (e0 >> 3) & 63, cb = (e0 >> 9) % nCB, s = (e0 >> 9) / nCB;
    const int o = cb * 32 + (lane & 31), k = 16 * s + bf_kpos(lane >> 5, e);
    float v = 0.f;
    if (o < cout && k < cin) v = transpose ? w[(size_t)k * ld + o] : w[(size_t)o * ld + k];
    const __bf16 hi = (__bf16)v;
    const __bf16 lo = (__bf16)(v - (float)hi);
    const size_t u = ((size_t)(s * nCB + cb) * 2) * 64 + lane;
    packed[u * 8 + e] = __builtin_bit_cast(unsigned short, hi);
    packed[(u + 64) * 8 + e] = __builtin_bit_cast(unsigned short, lo);
  }
}

// every weight of a model in ONE launch: blockIdx.y = tensor (descriptor table on the device), both images per tensor
struct PackDesc {      // = pcr_pack_desc
  const float *w;
  float *out;
  int rows, cols;
  int kind, reserved;
};
static_assert(sizeof(PackDesc) == sizeof(pcr_pack_desc), "pcr_pack_desc layout");

// one (hi, lo) element pair of the bf16 image of W (transpose = 0) or W^T (1): pack_weight_bf_kernel's body
__device__ __forceinline__ void pack_bf_pair(const float *__restrict__ w, int rows, int cols, int transpose, int e0,
                                             unsigned short *__restrict__ packed) {
  const int cout = transpose ? cols : rows, cin = transpose ? rows : cols;
  const int nCB = ceil32(cout) >> 5;
  const int e = e0 & 7, lane = (e0 >> 3) & 63, cb = (e0 >> 9) % nCB, s = (e0 >> 9) / nCB;
  const int o = cb * 32 + (lane & 31), k = 16 * s + bf_kpos(lane >> 5, e);
  float v = 0.f;
  if (o < cout && k < cin) v = transpose ? w[(size_t)k * cols + o] : w[(size_t)o * cols + k];
  const __bf16 hi = (__bf16)v;
  const __bf16 lo = (__bf16)(v - (float)hi);
  const size_t u = ((size_t)(s * nCB + cb) * 2) * 64 + lane;
  packed[u * 8 + e] = __builtin_bit_cast(unsigned short, hi);
  packed[(u + 64) * 8 + e] = __builtin_bit_cast(unsigned short, lo);
}

__global__ __launch_bounds__(256) void pack_weights_multi_kernel(const PackDesc *__restrict__ descs) {
  const PackDesc d = descs[blockIdx.y];
  const int rows = d.rows, cols = d.cols;
  if (cols == 0) {     // a bias vector: `rows` floats into the head of its zero-padded image
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < rows; e += gridDim.x * blockDim.x) d.out[e] = d.w[e];
    return;
  }
  if (d.kind == 1) {   // bf16 hi / lo images: W, then W^T
    const int p0 = ((cols + 15) >> 4) * (ceil32(rows) >> 5) * 512, p1 = ((rows + 15) >> 4) * (ceil32(cols) >> 5) * 512;
    unsigned short *img = reinterpret_cast<unsigned short *>(d.out);
    for (int e0 = blockIdx.x * blockDim.x + threadIdx.x; e0 < p0 + p1; e0 += gridDim.x * blockDim.x) {
      if (e0 < p0) pack_bf_pair(d.w, rows, cols, 0, e0, img);
      else pack_bf_pair(d.w, rows, cols, 1, e0 - p0, img + (size_t)p0 * 2);
    }
    return;
  }
  const int n0 = ceil8(cols) * ceil32(rows), n1 = ceil8(rows) * ceil32(cols);
  for (int e0 = blockIdx.x * blockDim.x + threadIdx.x; e0 < n0 + n1; e0 += gridDim.x * blockDim.x) {
    const bool tr = e0 >= n0;
    const int e = tr ? e0 - n0 : e0;
    const int cout = tr ? cols : rows, cin = tr ? rows : cols;
    const int OP = ceil32(cout);
    const int j = e & 3, hh = (e >> 2) & 1, o = (e >> 3) % OP, kb = (e >> 3) / OP;
    const int k = kb * 8 + 2 * j + hh;
    float v = 0.f;
    if (o < cout && k < cin) v = tr ? d.w[(size_t)k * cols + o] : d.w[(size_t)o * cols + k];
    d.out[e0] = v;
  }
}

template <class K>
static bool big_lds(K k) {
  return allow_big_lds(k);
}

}  // namespace

// grid.y: clouds are strided over at most this many workgroup rows, so that a launch has ~768 workgroups whatever the
// batch (small tensors: a workgroup then carries its weights, dW accumulators and statistics across several clouds)
static int wg_cloud_rows(int B, int gx, long part_floats = 0) {
  // (measured: letting wide layers on short clouds use fewer rows, to shrink their per-workgroup dW partials, costs
  // more in lost parallelism than the partial traffic it saves -- 0.59 -> 0.70 ms on the 256 x 256, L = 32 layer)
  // (re-measured with the vector fill of ragged tiles: 512 .. 2048 rows are all within noise on the whole step)
  (void)part_floats;
  int r = 768 / (gx > 0 ? gx : 1);
  if (r < 1) r = 1;
  return r < B ? r : B;
}

static int wg_groups(int B, int ntiles) {
  // workgroups per cloud: enough to fill the chip a few times over, few enough that the per-workgroup partials
  // (dW images, statistics) stay small: ~1024 workgroups per launch
  int g = (1024 + B - 1) / (B > 0 ? B : 1);
  if (g < 1) g = 1;
  if (g > ntiles) g = ntiles;
  return g;
}

PCR_EXPORT int pcr_train_groups(int B, int L) {
  const int ntiles = (L + kTT - 1) / kTT;
  const int g = wg_groups(B, ntiles);
  const int tpw = (ntiles + g - 1) / g;
  const int gx = (ntiles + tpw - 1) / tpw;
  return gx * wg_cloud_rows(B, gx);   // workgroups of a train-dense launch = rows of its partial buffers
}

// the same for a backward launch that accumulates dW (cout x cin): wide layers use fewer workgroup rows
PCR_EXPORT int pcr_train_groups_bwd(int B, int L, int cout, int cin) {
  const int ntiles = (L + kTT - 1) / kTT;
  const int g = wg_groups(B, ntiles);
  const int tpw = (ntiles + g - 1) / g;
  const int gx = (ntiles + tpw - 1) / tpw;
  const long pf = cout > 0 ? (long)ceil32(cout) * ceil32(cin) + ceil32(cout) : 0;
  return gx * wg_cloud_rows(B, gx, pf);
}

PCR_EXPORT int pcr_tdense_fwd_groups(const pcr_tdense_fwd *p) {
  if (!p) return 0;
  if (pcr_ts_fwd_ok(p)) return pcr_ts_fwd_grid(p, nullptr);
  return pcr_train_groups(p->B, p->L);
}

PCR_EXPORT int pcr_tdense_fwd_pooled(const pcr_tdense_fwd *p) { return p && pcr_ts_fwd_pools(p) ? 1 : 0; }

PCR_EXPORT int pcr_tdense_bwd_groups(const pcr_tdense_bwd *p) {
  if (!p) return 0;
  if (pcr_ts_bwd_ok(p)) return pcr_ts_bwd_grid(p, nullptr);
  return pcr_train_groups_bwd(p->B, p->L, p->dwp ? p->cout : 0, p->cin1 + p->cin2);
}

PCR_EXPORT int pcr_pack_weight_dev_f32(const float *w, int rows, int cols, int ld, int transpose, float *packed,
                                       pcr_stream_t stream) {
  if (!w || !packed || rows < 1 || cols < 1 || ld < cols || transpose < 0 || transpose > 2) return PCR_ERR_INVALID;
  const int n0 = ceil8(cols) * ceil32(rows), n1 = ceil8(rows) * ceil32(cols);
  const int total = transpose == 2 ? n0 + n1 : (transpose ? n1 : n0);
  hipLaunchKernelGGL(pack_weight_kernel, dim3((total + 255) / 256 > 1024 ? 1024 : (total + 255) / 256), dim3(256), 0,
                     pcr_s(stream), w, rows, cols, ld, transpose, packed);
  PCR_CHECK_LAUNCH();
  return PCR_OK;
}

PCR_EXPORT int pcr_pack_weight_bf16_dev_f32(const float *w, int rows, int cols, int ld, int transpose, float *packed,
                                            pcr_stream_t stream) {
  if (!w || !packed || rows < 1 || cols < 1 || ld < cols || transpose < 0 || transpose > 1) return PCR_ERR_INVALID;
  const int cout = transpose ? cols : rows, cin = transpose ? rows : cols;
  const int total = ((cin + 15) >> 4) * (ceil32(cout) >> 5) * 512;
  hipLaunchKernelGGL(pack_weight_bf_kernel, dim3((total + 255) / 256 > 1024 ? 1024 : (total + 255) / 256), dim3(256), 0,
                     pcr_s(stream), w, rows, cols, ld, transpose, reinterpret_cast<unsigned short *>(packed));
  PCR_CHECK_LAUNCH();
  return PCR_OK;
}

PCR_EXPORT int pcr_pack_weights_multi_f32(const pcr_pack_desc *descs_dev, int n, pcr_stream_t stream) {
  if (!descs_dev || n < 0 || n > 65535) return PCR_ERR_INVALID;
  if (n == 0) return PCR_OK;
  hipLaunchKernelGGL(pack_weights_multi_kernel, dim3(8, n), dim3(256), 0, pcr_s(stream),
                     reinterpret_cast<const PackDesc *>(descs_dev));
  PCR_CHECK_LAUNCH();
  return PCR_OK;
}

PCR_EXPORT int pcr_tdense_fwd_f32(const pcr_tdense_fwd *p, pcr_stream_t stream) {
  if (!p || !p->x || !p->wp || !p->y || p->B < 0 || p->cin1 < 1 || p->cin2 < 0 || p->cout < 1 || p->cout > 384 ||
      p->L < 1 || (p->cin2 && !p->x2) || (p->isc && !p->ish))
    return PCR_ERR_INVALID;
  if (p->stats && p->cout > 256) return PCR_ERR_INVALID;   // (statistics are taken by one thread per cout row)
  if (p->B == 0) return PCR_OK;
  if (p->B > 65535) return PCR_ERR_INVALID;
  pcr_note_arith(PCR_PREC_F32);
  if (pcr_ts_fwd_ok(p)) return pcr_ts_fwd_launch(p, pcr_s(stream));
  TFwd a;
  a.x = p->x; a.x2 = p->cin2 ? p->x2 : nullptr; a.cin1 = p->cin1; a.cin2 = p->cin2;
  a.isc = p->isc; a.ish = p->ish; a.in_relu = p->in_relu;
  a.wp = p->wp; a.bias = p->bias; a.res = p->res; a.out_relu = p->out_relu;
  a.y = p->y; a.cout = p->cout; a.L = p->L; a.stats = p->stats; a.B = p->B;
  const int ntiles = (p->L + kTT - 1) / kTT;
  const int g = wg_groups(p->B, ntiles);
  a.tpw = (ntiles + g - 1) / g;
  const int gx = (ntiles + a.tpw - 1) / a.tpw;
  const int cinP = ceil8(p->cin1 + p->cin2), coutP = ceil32(p->cout);
  const size_t lds = ((size_t)(cinP > coutP ? cinP : coutP) * kTRP + 2 * (size_t)p->cin1) * sizeof(float);
  if (lds > (size_t)kMaxDynLds) return PCR_ERR_INVALID;
  const int need = (cinP + 15) / 16;     // 16-byte pieces per thread of one input tile
  const int pfq = need <= 2 ? 2 : (need <= 4 ? 4 : (need <= 8 ? 8 : 0));
  const dim3 grid(gx, wg_cloud_rows(p->B, gx)), blk(kThreads);
  hipStream_t st = pcr_s(stream);
#define PCR_TF(WSv, NRv, Qv)                                                        \
  do {                                                                              \
    static bool ok = big_lds(tdense_fwd_kernel<WSv, NRv, Qv>);                      \
    (void)ok;                                                                       \
    hipLaunchKernelGGL((tdense_fwd_kernel<WSv, NRv, Qv>), grid, blk, lds, st, a);   \
  } while (0)
#define PCR_TFQ(WSv, NRv)                                                           \
  do {                                                                              \
    if (pfq == 2) PCR_TF(WSv, NRv, 2);                                              \
    else if (pfq == 4) PCR_TF(WSv, NRv, 4);                                         \
    else if (pfq == 8) PCR_TF(WSv, NRv, 8);                                         \
    else PCR_TF(WSv, NRv, 0);                                                       \
  } while (0)
  const int nb = coutP >> 5;
  if (nb == 1) PCR_TFQ(4, 1);
  else if (nb == 2) PCR_TFQ(2, 1);
  else if (nb <= 4) PCR_TFQ(1, 1);
  else if (nb <= 8) PCR_TFQ(1, 2);
  else PCR_TFQ(1, 3);
#undef PCR_TFQ
#undef PCR_TF
  PCR_CHECK_LAUNCH();
  return PCR_OK;
}

PCR_EXPORT int pcr_tdense_bwd_f32(const pcr_tdense_bwd *p, pcr_stream_t stream) {
  if (!p || !p->g || !p->x || p->B < 0 || p->cin1 < 1 || p->cin2 < 0 || p->cout < 1 || p->cout > 384 || p->L < 1 ||
      p->cin1 + p->cin2 > 288 || (p->cin2 && !p->x2) || p->dy_mode < 0 || p->dy_mode > 3)
    return PCR_ERR_INVALID;
  if ((p->dy_mode == 1 || p->dy_mode == 3) && (!p->ka || !p->kb || !p->kc || !p->y)) return PCR_ERR_INVALID;
  if (p->dy_mode == 2 && !p->y) return PCR_ERR_INVALID;
  if (p->dy_mode == 3 && (!p->argmax || p->K < 1 || p->S < 1 || p->S * p->K != p->L)) return PCR_ERR_INVALID;
  if (p->wpT && (!p->dx || (p->cin2 && !p->dx2))) return PCR_ERR_INVALID;
  if (p->dwp && !p->dbp) return PCR_ERR_INVALID;
  if (p->B == 0) return PCR_OK;
  if (p->B > 65535) return PCR_ERR_INVALID;
  pcr_note_arith(PCR_PREC_F32);   // (every training launch but the 128 x 128 bf16 backward below)
  if (pcr_ts_bwd_ok(p)) return pcr_ts_bwd_launch(p, pcr_s(stream));
  TBwd a;
  a.g = p->g; a.y = p->y; a.dy_mode = p->dy_mode; a.ka = p->ka; a.kb = p->kb; a.kc = p->kc;
  a.argmax = p->argmax; a.pooled = p->pooled; a.K = p->K; a.S = p->S;
  a.x = p->x; a.x2 = p->cin2 ? p->x2 : nullptr; a.cin1 = p->cin1; a.cin2 = p->cin2;
  a.isc = p->isc; a.ish = p->ish; a.iinv = p->iinv; a.in_relu = p->in_relu;
  a.wpT = p->wpT; a.wpT_bf = p->wpT_bf; a.dx = p->dx; a.dx2 = p->dx2; a.dstats = p->dstats; a.dwp = p->dwp; a.dbp = p->dbp;
  a.dw_stride = p->part_stride; a.db_stride = p->part_stride;
  a.cout = p->cout; a.L = p->L; a.B = p->B;
  static const int dbg = pcr_tune_int("PCR_TD_DBG");
  a.dbg = dbg;
  const int ntiles = (p->L + kTT - 1) / kTT;
  const int g = wg_groups(p->B, ntiles);
  a.tpw = (ntiles + g - 1) / g;
  const int gx = (ntiles + a.tpw - 1) / a.tpw;
  const int cinP = ceil32(p->cin1 + p->cin2), coutP = ceil32(p->cout);
  const size_t lds = ((size_t)((coutP > cinP ? coutP : cinP) + cinP) * kTRP + 3 * (size_t)p->cout + 3 * (size_t)p->cin1) *
                     sizeof(float);
  if (lds > (size_t)kMaxDynLds) return PCR_ERR_INVALID;
  constexpr int NTW = 4;
  const int items = (coutP >> 5) * (cinP >> 5);
  const int gz = p->dwp ? (items + 4 * NTW - 1) / (4 * NTW) : 1;
  const dim3 grid(gx, wg_cloud_rows(p->B, gx, p->dwp ? (long)coutP * cinP + coutP : 0), gz), blk(kThreads);
  hipStream_t st = pcr_s(stream);
  const int nx = cinP >> 5;
  const int rowsY = coutP > cinP ? coutP : cinP;
#define PCR_TB(WSv, NRXv, QYv, QXv)                                                              \
  do {                                                                                           \
    static bool ok = big_lds(tdense_bwd_kernel<WSv, NRXv, NTW, QYv, QXv>);                       \
    (void)ok;                                                                                    \
    hipLaunchKernelGGL((tdense_bwd_kernel<WSv, NRXv, NTW, QYv, QXv>), grid, blk, lds, st, a);    \
  } while (0)
#define PCR_TB4(WSv, NRXv)                                                              \
  do {                                                                                  \
    static bool ok = big_lds(tdense_bwd_kernel_o4<WSv, NRXv>);                          \
    (void)ok;                                                                           \
    hipLaunchKernelGGL((tdense_bwd_kernel_o4<WSv, NRXv>), grid, blk, lds, st, a);       \
  } while (0)
  static const int variant = pcr_tune_int("PCR_TD_VARIANT");   // tuning aid
  // narrow layers (<= 4 dW tiles, LDS <= 40 KB): four workgroups per CU; wider ones: two per CU (with operand prefetch
  // for the 64-channel square layers whose pieces fit the registers)
  // (measured at B = 512: 64 x 64, L = 3072: 0.94 -> 0.72 ms with four workgroups per CU, 0.63 with the batched fill;
  // 32 x 32, L = 4096: 0.59 with the prefetch variant against 0.62)
  // (64-wide inputs: three workgroups per CU without scratch beat four with 136 B of spills per lane, 0.61 vs 0.63 ms)
  if (variant != 2 && items <= 4 && lds <= 40 * 1024 && nx == 2) {
    static bool ok3 = big_lds(tdense_bwd_kernel_o3<2, 1>);
    (void)ok3;
    hipLaunchKernelGGL((tdense_bwd_kernel_o3<2, 1>), grid, blk, lds, st, a);
  } else
  if (items <= 4 && lds <= 40 * 1024 && variant != 2 && !(rowsY == 32 && cinP == 32)) {
    if (nx == 1) PCR_TB4(4, 1);
    else PCR_TB4(2, 1);
  } else if (rowsY == 32 && cinP == 32) {
    // one accumulator tile per wave (NTW = 4 carried three dead tiles and 160 B of spills): 0.58 -> 0.49 ms
    static bool okp = big_lds(tdense_bwd_kernel_p1<4, 1, 2, 2>);
    (void)okp;
    hipLaunchKernelGGL((tdense_bwd_kernel_p1<4, 1, 2, 2>), grid, blk, lds, st, a);
  }
  else if (rowsY == 64 && cinP == 64) PCR_TB(2, 1, 4, 4);
  else if (p->precision == PCR_PREC_BF16X3 && p->wpT_bf && p->wpT && coutP == 128 && cinP == 128 && p->cout == 128 &&
           p->cin1 == 128 && !p->cin2 && gz == 1) {
    // the 128 x 128 grouped-MLP layers on the bf16 matrix core (split bf16: dx and dW)
    static bool okb = big_lds(tdense_bwd_bf_kernel);
    (void)okb;
    pcr_note_arith(PCR_PREC_BF16X3);
    hipLaunchKernelGGL(tdense_bwd_bf_kernel, grid, blk, lds, st, a);
  }
  else if (nx == 1) PCR_TB(4, 1, 0, 0);
  else if (nx == 2) PCR_TB(2, 1, 0, 0);
  else if (nx <= 4) PCR_TB(1, 1, 0, 0);
  else if (nx <= 8) PCR_TB(1, 2, 0, 0);
  else PCR_TB(1, 3, 0, 0);
#undef PCR_TB4
#undef PCR_TB
  PCR_CHECK_LAUNCH();
  return PCR_OK;
}

// Many reductions in ONE launch (round 5): a training step's backward leaves ~40 partial-sum buffers (weight / bias /
// norm gradients), each summed by its own 6 us launch so far.  The jobs travel by value in the kernel arguments (no
// descriptor table to upload); workgroup (x, y) reduces 32 elements of job y, jobs shorter than x 32 elements leave.
constexpr int kReduceJobs = 64;
struct ReduceJobs {
  pcr_reduce_job j[kReduceJobs];
};

__global__ __launch_bounds__(256) void reduce_multi_kernel(ReduceJobs jobs) {
  __shared__ f32x4 red4[32][8];
  const pcr_reduce_job jb = jobs.j[blockIdx.y];
  const int total = jb.rows * jb.cols;
  if ((int)blockIdx.x * 32 >= total) return;
  const size_t stride = (size_t)jb.stride;
  const bool vec = ((jb.cols | jb.ld) & 3) == 0 && (stride & 3) == 0 &&
                   ((reinterpret_cast<uintptr_t>(jb.part) | reinterpret_cast<uintptr_t>(jb.out)) & 15) == 0;
  if (vec) {
    // reduce_parts4_kernel's scheme: 8 quads of elements x 32 part lanes, 16-byte loads on eight chains per lane (the
    // big records -- the chain kernels' dW partials, 30 MB per launch -- need the bytes in flight)
    const int ql = threadIdx.x & 7, pl = threadIdx.x >> 3;
    const int e = (blockIdx.x * 8 + ql) * 4;
    const bool ok = e < total;
    const int r = ok ? e / jb.cols : 0, c = ok ? e - r * jb.cols : 0;
    const float *p = jb.part + (size_t)r * jb.ld + c;
    f32x4 s[8];
#pragma unroll
    for (int k = 0; k < 8; k++) s[k] = f32x4{0.f, 0.f, 0.f, 0.f};
    int q = pl;
    for (; q + 224 < jb.nparts; q += 256) {
      f32x4 v[8];
#pragma unroll
      for (int k = 0; k < 8; k++) v[k] = ld4(p + (size_t)(q + 32 * k) * stride);
#pragma unroll
      for (int k = 0; k < 8; k++) s[k] += v[k];
    }
    for (; q < jb.nparts; q += 32) s[0] += ld4(p + (size_t)q * stride);
    red4[pl][ql] = ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
    __syncthreads();
    if (pl == 0 && ok) {
      f32x4 t = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int i = 0; i < 32; i++) t += red4[i][ql];
      *reinterpret_cast<f32x4 *>(jb.out + e) = t;
    }
    return;
  }
  float (*red)[32] = reinterpret_cast<float (*)[32]>(&red4[0][0]);     // [8][32]
  const int el = threadIdx.x & 31, pl = threadIdx.x >> 5;
  const int e = blockIdx.x * 32 + el;
  const bool ok = e < total;
  const int r = ok ? e / jb.cols : 0, c = ok ? e - r * jb.cols : 0;
  const float *p = jb.part + (size_t)r * jb.ld + c;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  int q = pl;
  for (; q + 24 < jb.nparts; q += 32) {
    const float v0 = p[(size_t)q * stride], v1 = p[(size_t)(q + 8) * stride], v2 = p[(size_t)(q + 16) * stride],
                v3 = p[(size_t)(q + 24) * stride];
    s0 += v0;
    s1 += v1;
    s2 += v2;
    s3 += v3;
  }
  for (; q < jb.nparts; q += 8) s0 += p[(size_t)q * stride];
  red[pl][el] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (pl == 0 && ok) {
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < 8; i++) t += red[i][el];
    jb.out[e] = t;
  }
}

PCR_EXPORT int pcr_reduce_multi_f32(const pcr_reduce_job *jobs, int n, pcr_stream_t stream) {
  if (n < 0 || (n && !jobs)) return PCR_ERR_INVALID;
  for (int i = 0; i < n; i++)
    if (!jobs[i].part || !jobs[i].out || jobs[i].nparts < 1 || jobs[i].rows < 1 || jobs[i].cols < 1 || jobs[i].ld < jobs[i].cols)
      return PCR_ERR_INVALID;
  for (int lo = 0; lo < n; lo += kReduceJobs) {
    const int m = n - lo < kReduceJobs ? n - lo : kReduceJobs;
    ReduceJobs a;
    int most = 0;
    for (int i = 0; i < m; i++) {
      a.j[i] = jobs[lo + i];
      const int total = jobs[lo + i].rows * jobs[lo + i].cols;
      most = total > most ? total : most;
    }
    for (int i = m; i < kReduceJobs; i++) a.j[i] = a.j[0];
    hipLaunchKernelGGL(reduce_multi_kernel, dim3((most + 31) / 32, m), dim3(256), 0, pcr_s(stream), a);
    PCR_CHECK_LAUNCH();
  }
  return PCR_OK;
}

PCR_EXPORT int pcr_reduce_parts_f32(const float *part, int nparts, long stride, int rows, int cols, int ld, float *out,
                                    pcr_stream_t stream) {
  if (!part || !out || nparts < 1 || rows < 1 || cols < 1 || ld < cols) return PCR_ERR_INVALID;
  const int total = rows * cols;
  const bool dense = (rows == 1 || ld == cols) && (stride & 3) == 0 && (total & 3) == 0 &&
                     (reinterpret_cast<uintptr_t>(part) & 15) == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0;
  if (dense && nparts >= 64)
    hipLaunchKernelGGL(reduce_parts4_kernel, dim3((total + 31) / 32), dim3(256), 0, pcr_s(stream), part, nparts,
                       (size_t)stride, total, out);
  else
  hipLaunchKernelGGL(reduce_parts_kernel, dim3((total + 31) / 32), dim3(256), 0, pcr_s(stream), part, nparts,
                     (size_t)stride, rows, cols, ld, out);
  PCR_CHECK_LAUNCH();
  return PCR_OK;
}

PCR_EXPORT int pcr_bn_fwd_finalize_f32(const pcr_bn_fwd_fin *p, pcr_stream_t stream) {
  if (!p || !p->part || p->nparts < 1 || p->C < 1 || p->R < 1 || !p->gamma || !p->beta || !p->scale || !p->shift ||
      !p->inv_scale || !p->mean || !p->invstd)
    return PCR_ERR_INVALID;
  BnFwdFin a;
  a.part = p->part; a.nparts = p->nparts; a.CP = ceil32(p->C); a.C = p->C; a.R = p->R;
  a.gamma = p->gamma; a.beta = p->beta; a.eps = p->eps; a.momentum = p->momentum;
  a.running_mean = p->running_mean; a.running_var = p->running_var;
  a.scale = p->scale; a.shift = p->shift; a.inv_scale = p->inv_scale; a.mean = p->mean; a.invstd = p->invstd;
  a.shift0 = p->shift0; a.shift0_stride = p->shift0_stride;
  hipLaunchKernelGGL(bn_fwd_finalize_kernel, dim3(a.CP / 32), dim3(1024), 0, pcr_s(stream), a);
  PCR_CHECK_LAUNCH();
  return PCR_OK;
}

PCR_EXPORT int pcr_bn_bwd_finalize_f32(const pcr_bn_bwd_fin *p, pcr_stream_t stream) {
  if (!p || !p->part || p->nparts < 1 || p->C < 1 || p->R < 1 || !p->gamma || !p->mean || !p->invstd || !p->ka ||
      !p->kb || !p->kc || !p->dgamma || !p->dbeta)
    return PCR_ERR_INVALID;
  BnBwdFin a;
  a.part = p->part; a.nparts = p->nparts; a.CP = ceil32(p->C); a.C = p->C; a.R = p->R;
  a.gamma = p->gamma; a.mean = p->mean; a.invstd = p->invstd;
  a.ka = p->ka; a.kb = p->kb; a.kc = p->kc; a.dgamma = p->dgamma; a.dbeta = p->dbeta;
  a.centre = p->centre;
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(a.CP / 32), dim3(1024), 0, pcr_s(stream), a);
  PCR_CHECK_LAUNCH();
  return PCR_OK;
}
