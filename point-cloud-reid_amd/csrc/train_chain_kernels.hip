// Fused per-token CHAINS of the attention blocks in training mode (round 5).
//
// The reference runs the tail of every attention block -- merge, LayerNorm, the two feed-forward layers on
// [res ; norm1], LayerNorm, optional residual (models/pointnet2_utils.py:90-114 Self_Attention, :407-437 FP_SA;
// models/attention.py:192-219 corss_attention) -- and its head -- position MLP, residual add, q / k / v projections
// -- as separate autograd nodes; so did this build until round 4: ~128 per-point dense / norm launches of 15-45 us and
// 86 partial-sum reductions per training step, 3.7 of its 10.1 ms.  Every one of those layers maps a
// token to a token, so a 64-token tile can walk the whole chain inside LDS:
//   pcr_attn_tail_fwd_f32   out  = LN2(W2 relu(W0 [res ; LN1(Wm msg)])) [+ res]                          ONE launch
//   pcr_attn_tail_bwd_f32   recomputes that chain for the tile (bit-identical: same code), then walks it backwards:
//                           d msg, d res, and per-workgroup partials of dWm, dW0, dW2, d gamma / d beta of both norms
//                           (accumulated on the matrix core in registers over the workgroup's tiles).       ONE launch
//   pcr_attn_head_{fwd,bwd}_f32   fp = x + P2 relu(P1 xyz + c1) + c2;  out = [Wa x ; Wb fp]   (q|k|v of a self block,
//                           k|v of a cross / FP block) and its backward (dx, partials of dP1, dc1, dP2, dc2, dWa, dWb).
// Nothing but the chain's inputs is kept for the backward.  All matrix phases are f32-input MFMAs through tile_dense2
// (the same packed weight images and fmaf chains as the unfused launches); every reduction is fixed-order (partials +
// pcr_reduce_parts_f32): gradients stay bit-identical from run to run.
#include <type_traits>
#include <utility>

#include "tile_dense.h"
#include "pcr_common.h"

namespace {

// compile-time loop: f(std::integral_constant<int, i>) for i = 0 .. N - 1 (an index that `if constexpr` and a register
// array subscript may both use)
template <class F, int... I>
__device__ __forceinline__ void ck_static_for_impl(F &&f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>()), ...);
}
template <int N, class F>
__device__ __forceinline__ void ck_static_for(F &&f) {
  ck_static_for_impl(static_cast<F &&>(f), std::make_integer_sequence<int, N>());
}

constexpr int kCT = 64, kCRP = 65;      // tokens per tile (TB = 2), LDS row pitch
#ifndef PCR_CHAIN_PF
#define PCR_CHAIN_PF 2
#endif
constexpr int kChainPF = PCR_CHAIN_PF;
// steps of the bf16 dW contraction unrolled together (all four: the wide backward kernels spill 36-167 registers)
#ifndef PCR_DW_UNROLL
#define PCR_DW_UNROLL 2
#endif

template <int CTRL>
__device__ __forceinline__ float ck_dpp(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float ck_quad_sum(float v) {
  v += ck_dpp<0xB1>(v);
  v += ck_dpp<0x4E>(v);
  return v;
}

// dense layer with a compile-time number of 32-row output blocks (wave / tile split and rounds follow from it)
// PREC 0: f32-input MFMAs on a pcr_pack_weight image; 1: split bf16 (three bf16 MFMAs per product, f32 accumulation) on a
// bf16 hi / lo image, the f32 tile converted where it is consumed (tile_dense.h)
template <int PREC, int NCB, class Epi>
__device__ __forceinline__ void cdense(const float *in, int CP, const float *wp, bool sync_epi, Epi epi,
                                       const float *init = nullptr) {
  constexpr int WSEL = NCB >= 3 ? 1 : (NCB == 2 ? 2 : 4);
  constexpr int NR = NCB >= 3 ? (NCB + 3) / 4 : 1;
  if constexpr (PREC == 0) {
    tile_dense2<2, NR, WSEL>(in, CP, wp, NCB * 32, sync_epi, epi, init);
  } else {
    // (weight ring of kChainPF steps instead of tile_dense2p's four: the backward kernels sit at the register limit)
    tile_dense_bf_impl<2, DenseShape<NR, WSEL>::nr, DenseShape<NR, WSEL>::ways, false, 3, Epi, kChainPF>(
        in, CP, wp, NCB * 32, sync_epi, epi, init);
  }
}

// LayerNorm over the C channel rows of buf for each of the 64 token columns.  Thread (token t = tid & 63, part = wave):
// channels part, part + 4, ...  Two passes (mean, centred variance), eps inside the sqrt.
//   XHAT: buf keeps the normalised value (no affine); the affine result goes to `nout` (may be null)
//   else: buf gets the affine result in place
// rstd (64 floats) keeps 1 / sqrt(var + eps) per token when not null.  Ends with a barrier.
template <int C, bool XHAT>
__device__ __forceinline__ void ck_ln_fwd(float *buf, const float *__restrict__ g, const float *__restrict__ bta,
                                          float eps, float *nout, float *rstd, float *red) {
  constexpr int RP = kCRP, T = kCT, NJ = C / 4;
  const int tid = threadIdx.x, t = tid & 63, part = tid >> 6;
  float x[NJ];
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < NJ; j++) {
    x[j] = buf[(part + 4 * j) * RP + t];
    s += x[j];
  }
  red[part * T + t] = s;
  __syncthreads();
  const float mean = ((red[t] + red[T + t]) + (red[2 * T + t] + red[3 * T + t])) * (1.0f / C);
  float v = 0.f;
#pragma unroll
  for (int j = 0; j < NJ; j++) {
    x[j] -= mean;
    v += x[j] * x[j];
  }
  red[(4 + part) * T + t] = v;
  __syncthreads();
  const float var = ((red[4 * T + t] + red[5 * T + t]) + (red[6 * T + t] + red[7 * T + t])) * (1.0f / C);
  const float r = 1.0f / sqrtf(var + eps);
#pragma unroll
  for (int j = 0; j < NJ; j++) {
    const int c = part + 4 * j;
    const float xh = x[j] * r;
    if constexpr (XHAT) {
      buf[c * RP + t] = xh;
      if (nout) nout[c * RP + t] = fmaf(xh, g[c], bta[c]);
    } else {
      buf[c * RP + t] = fmaf(xh, g[c], bta[c]);
    }
  }
  if (rstd && part == 0) rstd[t] = r;
  __syncthreads();
}

// LayerNorm backward per token: dy (rows of `dy`) and the normalised input xh -> dx written to `dx` (may alias dy or xh):
//   a = gamma dy;  dx = rstd (a - mean(a) - xh mean(a xh)).   Ends with a barrier.
template <int C>
__device__ __forceinline__ void ck_ln_bwd(const float *dy, const float *xh, float *dx, const float *__restrict__ g,
                                          const float *rstd, float *red) {
  constexpr int RP = kCRP, T = kCT, NJ = C / 4;
  const int tid = threadIdx.x, t = tid & 63, part = tid >> 6;
  float a[NJ], xv[NJ];
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int j = 0; j < NJ; j++) {
    const int c = part + 4 * j;
    a[j] = g[c] * dy[c * RP + t];
    xv[j] = xh[c * RP + t];
    s1 += a[j];
    s2 = fmaf(a[j], xv[j], s2);
  }
  red[part * T + t] = s1;
  red[(4 + part) * T + t] = s2;
  __syncthreads();
  const float m1 = ((red[t] + red[T + t]) + (red[2 * T + t] + red[3 * T + t])) * (1.0f / C);
  const float m2 = ((red[4 * T + t] + red[5 * T + t]) + (red[6 * T + t] + red[7 * T + t])) * (1.0f / C);
  const float r = rstd[t];
#pragma unroll
  for (int j = 0; j < NJ; j++) dx[(part + 4 * j) * RP + t] = r * (a[j] - m1 - xv[j] * m2);
  __syncthreads();
}

// per-row sums over the tile's 64 tokens, accumulated over the workgroup's tiles: thread (row (tid >> 2) + 64 p, token
// quarter tid & 3); sa += sum X, sb += sum X * Y (Y may be null).  The quarters of a row meet once, at the end (ck_quad_sum).
template <int ROWS>
__device__ __forceinline__ void ck_rowsum(const float *X, const float *Y, float (&sa)[(ROWS + 63) / 64],
                                          float (&sb)[(ROWS + 63) / 64]) {
  constexpr int RP = kCRP;
  const int tid = threadIdx.x, q16 = 16 * (tid & 3);
#pragma unroll
  for (int p = 0; p < (ROWS + 63) / 64; p++) {
    const int r = (tid >> 2) + 64 * p;
    if (r < ROWS) {
      const float *xr = X + r * RP + q16;
      float a0 = 0.f, a1 = 0.f, b0 = 0.f, b1 = 0.f;
      if (Y) {
        const float *yr = Y + r * RP + q16;
#pragma unroll
        for (int t = 0; t < 16; t += 2) {
          a0 += xr[t];
          a1 += xr[t + 1];
          b0 = fmaf(xr[t], yr[t], b0);
          b1 = fmaf(xr[t + 1], yr[t + 1], b1);
        }
      } else {
#pragma unroll
        for (int t = 0; t < 16; t += 2) {
          a0 += xr[t];
          a1 += xr[t + 1];
        }
      }
      sa[p] += a0 + a1;
      sb[p] += b0 + b1;
    }
  }
}

template <int ROWS>
__device__ __forceinline__ void ck_rowsum_store(float *dst_a, float *dst_b, const float (&sa)[(ROWS + 63) / 64],
                                                const float (&sb)[(ROWS + 63) / 64]) {
  const int tid = threadIdx.x;
#pragma unroll
  for (int p = 0; p < (ROWS + 63) / 64; p++) {
    const int r = (tid >> 2) + 64 * p;
    const float a = ck_quad_sum(sa[p]), b = ck_quad_sum(sb[p]);
    if ((tid & 3) == 0 && r < ROWS) {
      if (dst_a) dst_a[r] = a;
      if (dst_b) dst_b[r] = b;
    }
  }
}

// dW tiles (32 x 32 blocks of dY X^T, contraction over the tile's 64 tokens) dealt round-robin to the waves over ALL the
// weight matrices of a chain: global item i belongs to wave i & 3, accumulator i >> 2; a stage owns the items
// [BASE, BASE + NOB * NIB) = (row block of dY, row block of X).
// PREC 1: the contraction over the tile's 64 tokens in four steps of 16 on the bf16 matrix core (A = dY rows, B = X rows,
// token 16 s + 8 h + j in element j of lane half h for both; operands split where they are consumed, as the 128 x 128
// grouped backward does: train_kernels.hip)
template <int PREC, int NTW, int BASE, int NOB, int NIB>
__device__ __forceinline__ void ck_dw_acc(f32x16 (&acc)[NTW], const float *DY, const float *X) {
  constexpr int RP = kCRP, T = kCT;
  const int lane = threadIdx.x & 63, l31 = lane & 31, h = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  ck_static_for<NTW>([&](auto itc) {
    constexpr int it = decltype(itc)::value;
    if constexpr (4 * it + 3 >= BASE && 4 * it < BASE + NOB * NIB) {
      const int item = wave + 4 * it - BASE;
      if (item >= 0 && item < NOB * NIB) {
        const int ob = item / NIB, ib = item - ob * NIB;
        if constexpr (PREC == 0) {
          const float *ap = DY + (ob * 32 + l31) * RP + h;
          const float *bp = X + (ib * 32 + l31) * RP + h;
#pragma unroll 8
          for (int ks = 0; ks < T / 2; ks++)
            acc[it] = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[2 * ks], bp[2 * ks], acc[it], 0, 0, 0);
        } else {
          const float *ap = DY + (ob * 32 + l31) * RP + 8 * h;
          const float *bp = X + (ib * 32 + l31) * RP + 8 * h;
#pragma unroll PCR_DW_UNROLL
          for (int s = 0; s < T / 16; s++) {
            float xa8[8], xb8[8];
#pragma unroll
            for (int j = 0; j < 8; j++) {
              xa8[j] = ap[16 * s + j];
              xb8[j] = bp[16 * s + j];
            }
            bf16x8 ah, al, bh, bl;
            bf_split8(xa8, ah, al, true);
            bf_split8(xb8, bh, bl, true);
            acc[it] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc[it], 0, 0, 0);
            acc[it] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc[it], 0, 0, 0);
            acc[it] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[it], 0, 0, 0);
          }
        }
      }
    }
  });
}

template <int NTW, int BASE, int NOB, int NIB>
__device__ __forceinline__ void ck_dw_store(const f32x16 (&acc)[NTW], float *dw) {   // dw: [NOB * 32][NIB * 32] row-major
  const int lane = threadIdx.x & 63, l31 = lane & 31, h = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  ck_static_for<NTW>([&](auto itc) {
    constexpr int it = decltype(itc)::value;
    if constexpr (4 * it + 3 >= BASE && 4 * it < BASE + NOB * NIB) {
      const int item = wave + 4 * it - BASE;
      if (item >= 0 && item < NOB * NIB) {
        const int ob = item / NIB, ib = item - ob * NIB;
#pragma unroll
        for (int r = 0; r < 16; r++)
          dw[(size_t)(ob * 32 + (r & 3) + 8 * (r >> 2) + 4 * h) * (NIB * 32) + ib * 32 + l31] = acc[it][r];
      }
    }
  });
}

// rows [0, C) x 64 tokens of an LDS tile -> (C, L) tensor of one cloud at token t0; F(c, t) reads the value
template <class F>
__device__ __forceinline__ void ck_store_tile(float *dst, int C, int L, int t0, F f) {
  constexpr int T = kCT;
  const int valid = L - t0 < T ? L - t0 : T;
  if ((L & 3) == 0) {
    const int nq = valid >> 2;
    for (int e = threadIdx.x; e < C * 16; e += kThreads) {
      const int c = e >> 4, q = e & 15;
      if (q < nq) {
        f32x4 v = {f(c, 4 * q), f(c, 4 * q + 1), f(c, 4 * q + 2), f(c, 4 * q + 3)};
        *reinterpret_cast<f32x4 *>(dst + (size_t)c * L + t0 + 4 * q) = v;
      }
    }
  } else {
    for (int e = threadIdx.x; e < C * T; e += kThreads) {
      const int c = e >> 6, t = e & 63;
      if (t < valid) dst[(size_t)c * L + t0 + t] = f(c, t);
    }
  }
}

// Up to three [R_i][64] tiles of (R_i, L) tensors into LDS with ALL of a thread's 16-byte loads in flight before the first
// LDS write (three load_tile calls are three serial HBM round trips for a workgroup that has the CU to itself).  Whole,
// 16-byte aligned tiles only (the caller falls back to load_tile otherwise); a null src skips that tile.
template <int R0, int R1, int R2>
__device__ __forceinline__ void ck_load3(float *d0, const float *s0, float *d1, const float *s1, float *d2, const float *s2,
                                         int L, int t0) {
  constexpr int RP = kCRP, TOT = (R0 + R1 + R2) * 16, NP = (TOT + kThreads - 1) / kThreads;
  f32x4 v[NP];
#pragma unroll
  for (int u = 0; u < NP; u++) {
    const int e = threadIdx.x + u * kThreads;
    const int r = e >> 4, q = e & 15;
    const float *src = r < R0 ? s0 + (size_t)r * L : (r < R0 + R1 ? s1 + (size_t)(r - R0) * L : s2 + (size_t)(r - R0 - R1) * L);
    const bool ok = e < TOT && (r < R0 ? s0 != nullptr : (r < R0 + R1 ? s1 != nullptr : s2 != nullptr));
    v[u] = ok ? *reinterpret_cast<const f32x4 *>(src + t0 + 4 * q) : f32x4{0.f, 0.f, 0.f, 0.f};
  }
#pragma unroll
  for (int u = 0; u < NP; u++) {
    const int e = threadIdx.x + u * kThreads;
    const int r = e >> 4, q = e & 15;
    const bool ok = e < TOT && (r < R0 ? s0 != nullptr : (r < R0 + R1 ? s1 != nullptr : s2 != nullptr));
    if (ok) {
      float *d = (r < R0 ? d0 + r * RP : (r < R0 + R1 ? d1 + (r - R0) * RP : d2 + (r - R0 - R1) * RP)) + 4 * q;
      d[0] = v[u][0];
      d[1] = v[u][1];
      d[2] = v[u][2];
      d[3] = v[u][3];
    }
  }
}
__device__ __forceinline__ bool ck_whole(const float *p0, const float *p1, const float *p2, int L, int t0) {
  return (L & 3) == 0 && t0 + kCT <= L &&
         (((reinterpret_cast<size_t>(p0) | reinterpret_cast<size_t>(p1) | reinterpret_cast<size_t>(p2)) & 15) == 0);
}

// ------------------------------------------------------------------------------------------ attention tail ----
struct TailArgs {
  const float *msg, *res;             // (B,D,L), (B,C1,L)
  const float *wm, *w0, *w2;          // packed images of merge (D,D), mlp[0] (HID, C1+D), mlp[2] (OUT,HID)
  const float *wmT, *w0T, *w2T;       // packed transposes (backward)
  const float *g1, *b1, *g2, *b2;     // LayerNorm affine (D), (OUT)
  float eps;
  float *out;                         // forward: (B,OUT,L)
  const float *dout;                  // backward
  float *dmsg, *dres;                 // (B,D,L), (B,C1,L)
  float *parts;                       // [workgroups][part_stride]
  long part_stride;
  int B, L, tpc, total;               // tiles per cloud, tiles in all
};

template <int D, int C1, int HID, int OUT>
struct TailShape {
  static constexpr int CU = C1 + D, CUP8 = (CU + 7) & ~7, CUP32 = (CU + 31) & ~31;
  static constexpr int NT = (D / 32) * (D / 32) + (HID / 32) * (CUP32 / 32) + (OUT / 32) * (HID / 32);   // dW tiles
  static constexpr int NTW = (NT + 3) / 4;
  static constexpr int B_WM = 0, B_W0 = (D / 32) * (D / 32), B_W2 = B_W0 + (HID / 32) * (CUP32 / 32);
  // partial record of one workgroup (floats)
  static constexpr int O_WM = 0, O_W0 = D * D, O_W2 = O_W0 + HID * CUP32, O_G1 = O_W2 + OUT * HID, O_B1 = O_G1 + D,
                       O_G2 = O_B1 + D, O_B2 = O_G2 + OUT, REC = O_B2 + OUT;
  // backward: d out gets its own buffer (all three input tiles of a tile are then requested at once and msg is not loaded
  // twice) where the 160 KB allow it; else it shares msg's rows
  static constexpr bool SEP = (size_t)(D + OUT + D + CUP32 + HID + OUT) * kCRP * 4 + 10 * kCT * 4 <= (size_t)kMaxDynLds;
  static constexpr int rowsA(bool bwd) { return bwd ? (SEP ? D + OUT : (D > OUT ? D : OUT)) : D; }
  static constexpr size_t lds(bool bwd) {
    return ((size_t)(rowsA(bwd) + D + CUP32 + HID + OUT) * kCRP + 8 * kCT + 2 * kCT) * sizeof(float);
  }
};

template <int D, int C1, int HID, int OUT, bool RESID, bool BWD, int PREC>
__global__ __launch_bounds__(kThreads) void attn_tail_kernel(TailArgs a) {
  using S = TailShape<D, C1, HID, OUT>;
  constexpr int RP = kCRP, T = kCT, CU = S::CU, CUP8 = S::CUP8, CUP32 = S::CUP32, NTW = S::NTW;
  // PREC: bit 0 = the backward's matrix phases (dx, dW), bit 1 = the forward chain (and its recomputation) as split bf16.
  // 1 (the default of training): the forward stays on the unfused launches' own fmaf chains -- every ReLU mask and the
  // values the next layers see are those of the f32 graph -- and only the gradient products run on the bf16 matrix core
  constexpr int PF = (PREC >> 1) & 1, PB = PREC & 1;
  static_assert(D % 32 == 0 && HID % 32 == 0 && OUT % 32 == 0 && (!RESID || OUT == C1), "chain shape");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float *A = smem;                                   // msg (backward without SEP: d out, then msg again); then d msg
  float *F = (BWD && S::SEP) ? A + D * RP : A;       // backward: d out
  float *Bm = A + S::rowsA(BWD) * RP;                // merge output -> normalised (x hat)
  float *Cb = Bm + D * RP;                           // U = [res ; norm1]; backward: dU
  float *Dh = Cb + CUP32 * RP;                       // relu(W0 U); backward: its gradient
  float *E = Dh + HID * RP;                          // W2 f0 -> normalised; backward: its gradient
  float *red = E + OUT * RP;                         // [8][64] LayerNorm partial sums
  float *rstd1 = red + 8 * T, *rstd2 = rstd1 + T;
  const int tid = threadIdx.x, L = a.L;

  f32x16 acc[BWD ? NTW : 1];
  float sg1[(D + 63) / 64], sb1[(D + 63) / 64], sg2[(OUT + 63) / 64], sb2[(OUT + 63) / 64];
  if constexpr (BWD) {
#pragma unroll
    for (int i = 0; i < NTW; i++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[i][r] = 0.f;
#pragma unroll
    for (int p = 0; p < (D + 63) / 64; p++) sg1[p] = sb1[p] = 0.f;
#pragma unroll
    for (int p = 0; p < (OUT + 63) / 64; p++) sg2[p] = sb2[p] = 0.f;
  }
  // rows [CU, CUP32) of U are never written by the forward part: zero once (the backward's dU writes zeros there)
  for (int e = tid; e < (CUP32 - CU) * RP; e += kThreads) Cb[CU * RP + e] = 0.f;

  for (int tile = blockIdx.x; tile < a.total; tile += gridDim.x) {
    const int b = tile / a.tpc, t0 = (tile - b * a.tpc) * T;
    const float *msgb = a.msg + (size_t)b * D * L, *resb = a.res + (size_t)b * C1 * L;
    const float *doutb = BWD ? a.dout + (size_t)b * OUT * L : nullptr;
    __syncthreads();     // the previous tile's stores have read their buffers
    // (the one shape without a buffer of its own for d out is also at the register limit: it keeps the serial loads)
    if ((!BWD || S::SEP) && ck_whole(msgb, resb, (BWD && S::SEP) ? doutb : nullptr, L, t0)) {
      ck_load3<D, C1, (BWD && S::SEP) ? OUT : 0>(A, msgb, Cb, resb, F, doutb, L, t0);
    } else {
      load_tile(A, RP, msgb, D, D, L, t0, T);
      load_tile(Cb, RP, resb, C1, C1, L, t0, T);
      if constexpr (BWD && S::SEP) load_tile(F, RP, doutb, OUT, OUT, L, t0, T);
    }
    __syncthreads();
    cdense<PF, D / 32>(A, D, a.wm, false, [&](float v, int o, int t) { Bm[o * RP + t] = v; });
    __syncthreads();
    ck_ln_fwd<D, true>(Bm, a.g1, a.b1, a.eps, Cb + C1 * RP, rstd1, red);
    cdense<PF, HID / 32>(Cb, CUP8, a.w0, false, [&](float v, int o, int t) { Dh[o * RP + t] = fmaxf(v, 0.f); });
    __syncthreads();
    cdense<PF, OUT / 32>(Dh, HID, a.w2, false, [&](float v, int o, int t) { E[o * RP + t] = v; });
    __syncthreads();
    if constexpr (!BWD) {
      ck_ln_fwd<OUT, false>(E, a.g2, a.b2, a.eps, nullptr, nullptr, red);
      ck_store_tile(a.out + (size_t)b * OUT * L, OUT, L, t0, [&](int c, int t) {
        float v = E[c * RP + t];
        if constexpr (RESID) v += Cb[c * RP + t];
        return v;
      });
    } else {
      ck_ln_fwd<OUT, true>(E, a.g2, a.b2, a.eps, nullptr, rstd2, red);
      // ---- backward ----
      if constexpr (!S::SEP) {
        load_tile(F, RP, doutb, OUT, OUT, L, t0, T);    // (over msg, which is reloaded for dWm below)
        __syncthreads();
      }
      ck_rowsum<OUT>(F, E, sb2, sg2);                   // d beta2 += sum dout, d gamma2 += sum dout * xhat2
      __syncthreads();
      ck_ln_bwd<OUT>(F, E, E, a.g2, rstd2, red);        // E = dG
      ck_dw_acc<PB, NTW, S::B_W2, OUT / 32, HID / 32>(acc, E, Dh);
      cdense<PB, HID / 32>(E, OUT, a.w2T, true, [&](float v, int o, int t) {
        const float f = Dh[o * RP + t];
        Dh[o * RP + t] = f > 0.f ? v : 0.f;
      });                                               // Dh = dF0 (masked)
      __syncthreads();
      ck_dw_acc<PB, NTW, S::B_W0, HID / 32, CUP32 / 32>(acc, Dh, Cb);
      cdense<PB, CUP32 / 32>(Dh, HID, a.w0T, true, [&](float v, int o, int t) { Cb[o * RP + t] = v; });   // Cb = dU
      __syncthreads();
      // d res = dU[0, C1) (+ d out: the residual), then msg comes back into A
      ck_store_tile(a.dres + (size_t)b * C1 * L, C1, L, t0, [&](int c, int t) {
        float v = Cb[c * RP + t];
        if constexpr (RESID) v += F[c * RP + t];
        return v;
      });
      ck_rowsum<D>(Cb + C1 * RP, Bm, sb1, sg1);         // d beta1, d gamma1
      __syncthreads();
      if constexpr (!S::SEP) load_tile(A, RP, msgb, D, D, L, t0, T);
      ck_ln_bwd<D>(Cb + C1 * RP, Bm, Bm, a.g1, rstd1, red);   // Bm = dM (its barrier also covers the msg tile)
      ck_dw_acc<PB, NTW, S::B_WM, D / 32, D / 32>(acc, Bm, A);
      cdense<PB, D / 32>(Bm, D, a.wmT, true, [&](float v, int o, int t) { A[o * RP + t] = v; });
      __syncthreads();
      ck_store_tile(a.dmsg + (size_t)b * D * L, D, L, t0, [&](int c, int t) { return A[c * RP + t]; });
    }
  }
  if constexpr (BWD) {
    float *rec = a.parts + (size_t)blockIdx.x * a.part_stride;
    ck_dw_store<NTW, S::B_WM, D / 32, D / 32>(acc, rec + S::O_WM);
    ck_dw_store<NTW, S::B_W0, HID / 32, CUP32 / 32>(acc, rec + S::O_W0);
    ck_dw_store<NTW, S::B_W2, OUT / 32, HID / 32>(acc, rec + S::O_W2);
    ck_rowsum_store<D>(rec + S::O_B1, rec + S::O_G1, sb1, sg1);
    ck_rowsum_store<OUT>(rec + S::O_B2, rec + S::O_G2, sb2, sg2);
  }
}

// kernel PREC code from the two ABI fields (a forward on the bf16 core implies the backward there too: the recomputation
// must reproduce the forward bit for bit, and the opt-in exists for speed)
int chain_prec(int precision, int fwd_precision) {
  if (fwd_precision == PCR_PREC_BF16X3) return 3;
  return precision == PCR_PREC_BF16X3 ? 1 : 0;
}

int ck_ncu() {
  static const int n = [] {
    hipDeviceProp_t pr;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&pr, dev) != hipSuccess) return 256;
    return pr.multiProcessorCount > 0 ? pr.multiProcessorCount : 256;
  }();
  return n;
}

// the shapes of the reference's blocks: (d_model, c1 = width of the residual / first mlp input, hidden, out)
//   self (32,32,64,32) (64,64,128,64); cross (64,64,128,64); FP (64,64,128,128) (64,32,128,64) (64,3,128,32)
#define PCR_TAIL_SHAPES(X) \
  X(32, 32, 64, 32)        \
  X(64, 64, 128, 64)       \
  X(64, 64, 128, 128)      \
  X(64, 32, 128, 64)       \
  X(64, 3, 128, 32)

struct TailInfo {
  bool ok;
  size_t lds;
  int rec;
};

TailInfo tail_info(int d, int c1, int hid, int out, bool bwd) {
#define X(Dv, C1v, Hv, Ov) \
  if (d == Dv && c1 == C1v && hid == Hv && out == Ov) return {true, TailShape<Dv, C1v, Hv, Ov>::lds(bwd), TailShape<Dv, C1v, Hv, Ov>::REC};
  PCR_TAIL_SHAPES(X)
#undef X
  return {false, 0, 0};
}

int tail_grid(const pcr_attn_tail *p, bool bwd) {
  const TailInfo in = tail_info(p->d, p->c1, p->hid, p->out, bwd);
  if (!in.ok) return 0;
  const long tiles = (long)p->B * ((p->L + kCT - 1) / kCT);
  const long occ = (long)(kMaxDynLds / in.lds) < 1 ? 1 : (long)(kMaxDynLds / in.lds);
  const long cap = (long)ck_ncu() * (occ > 2 ? 2 : occ);
  return (int)(tiles < cap ? tiles : cap);
}

template <int D, int C1, int HID, int OUT>
int tail_launch(const pcr_attn_tail *p, bool bwd, hipStream_t st) {
  TailArgs a;
  a.msg = p->msg; a.res = p->res; a.wm = p->wm; a.w0 = p->w0; a.w2 = p->w2;
  a.wmT = p->wmT; a.w0T = p->w0T; a.w2T = p->w2T;
  a.g1 = p->g1; a.b1 = p->b1; a.g2 = p->g2; a.b2 = p->b2; a.eps = p->eps;
  a.out = p->outp; a.dout = p->dout; a.dmsg = p->dmsg; a.dres = p->dres;
  a.parts = p->parts; a.part_stride = p->part_stride;
  a.B = p->B; a.L = p->L; a.tpc = (p->L + kCT - 1) / kCT; a.total = p->B * a.tpc;
  const int grid = tail_grid(p, bwd);
  const size_t lds = TailShape<D, C1, HID, OUT>::lds(bwd);
  const bool resid = p->residual != 0;
  const int prec = chain_prec(p->precision, p->fwd_precision);
#define PCR_TL1(R, BW, PR)                                                                         \
  do {                                                                                             \
    static bool ok = allow_big_lds(attn_tail_kernel<D, C1, HID, OUT, R, BW, PR>);                  \
    (void)ok;                                                                                      \
    hipLaunchKernelGGL((attn_tail_kernel<D, C1, HID, OUT, R, BW, PR>), dim3(grid), dim3(kThreads), lds, st, a); \
  } while (0)
#define PCR_TL(R, BW)                                \
  do {                                               \
    if (prec == 3) PCR_TL1(R, BW, 3);                \
    else if (prec == 1 && BW) PCR_TL1(R, BW, (BW ? 1 : 0)); \
    else PCR_TL1(R, BW, 0);                          \
  } while (0)
  if constexpr (OUT == C1) {
    if (resid) {
      if (bwd) PCR_TL(true, true);
      else PCR_TL(true, false);
      PCR_CHECK_LAUNCH();
      return PCR_OK;
    }
  }
  if (resid) return PCR_ERR_INVALID;
  if (bwd) PCR_TL(false, true);
  else PCR_TL(false, false);
#undef PCR_TL
#undef PCR_TL1
  PCR_CHECK_LAUNCH();
  return PCR_OK;
}

int tail_dispatch(const pcr_attn_tail *p, bool bwd, hipStream_t st) {
#define X(Dv, C1v, Hv, Ov) \
  if (p->d == Dv && p->c1 == C1v && p->hid == Hv && p->out == Ov) return tail_launch<Dv, C1v, Hv, Ov>(p, bwd, st);
  PCR_TAIL_SHAPES(X)
#undef X
  return PCR_ERR_INVALID;
}

// ------------------------------------------------------------------------------------------ attention head ----
// fp = x + P2 relu(P1 xyz + c1) + c2;  out = [W_0 s_0 ; W_1 s_1 [; W_2 s_2]] with s_j = x or fp (bit j of SRC: fp)
struct HeadArgs {
  const float *x, *xyz;               // (B,C,L), (B,3,L)
  const float *p1, *p2, *c1, *c2;     // packed (HD,3), (C,HD); biases zero-padded to 32
  const float *p2T;                   // packed (HD,C) = P2^T (backward)
  const float *w[3], *wT[3];          // packed (D,C) and (C,D)
  float *out;                         // forward: (B, NP D, L)
  const float *dout;
  float *dx;                          // (B,C,L)
  float *parts;
  long part_stride;
  int B, L, tpc, total;
};

template <int C, int HD, int D, int NP>
struct HeadShape {
  static constexpr int T_P1 = HD / 32, T_P2 = (C / 32) * (HD / 32), T_W = (D / 32) * (C / 32);
  static constexpr int NT = T_P1 + T_P2 + NP * T_W, NTW = (NT + 3) / 4;
  static constexpr int B_P1 = 0, B_P2 = T_P1, B_W = T_P1 + T_P2;
  static constexpr int O_P1 = 0, O_P2 = HD * 32, O_W = O_P2 + C * HD, O_C1 = O_W + NP * D * C, O_C2 = O_C1 + HD,
                       REC = O_C2 + C;
};

template <int C, int HD, int D, int NP, int SRC, bool BWD, int PREC>
__global__ __launch_bounds__(kThreads) void attn_head_kernel(HeadArgs a) {
  using S = HeadShape<C, HD, D, NP>;
  constexpr int RP = kCRP, T = kCT, NTW = S::NTW;
  constexpr int PF = (PREC >> 1) & 1, PB = PREC & 1;      // (as attn_tail_kernel)
  static_assert(C % 32 == 0 && HD % 32 == 0 && D % 32 == 0 && NP >= 1 && NP <= 3, "chain shape");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float *XZ = smem;                    // [32]: xyz in rows 0..2, zeros below
  float *X = XZ + 32 * RP;             // [C] x; backward: d fp, then d x
  float *Hb = X + C * RP;              // [HD] relu(P1 xyz + c1); backward: its gradient
  float *FP = Hb + HD * RP;            // [C] fp
  float *G = FP + C * RP;              // [NP D] the projections (forward) / their gradients (backward)
  const int tid = threadIdx.x, L = a.L;

  f32x16 acc[BWD ? NTW : 1];
  float sc1[(HD + 63) / 64], sc2[(C + 63) / 64], dummy1[(HD + 63) / 64], dummy2[(C + 63) / 64];
  if constexpr (BWD) {
#pragma unroll
    for (int i = 0; i < NTW; i++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[i][r] = 0.f;
#pragma unroll
    for (int p = 0; p < (HD + 63) / 64; p++) sc1[p] = dummy1[p] = 0.f;
#pragma unroll
    for (int p = 0; p < (C + 63) / 64; p++) sc2[p] = dummy2[p] = 0.f;
  }
  for (int e = tid; e < 29 * RP; e += kThreads) XZ[3 * RP + e] = 0.f;

  for (int tile = blockIdx.x; tile < a.total; tile += gridDim.x) {
    const int b = tile / a.tpc, t0 = (tile - b * a.tpc) * T;
    const float *xb = a.x + (size_t)b * C * L, *xzb = a.xyz + (size_t)b * 3 * L;
    const float *doutb = BWD ? a.dout + (size_t)b * NP * D * L : nullptr;
    __syncthreads();
    if (ck_whole(xb, xzb, doutb, L, t0)) {
      ck_load3<3, C, BWD ? NP * D : 0>(XZ, xzb, X, xb, G, doutb, L, t0);
    } else {
      load_tile(XZ, RP, xzb, 3, 3, L, t0, T);
      load_tile(X, RP, xb, C, C, L, t0, T);
      if constexpr (BWD) load_tile(G, RP, doutb, NP * D, NP * D, L, t0, T);
    }
    __syncthreads();
    cdense<0, HD / 32>(XZ, 8, a.p1, false, [&](float v, int o, int t) { Hb[o * RP + t] = fmaxf(v, 0.f); }, a.c1);
    __syncthreads();
    cdense<PF, C / 32>(Hb, HD, a.p2, false, [&](float v, int o, int t) { FP[o * RP + t] = v + X[o * RP + t]; }, a.c2);
    __syncthreads();
    if constexpr (!BWD) {
      ck_static_for<NP>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        const float *src = ((SRC >> j) & 1) ? FP : X;
        cdense<PF, D / 32>(src, C, a.w[j], false, [&](float v, int o, int t) { G[(j * D + o) * RP + t] = v; });
      });
      __syncthreads();
      ck_store_tile(a.out + (size_t)b * NP * D * L, NP * D, L, t0, [&](int c, int t) { return G[c * RP + t]; });
    } else {
      // dW_j += G_j s_j^T (x and fp are both still in place)
      ck_static_for<NP>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        const float *src = ((SRC >> j) & 1) ? FP : X;
        ck_dw_acc<PB, NTW, S::B_W + j * S::T_W, D / 32, C / 32>(acc, G + j * D * RP, src);
      });
      __syncthreads();                  // x is dead from here: its rows take d fp
      // d fp = sum over the fp-sourced projections of W_j^T G_j
      {
        bool first = true;
        ck_static_for<NP>([&](auto jc) {
          constexpr int j = decltype(jc)::value;
          if constexpr ((SRC >> j) & 1) {
            if (first) cdense<PB, C / 32>(G + j * D * RP, D, a.wT[j], false, [&](float v, int o, int t) { X[o * RP + t] = v; });
            else cdense<PB, C / 32>(G + j * D * RP, D, a.wT[j], false, [&](float v, int o, int t) { X[o * RP + t] += v; });
            first = false;
            __syncthreads();
          }
        });
      }
      ck_rowsum<C>(X, nullptr, sc2, dummy2);                  // d c2 += sum d fp
      ck_dw_acc<PB, NTW, S::B_P2, C / 32, HD / 32>(acc, X, Hb);   // d P2 += d fp h^T
      cdense<PB, HD / 32>(X, C, a.p2T, true, [&](float v, int o, int t) {
        const float hv = Hb[o * RP + t];
        Hb[o * RP + t] = hv > 0.f ? v : 0.f;
      });                                                     // Hb = d h (masked)
      __syncthreads();
      ck_rowsum<HD>(Hb, nullptr, sc1, dummy1);                // d c1
      ck_dw_acc<PB, NTW, S::B_P1, HD / 32, 1>(acc, Hb, XZ);       // d P1 += d h xyz^T
      // d x = d fp + sum over the x-sourced projections of W_j^T G_j
      ck_static_for<NP>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        if constexpr (!((SRC >> j) & 1)) {
          cdense<PB, C / 32>(G + j * D * RP, D, a.wT[j], false, [&](float v, int o, int t) { X[o * RP + t] += v; });
          __syncthreads();
        }
      });
      __syncthreads();
      ck_store_tile(a.dx + (size_t)b * C * L, C, L, t0, [&](int c, int t) { return X[c * RP + t]; });
    }
  }
  if constexpr (BWD) {
    float *rec = a.parts + (size_t)blockIdx.x * a.part_stride;
    ck_dw_store<NTW, S::B_P1, HD / 32, 1>(acc, rec + S::O_P1);
    ck_dw_store<NTW, S::B_P2, C / 32, HD / 32>(acc, rec + S::O_P2);
    ck_static_for<NP>([&](auto jc) {
      constexpr int j = decltype(jc)::value;
      ck_dw_store<NTW, S::B_W + j * S::T_W, D / 32, C / 32>(acc, rec + S::O_W + j * D * C);
    });
    ck_rowsum_store<HD>(rec + S::O_C1, nullptr, sc1, dummy1);
    ck_rowsum_store<C>(rec + S::O_C2, nullptr, sc2, dummy2);
  }
}

constexpr size_t head_lds(int C, int HD, int D, int NP) { return (size_t)(32 + 2 * C + HD + NP * D) * kCRP * sizeof(float); }

// (C, HD, D, NP, SRC): self blocks of SA1 / SA2 (q | k | v of fp), k | v of the cross and FP blocks (k of x, v of fp),
// and q | k | v of a cross block in one piece (q, k of x; v of fp)
#define PCR_HEAD_SHAPES(X)   X(32, 32, 32, 3, 7)        X(64, 64, 64, 3, 7)        X(64, 64, 64, 2, 2)        X(128, 64, 64, 2, 2)       X(64, 64, 64, 3, 4)

struct HeadInfo {
  bool ok;
  size_t lds;
  int rec;
};

HeadInfo head_info(int c, int hd, int d, int np, int src) {
#define X(Cv, Hv, Dv, Nv, Sv)   if (c == Cv && hd == Hv && d == Dv && np == Nv && src == Sv) return {true, head_lds(Cv, Hv, Dv, Nv), HeadShape<Cv, Hv, Dv, Nv>::REC};
  PCR_HEAD_SHAPES(X)
#undef X
  return {false, 0, 0};
}

int head_grid(const pcr_attn_head *p) {
  const HeadInfo in = head_info(p->c, p->hd, p->d, p->np, p->src);
  if (!in.ok) return 0;
  const long tiles = (long)p->B * ((p->L + kCT - 1) / kCT);
  const long occ = (long)(kMaxDynLds / in.lds) < 1 ? 1 : (long)(kMaxDynLds / in.lds);
  const long cap = (long)ck_ncu() * (occ > 2 ? 2 : occ);
  return (int)(tiles < cap ? tiles : cap);
}

template <int C, int HD, int D, int NP, int SRC>
int head_launch(const pcr_attn_head *p, bool bwd, hipStream_t st) {
  HeadArgs a;
  a.x = p->x; a.xyz = p->xyz; a.p1 = p->p1; a.p2 = p->p2; a.c1 = p->c1; a.c2 = p->c2; a.p2T = p->p2T;
  for (int j = 0; j < 3; j++) {
    a.w[j] = p->w[j];
    a.wT[j] = p->wT[j];
  }
  a.out = p->outp; a.dout = p->dout; a.dx = p->dx; a.parts = p->parts; a.part_stride = p->part_stride;
  a.B = p->B; a.L = p->L; a.tpc = (p->L + kCT - 1) / kCT; a.total = p->B * a.tpc;
  const int grid = head_grid(p);
  const size_t lds = head_lds(C, HD, D, NP);
#define PCR_HL(BW, PR)                                                                              \
  do {                                                                                             \
    static bool ok = allow_big_lds(attn_head_kernel<C, HD, D, NP, SRC, BW, PR>);                   \
    (void)ok;                                                                                      \
    hipLaunchKernelGGL((attn_head_kernel<C, HD, D, NP, SRC, BW, PR>), dim3(grid), dim3(kThreads), lds, st, a); \
  } while (0)
  const int prec = chain_prec(p->precision, p->fwd_precision);
  if (bwd) {
    if (prec == 3) PCR_HL(true, 3);
    else if (prec == 1) PCR_HL(true, 1);
    else PCR_HL(true, 0);
  } else {
    if (prec == 3) PCR_HL(false, 3);
    else PCR_HL(false, 0);
  }
#undef PCR_HL
  PCR_CHECK_LAUNCH();
  return PCR_OK;
}

int head_dispatch(const pcr_attn_head *p, bool bwd, hipStream_t st) {
#define X(Cv, Hv, Dv, Nv, Sv)   if (p->c == Cv && p->hd == Hv && p->d == Dv && p->np == Nv && p->src == Sv) return head_launch<Cv, Hv, Dv, Nv, Sv>(p, bwd, st);
  PCR_HEAD_SHAPES(X)
#undef X
  return PCR_ERR_INVALID;
}

bool head_common_ok(const pcr_attn_head *p) {
  if (!p || !p->x || !p->xyz || !p->p1 || !p->p2 || !p->c1 || !p->c2 || p->B < 0 || p->L < 1 ||
      !head_info(p->c, p->hd, p->d, p->np, p->src).ok)
    return false;
  for (int j = 0; j < p->np; j++)
    if (!p->w[j]) return false;
  return true;
}

}  // namespace

PCR_EXPORT int pcr_attn_head_ok(int c, int hd, int d, int np, int src) { return head_info(c, hd, d, np, src).ok ? 1 : 0; }

PCR_EXPORT int pcr_attn_head_part_floats(int c, int hd, int d, int np, int src) { return head_info(c, hd, d, np, src).rec; }

PCR_EXPORT int pcr_attn_head_groups(const pcr_attn_head *p) { return p ? head_grid(p) : 0; }

PCR_EXPORT int pcr_attn_head_fwd_f32(const pcr_attn_head *p, pcr_stream_t stream) {
  if (!head_common_ok(p) || !p->outp) return PCR_ERR_INVALID;
  if (p->B == 0) return PCR_OK;
  pcr_note_arith(p->fwd_precision == PCR_PREC_BF16X3 ? PCR_PREC_BF16X3 : PCR_PREC_F32);
  return head_dispatch(p, false, pcr_s(stream));
}

PCR_EXPORT int pcr_attn_head_bwd_f32(const pcr_attn_head *p, pcr_stream_t stream) {
  if (!head_common_ok(p) || !p->p2T || !p->dout || !p->dx || !p->parts ||
      p->part_stride < pcr_attn_head_part_floats(p->c, p->hd, p->d, p->np, p->src))
    return PCR_ERR_INVALID;
  for (int j = 0; j < p->np; j++)
    if (!p->wT[j]) return PCR_ERR_INVALID;
  if (p->B == 0) return PCR_OK;
  pcr_note_arith(chain_prec(p->precision, p->fwd_precision) ? PCR_PREC_BF16X3 : PCR_PREC_F32);
  return head_dispatch(p, true, pcr_s(stream));
}

PCR_EXPORT int pcr_attn_tail_ok(int d, int c1, int hid, int out, int residual) {
  return tail_info(d, c1, hid, out, true).ok && (!residual || out == c1) ? 1 : 0;
}

PCR_EXPORT int pcr_attn_tail_part_floats(int d, int c1, int hid, int out) { return tail_info(d, c1, hid, out, true).rec; }

PCR_EXPORT int pcr_attn_tail_groups(const pcr_attn_tail *p) { return p ? tail_grid(p, true) : 0; }

PCR_EXPORT int pcr_attn_tail_fwd_f32(const pcr_attn_tail *p, pcr_stream_t stream) {
  if (!p || !p->msg || !p->res || !p->wm || !p->w0 || !p->w2 || !p->g1 || !p->b1 || !p->g2 || !p->b2 || !p->outp ||
      p->B < 0 || p->L < 1 || !pcr_attn_tail_ok(p->d, p->c1, p->hid, p->out, p->residual))
    return PCR_ERR_INVALID;
  if (p->B == 0) return PCR_OK;
  pcr_note_arith(p->fwd_precision == PCR_PREC_BF16X3 ? PCR_PREC_BF16X3 : PCR_PREC_F32);
  return tail_dispatch(p, false, pcr_s(stream));
}

PCR_EXPORT int pcr_attn_tail_bwd_f32(const pcr_attn_tail *p, pcr_stream_t stream) {
  if (!p || !p->msg || !p->res || !p->wm || !p->w0 || !p->w2 || !p->wmT || !p->w0T || !p->w2T || !p->g1 || !p->b1 ||
      !p->g2 || !p->b2 || !p->dout || !p->dmsg || !p->dres || !p->parts || p->B < 0 || p->L < 1 ||
      !pcr_attn_tail_ok(p->d, p->c1, p->hid, p->out, p->residual) ||
      p->part_stride < pcr_attn_tail_part_floats(p->d, p->c1, p->hid, p->out))
    return PCR_ERR_INVALID;
  if (p->B == 0) return PCR_OK;
  pcr_note_arith(chain_prec(p->precision, p->fwd_precision) ? PCR_PREC_BF16X3 : PCR_PREC_F32);
  return tail_dispatch(p, true, pcr_s(stream));
}
