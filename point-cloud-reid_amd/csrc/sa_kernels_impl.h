// Grouped set-abstraction MLP kernels (pcr_sa_mlp_f32).  This file is the body of three translation units:
//   sa_kernels.hip      PCR_SA_PREC 0  f32-input MFMA (exact fmaf chains), every shape, and the C-ABI entry points
//   sa_kernels_bf3.hip  PCR_SA_PREC 1  split bf16 (three v_mfma_f32_32x32x16_bf16 per product, ~2e-6 of the f32 result)
//   sa_kernels_bf1.hip  PCR_SA_PREC 2  plain bf16 activations / weights, f32 accumulate (BASELINE config 2 as stated)
// The bf16 units instantiate the explicit-shape kernels only and export pcr_sa2_try_bf3 / _bf1 to the first one.
#pragma once
#ifndef PCR_SA_PREC
#define PCR_SA_PREC 0
#endif
#include <stdio.h>

#include <vector>

#include "tile_dense.h"

// 1: the first weight fragments of layers 2 AND 3 are requested one phase early; 2: layer 3 only (layer 2's
// would stay live across the gather / max-pool phases and push the kernel over 192 VGPRs = one workgroup per CU)
#ifndef PCR_RING
#define PCR_RING 2
#endif
// bf16 forms: bit 0: the first weight steps of layer 2 are requested early (kernel top / after the previous tile's layer 3),
// bit 1: those of layer 3 right after layer 2's k-loop.  Both at once cost a workgroup of residency everywhere (the
// early fragments stay live across the other layer's epilogue); which ONE fits without that depends on the kernel:
// K-row kernel and 64-row ragged tiles: layer 3 (the longer call; measured -8 % on the 128-channel K-row layer, -2 % on
// the 64-row ragged one); 128-row ragged tiles: neither (layer 2 early measured +5 %, layer 3 early costs the second workgroup).
#ifndef PCR_BF_RING_FUSED
#define PCR_BF_RING_FUSED 2
#endif
#ifndef PCR_BF_RING_RAG2
#define PCR_BF_RING_RAG2 2
#endif
#ifndef PCR_BF_RING_RAG4
#define PCR_BF_RING_RAG4 0
#endif

namespace {
constexpr int kPrec = PCR_SA_PREC;
#if PCR_SA_PREC == 0
// ---------------------------------------------------------------- grouped SA MLP ----
struct SaArgs {
  pcr_sa_params p;
  int C0, C0P, RP, TB, CPW, rowsA, rowsB;
};

__global__ __launch_bounds__(kThreads) void sa_mlp_kernel(SaArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const pcr_sa_params &p = a.p;
  float *bufA = smem;
  float *bufB = smem + a.rowsA * a.RP;
  int *sidx = reinterpret_cast<int *>(bufB + a.rowsB * a.RP);
  const int tid = threadIdx.x;
  const size_t b = blockIdx.y;
  const int c0 = blockIdx.x * a.CPW;
  const int nc = (p.S - c0 < a.CPW) ? p.S - c0 : a.CPW;
  const int rows = nc * p.K, ROWS = 32 * a.TB, RP = a.RP;
  const int K = p.K, D = p.D, N = p.N;

  for (int r = tid; r < ROWS; r += kThreads)
    sidx[r] = r < rows ? p.idx[(b * p.S + c0) * K + r] : -1;
  __syncthreads();

  // gather + relative / edge features -> bufA [C0P][RP]
  const float *xyz = p.xyz + b * N * 3;
  const float *feat = D ? p.feat + b * D * N : nullptr;
  const size_t fs_c = p.feat_point_major ? 1 : (size_t)N, fs_n = p.feat_point_major ? (size_t)D : 1;
  for (int e = tid; e < a.C0P * ROWS; e += kThreads) {
    const int ch = e / ROWS, r = e - ch * ROWS;
    float v = 0.f;
    if (ch < a.C0 && r < rows) {
      const int s = c0 + r / K;
      const int ci = p.centre_idx ? p.centre_idx[b * p.S + s] : s;
      const int i = sidx[r];
      if (ch < 3) {
        v = xyz[i * 3 + ch] - xyz[ci * 3 + ch];
      } else if (p.mode == 0) {
        const int f = ch - 3;
        if (f < D) v = feat[fs_c * f + fs_n * ci];
        else v = feat[fs_c * (f - D) + fs_n * i] - feat[fs_c * (f - D) + fs_n * ci];
      } else {
        v = feat[fs_c * (ch - 3) + fs_n * i];
      }
    }
    bufA[ch * RP + r] = v;
  }
  __syncthreads();

  const int c1 = p.c1, c2 = p.c2, c3 = p.c3;
  {
    const float *sc = p.scale[0], *sh = p.shift[0];
    const int lim = ceil8(c1);
    tile_dense(bufA, a.C0P, RP, a.TB, p.wp[0], ceil32(c1), [&](float v, int o, int t) {
      if (o < lim) bufB[o * RP + t] = o < c1 ? fmaxf(v * sc[o] + sh[o], 0.f) : 0.f;
    });
  }
  __syncthreads();
  {
    const float *sc = p.scale[1], *sh = p.shift[1];
    const int lim = ceil8(c2);
    tile_dense(bufB, ceil8(c1), RP, a.TB, p.wp[1], ceil32(c2), [&](float v, int o, int t) {
      if (o < lim) bufA[o * RP + t] = o < c2 ? fmaxf(v * sc[o] + sh[o], 0.f) : 0.f;
    });
  }
  __syncthreads();
  {
    const float *sc = p.scale[2], *sh = p.shift[2];
    tile_dense(bufA, ceil8(c2), RP, a.TB, p.wp[2], ceil32(c3), [&](float v, int o, int t) {
      if (o < c3) bufB[o * RP + t] = fmaxf(v * sc[o] + sh[o], 0.f);
    });
  }
  __syncthreads();
  // max over the K neighbours of each centre
  for (int e = tid; e < c3 * nc; e += kThreads) {
    const int c = e / c3, o = e - c * c3;
    const float *row = bufB + o * RP + c * K;
    float m = row[0];
    for (int k = 1; k < K; k++) m = fmaxf(m, row[k]);
    if (p.out_point_major) p.out[(b * p.S + c0 + c) * c3 + o] = m;
    else p.out[(b * c3 + o) * p.S + c0 + c] = m;
  }
}

#endif   // PCR_SA_PREC == 0 (the generic kernel)

// ------------------------------------------------- grouped SA MLP, second generation ----
// Layer 1 is linear in its input rows [dxyz, f_c, f_i - f_c] (edge) or [dxyz, f_i] (query-and-
// group), so  W1 row = Wa dxyz + P[i] + Q[c]  with the per-POINT tables P = Wf f, Q = (Wc - Wf) f
// computed once per cloud by dense_pm_kernel (K times fewer FLOPs than per (centre,neighbour)
// row).  The kernel gathers P rows (16-byte loads) straight into the layer-1 activation tile,
// then runs layers 2 and 3 on the matrix core IN PLACE in one LDS buffer and reduces max over K.

// ReLU / max on the BIT patterns (signed integer max): exact for every finite input -- a float >= +0 has a
// non-negative pattern that orders like the float, any negative float has a negative pattern -- as long as the
// result is only ever used through max(., 0); one VALU instruction, no NaN-canonicalisation prefix.
__device__ __forceinline__ float relu_bits(float v) {
  const int b = __float_as_int(v);
  return __int_as_float(b > 0 ? b : 0);
}
template <int CTRL>
__device__ __forceinline__ int dpp_i32(int v) {
  return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, true);
}
__device__ __forceinline__ int imax(int a, int b) { return a > b ? a : b; }

struct Sa2Args {
  int B, N, S, K, c1, c2, c3, CPW;
  const float *xyz;
  const int *idx, *centre_idx;
  const float *wa;          // (c1,3) row-major
  const float *pq;          // (B,N,pqw) point-major or null (no features)
  int pqw, qoff;            // row width; offset of Q inside a row, -1 = no Q term
  int dbg;                  // PCR_SA_DBG ablation mask (diagnostics only; 0 in production)
  const float *wp2, *wp3;
  const float *sh1, *sh2, *sh3;   // folded BatchNorm shifts (sh2/sh3 zero-padded to a multiple of 32)
  int out_pm;               // out is (B,S,c3)
  const float *wap;         // packed (c1,3) image of wa, or null
  int l1m;                  // layer 1 on the matrix core (wap given and c1 in layer 2's cout-block class)
  float *out;
  int *claim;               // sa_stream_kernel: eight item counters, 1024 ints apart, zeroed by the launch; null = dealt items
  int xt;                   // sa_stream_kernel: the tables carry layer 1's coordinate term and shift (pcr_sa_params.pq_has_xyz)
};

// NR / NR2: cout-block rounds per wave of layer 3 / layer 2 (2 when the layer has more than 4 x 32 couts)
// RKB > 0 (narrow layers: c1, c2 <= 8 RKB): ALL weight fragments of layers 2 and 3 are fetched into registers at
// the top of the kernel, behind the index staging and the gathers; the two dense calls then run without a single
// weight load (their 4- or 8-block k-loops were mostly L2 latency)
// IMG (bf16 forms with layer 1 on the matrix core): the activations between the layers live in LDS as a bf image
// (tile_dense.h) written by the producing layer's epilogue, already converted, instead of the f32 tile
template <int TB, int NR, int W2, int W3, bool MAXE, int NR2 = NR, int RKB = 0, int PREC = 0, bool IMG = false>
__global__ __launch_bounds__(kThreads) void sa_fused_kernel(Sa2Args a) {
  static_assert(PREC == 0 || RKB == 0, "resident f32 weight fragments belong to the f32 form");
  static_assert(!IMG || (PREC != 0 && MAXE), "bf images belong to the bf16 forms");
  constexpr bool kLoImg = PREC == 1;
  constexpr int ROWS = 32 * TB, RP = ROWS + 1;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int c1 = a.c1, c2 = a.c2, c3 = a.c3, K = a.K;
  constexpr bool kRes = RKB > 0;
  constexpr int RK = kRes ? RKB : 2;
  f32x4 wres2[RK][DenseShape<NR2, W2>::nr], wres3[RK][DenseShape<NR, W3>::nr];
  if constexpr (kRes) {
    tile_dense_ring_load<DenseShape<NR2, W2>::nr, DenseShape<NR2, W2>::ways, RK>(a.wp2, c1, ceil32(c2), wres2);
    tile_dense_ring_load<DenseShape<NR, W3>::nr, DenseShape<NR, W3>::ways, RK>(a.wp3, ceil8(c2), ceil32(c3), wres3);
  }
  // bf16 forms: the first steps of layer 2's weight stream are requested here, behind them the index staging and
  // layer 1; layer 3's right after layer 2's k-loop (the L2 round trips never sit between a barrier and the first MFMA)
  constexpr bool kBfR2 = PREC != 0 && (PCR_BF_RING_FUSED & 1), kBfR3 = PREC != 0 && (PCR_BF_RING_FUSED & 2);
  BfRingOf<NR2, W2> bring2;
  BfRingOf<NR, W3> bring3;
  if constexpr (kBfR2) bf_ring_load2<PREC, NR2, W2>(a.wp2, c1, ceil32(c2), bring2);
  auto hook3 = [&]() {
    if constexpr (kBfR3) bf_ring_load2<PREC, NR, W3>(a.wp3, ceil8(c2), ceil32(c3), bring3);
  };
  // MAXE (K % 16 == 0): the max over K is taken from the layer-3 accumulators (16-lane DPP groups ->
  // gmax[c3][ROWS/16]) and the (c3 x rows) layer-3 output is never materialised in LDS
  int rowsC = c1 > ceil32(c2) ? c1 : ceil32(c2);
  if (!MAXE && ceil32(c3) > rowsC) rowsC = ceil32(c3);
  float *buf = smem;                                        // [rowsC][RP]
  float *sdx = buf + rowsC * RP;                            // [4][ROWS]: dx, dy, dz and a zero row (MFMA k = 3)
  int *sidx = reinterpret_cast<int *>(sdx + 4 * ROWS);      // [ROWS] neighbour, [ROWS] centre point
  int *scen = sidx + ROWS;
  float *sq = reinterpret_cast<float *>(scen + ROWS);     // [CPW][c1] per-centre Q rows + shift
  float *gmax = sdx;   // [c3][2*TB] (MAXE only): reuses the staging arrays, all dead once layer 1 is built
  const int tid = threadIdx.x;
  // XCD-aware mapping: workgroups are dealt round-robin over the 8 XCDs (each with its own L2), so the workgroups of
  // ONE cloud -- which gather rows of the same table -- would pull that table into eight L2s.  Linear id w is remapped
  // so that the ids an XCD receives (w % 8 == x) cover whole consecutive clouds (bijective for any grid size).
  int bsw, xsw;
  {
    const unsigned nwg = gridDim.x * gridDim.y, orig = blockIdx.y * gridDim.x + blockIdx.x;
    const unsigned xcd = orig & 7u, q = nwg >> 3, rr = nwg & 7u;
    const unsigned swz = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + (orig >> 3);
    bsw = (int)(swz / gridDim.x);
    xsw = (int)(swz - (unsigned)bsw * gridDim.x);
  }
  const size_t b = (size_t)bsw;
  const int c0 = xsw * a.CPW;
  const int nc = (a.S - c0 < a.CPW) ? a.S - c0 : a.CPW;
  const int rows = nc * K;
  const float *xyz = a.xyz + b * a.N * 3;

  for (int r = tid; r < ROWS; r += kThreads) {
    int i = -1, ci = -1;
    float dx = 0.f, dy = 0.f, dz = 0.f;
    if (r < rows) {
      const int s = c0 + r / K;
      ci = a.centre_idx ? a.centre_idx[b * a.S + s] : s;
      i = a.idx[(b * a.S + c0) * K + r];
      dx = xyz[i * 3] - xyz[ci * 3];
      dy = xyz[i * 3 + 1] - xyz[ci * 3 + 1];
      dz = xyz[i * 3 + 2] - xyz[ci * 3 + 2];
    }
    sidx[r] = i;
    scen[r] = ci;
    sdx[r] = dx;
    sdx[ROWS + r] = dy;
    sdx[2 * ROWS + r] = dz;
    sdx[3 * ROWS + r] = 0.f;
  }
  __syncthreads();
  if (!(a.dbg & 1)) {
  // layer 1 (VALU + gathers): four output channels per item, rows fastest across lanes; the
  // 16-byte P-row gathers of four items are issued before any of them is consumed; the per-centre
  // Q rows (+ folded BatchNorm shift) and the dxyz weights are staged once in LDS.  The host has
  // folded the BatchNorm scale into wa / P / Q, so the layer is  relu(wa dxyz + P[i] + Q[c]).
  const float *pq = a.pq ? a.pq + b * a.N * (size_t)a.pqw : nullptr;
  const bool has_q = pq && a.qoff >= 0;
  for (int e = tid; e < nc * c1; e += kThreads) {
    const int c = e / c1, o = e - c * c1;
    sq[e] = a.sh1[o] + (has_q ? pq[(size_t)scen[c * K] * a.pqw + a.qoff + o] : 0.f);
  }
  __syncthreads();
  if (IMG || a.l1m) {
    // Layer 1 on the matrix core: relu(Wa dxyz + (shift + Q[c]) + P[i]) per 32 x 32 tile is two MFMAs (k = dx, dy |
    // dz, 0) on accumulators SEEDED with the centre's shift + Q row, plus the table piece in the epilogue -- the fma
    // order of the VALU form below, so the bits are the same.  A lane gathers its token's pieces in the accumulator
    // layout (four runs of four couts), one tile ahead.  Tiles are dealt to the waves like layer 2's.
    constexpr int WAYS = DenseShape<NR2, W2>::ways;
    constexpr int TBW = (TB + WAYS - 1) / WAYS;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, h = lane >> 5;
    const int nCB = ceil32(c1) >> 5;
    const int cb = WAYS == 1 ? wave : (WAYS == 2 ? (wave & 1) : 0);
    const int tb0 = WAYS == 1 ? 0 : (WAYS == 2 ? (wave >> 1) : wave);
    if (cb < nCB) {
      const f32x4 av = reinterpret_cast<const f32x4 *>(a.wap)[(size_t)cb * 64 + l31 * 2 + h];   // k = h, 2+h, (4+h, 6+h)
      f32x4 pc[4], pn[4];
      auto gather = [&](f32x4 (&p)[4], int tb) {
        const int i = sidx[tb * 32 + l31];
        const float *pr = pq + (size_t)(i < 0 ? 0 : i) * a.pqw + cb * 32 + 4 * h;
#pragma unroll
        for (int g = 0; g < 4; g++) p[g] = *reinterpret_cast<const f32x4 *>(pr + 8 * g);
      };
      if (pq && tb0 < TB) gather(pc, tb0);
#pragma unroll
      for (int j = 0; j < TBW; j++) {
        const int tb = tb0 + j * WAYS;
        if (tb < TB) {
          const int t = tb * 32 + l31;
          if (pq && tb + WAYS < TB) gather(pn, tb + WAYS);
          int c = t / K;
          c = c < nc ? c : nc - 1;
          const float *sr = sq + c * c1 + cb * 32 + 4 * h;
          f32x16 acc;
#pragma unroll
          for (int g = 0; g < 4; g++) {
            const f32x4 s4 = *reinterpret_cast<const f32x4 *>(sr + 8 * g);
#pragma unroll
            for (int q = 0; q < 4; q++) acc[4 * g + q] = s4[q];
          }
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[0], sdx[h * ROWS + t], acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[1], sdx[(2 + h) * ROWS + t], acc, 0, 0, 0);
          if constexpr (IMG) {
            if (pq) {
#pragma unroll
              for (int rr = 0; rr < 16; rr++) acc[rr] += pc[rr >> 2][rr & 3];
            }
            bf_store_tile<kLoImg>(buf, ROWS, acc, cb, tb, l31, h);
          } else {
          float *dst = buf + (cb * 32 + 4 * h) * RP + t;
#pragma unroll
          for (int rr = 0; rr < 16; rr++) {
            const float v = pq ? acc[rr] + pc[rr >> 2][rr & 3] : acc[rr];
            dst[((rr & 3) + 8 * (rr >> 2)) * RP] = relu_bits(v);
          }
          }
#pragma unroll
          for (int g = 0; g < 4; g++) pc[g] = pn[g];
        }
      }
    }
  } else {
  const int total = ROWS * (c1 >> 2);
  constexpr int dR = kThreads % ROWS, dO = kThreads / ROWS;
  int r = tid % ROWS, oq = tid / ROWS;
  for (int e0 = tid; e0 < total; e0 += 4 * kThreads) {
    f32x4 p4[4];
    int rr[4], oo[4];
#pragma unroll
    for (int u = 0; u < 4; u++) {
      rr[u] = r;
      oo[u] = oq << 2;
      p4[u] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (pq && e0 + u * kThreads < total && r < rows)
        p4[u] = *reinterpret_cast<const f32x4 *>(pq + (size_t)sidx[r] * a.pqw + oo[u]);
      r += dR;
      oq += dO;
      if (r >= ROWS) { r -= ROWS; oq++; }
    }
#pragma unroll
    for (int u = 0; u < 4; u++) {
      if (e0 + u * kThreads < total) {
        const int rw = rr[u], o = oo[u];
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (rw < rows) {
          const float dx = sdx[rw], dy = sdx[ROWS + rw], dz = sdx[2 * ROWS + rw];
          const float *w = a.wa + o * 3;
          const float *qr = sq + (rw / K) * c1 + o;
#pragma unroll
          for (int j = 0; j < 4; j++) {
            // the order of the matrix-core form of this layer (sa_rag_kernel, W1): an fma chain over dx, dy, dz
            // seeded with shift (+ Q), then the table piece -- both kernels give the same bits
            const float t = fmaf(w[3 * j + 2], dz, fmaf(w[3 * j + 1], dy, fmaf(w[3 * j], dx, qr[j])));
            v[j] = relu_bits(t + p4[u][j]);
          }
        }
#pragma unroll
        for (int j = 0; j < 4; j++) buf[(o + j) * RP + rw] = v[j];
      }
    }
  }
  }
  }
  __syncthreads();
  for (int rep = 0; rep < ((a.dbg & 8) ? 4 : 1); rep++) {   // dbg bit 8: 4x the matrix work (diagnostic)
  if (rep) __syncthreads();
  if (!(a.dbg & 2)) {
    {
      // BatchNorm scale is folded into wp2/wp3 by the host, the shift seeds the accumulators
      // the buffer has ceil32(c2) rows, so every accumulator row is stored unconditionally (rows past c2
      // see zero weights and a zero seed -> relu(0) = 0, which is the zero padding layer 3 wants)
      auto epi2 = [&](float v, int o, int t) { buf[o * RP + t] = relu_bits(v); };
      if constexpr (kRes)
        tile_dense2<TB, NR2, W2, false, decltype(epi2), DenseNoHook, RK, true>(buf, c1, a.wp2, ceil32(c2), true, epi2,
                                                                               a.sh2, wres2);
      else if constexpr (IMG)
        tile_dense2p<PREC, TB, NR2, W2, true, true>(buf, c1, a.wp2, ceil32(c2), true,
                                                    [&](const f32x16 &acc, int cb, int tb, int l31, int h) {
          bf_store_tile<kLoImg>(buf, ROWS, acc, cb, tb, l31, h);
        }, a.sh2, nullptr, hook3, 0, kBfR2 ? &bring2 : nullptr);
      else if constexpr (PREC != 0)
        tile_dense2p<PREC, TB, NR2, W2>(buf, c1, a.wp2, ceil32(c2), true, epi2, a.sh2, nullptr, hook3, 0,
                                        kBfR2 ? &bring2 : nullptr);
      else
        tile_dense2p<PREC, TB, NR2, W2>(buf, c1, a.wp2, ceil32(c2), true, epi2, a.sh2);
    }
    __syncthreads();
    if constexpr (MAXE) {
      constexpr int NG = 2 * TB;
      // (signed maxima of the bit patterns, ReLU at the end: see relu_bits)
      auto epi3 = [&](const f32x16 &acc, int cb, int tb, int l31, int h) {
#pragma unroll
        for (int r = 0; r < 16; r++) {
          const int o = cb * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
          int v = __float_as_int(acc[r]);
          v = imax(v, dpp_i32<0xB1>(v));    // quad_perm [1,0,3,2]  : lane ^ 1
          v = imax(v, dpp_i32<0x4E>(v));    // quad_perm [2,3,0,1]  : lane ^ 2
          v = imax(v, dpp_i32<0x141>(v));   // row_half_mirror      : other quad of the 8-lane half
          v = imax(v, dpp_i32<0x140>(v));   // row_mirror           : other half of the 16-lane row
          if ((l31 & 15) == 0) gmax[o * NG + tb * 2 + (l31 >> 4)] = __int_as_float(imax(v, 0));   // ceil32(c3) rows
        }
      };
      if constexpr (kRes)
        tile_dense2<TB, NR, W3, true, decltype(epi3), DenseNoHook, RK, true>(buf, ceil8(c2), a.wp3, ceil32(c3), false,
                                                                             epi3, a.sh3, wres3);
      else if constexpr (PREC != 0)
        tile_dense2p<PREC, TB, NR, W3, true, IMG>(buf, ceil8(c2), a.wp3, ceil32(c3), false, epi3, a.sh3, nullptr,
                                                  DenseNoHook(), 0, kBfR3 ? &bring3 : nullptr);
      else
        tile_dense2p<PREC, TB, NR, W3, true, IMG>(buf, ceil8(c2), a.wp3, ceil32(c3), false, epi3, a.sh3);
    } else {
      tile_dense2p<PREC, TB, NR, W3>(buf, ceil8(c2), a.wp3, ceil32(c3), true, [&](float v, int o, int t) {
        buf[o * RP + t] = fmaxf(v, 0.f);
      }, a.sh3);
    }
  }
  }
  __syncthreads();
  if (a.dbg & 4) return;
  if constexpr (MAXE) {
    constexpr int NG = 2 * TB;
    const int gpc = K >> 4;  // 16-row groups per centre
    // centre fastest across lanes: the nc outputs of one channel are adjacent in (B,c3,S), so a wave
    // store touches 64/nc lines instead of 64 (the store is still nc*4 bytes per line: see DESIGN.md 4.1)
    for (int e = tid; e < c3 * nc; e += kThreads) {
      const int o = a.out_pm ? e % c3 : e / nc, c = a.out_pm ? e / c3 : e - o * nc;
      const float *g = gmax + o * NG + c * gpc;
      float m = g[0];
      for (int k = 1; k < gpc; k++) m = fmaxf(m, g[k]);
      if (a.out_pm) a.out[(b * a.S + c0 + c) * c3 + o] = m;
      else a.out[(b * c3 + o) * a.S + c0 + c] = m;
    }
  } else {
    for (int e = tid; e < c3 * nc; e += kThreads) {
      const int o = a.out_pm ? e % c3 : e / nc, c = a.out_pm ? e / c3 : e - o * nc;
      const float *row = buf + o * RP + c * K;
      float m = row[0];
      for (int k = 1; k < K; k++) m = fmaxf(m, row[k]);
      if (a.out_pm) a.out[(b * a.S + c0 + c) * c3 + o] = m;
      else a.out[(b * c3 + o) * a.S + c0 + c] = m;
    }
  }
}

// per-wave LDS of the ragged form, in ints: pair maxima [16][C3] | row map [16 K] (indexed form only) | end-of-centre
// flags, a byte per pair of the item
__host__ __device__ constexpr int sas_rag_wave_ints(int c3, int K, bool tab) {
  return 16 * c3 + (tab ? 0 : 16 * K) + ((16 * K / 2 + 15) / 16) * 4;
}

constexpr int kSasClaimInts = 8 * 1024;   // sa_stream_kernel's item counters: one per XCD slot, 4 KB apart
// which K-row launches claim their items (measured at the bench batches, same box: <= 64 channels gain 5-8 % at N = 1024 and
// 8-16 % at N = 4096 -- besides keeping the XCD's waves on one or two clouds, claiming is a dynamic schedule without a tail;
// 128 channels LOSE 1-3 % at every size (their blocks are 2.5x longer, the claim buys nothing and costs its atomics);
// N = 128 loses too (items are short, the claim's latency sits on every one))
inline bool sas_claims(int c1, int c3, int N) {
  static const int min_n_t = pcr_tune_int("PCR_SA_CLAIM_MIN_N"), max_c_t = pcr_tune_int("PCR_SA_CLAIM_MAX_C");   // tuning builds only
  const int max_c = max_c_t > 0 ? max_c_t : 64;
  return N >= (min_n_t > 0 ? min_n_t : 1024) && c1 <= max_c && c3 <= max_c;
}
constexpr int kTraceWgs = 1024, kTraceTiles = 24, kTraceMarks = 8;
__device__ unsigned long long g_rag_trace[kTraceWgs * (2 + kTraceTiles * kTraceMarks)];

#if PCR_SA_PREC != 0
// ---- wave-autonomous K-row kernel: c1 = c2 = c3 = 32 NCB (the Point-Transformer's kNN-grouped SA layers), K % 16 == 0,
// layer 1 on the matrix core.  A wave owns 32 rows (tokens) of the grouped tensor from the index load to the 16-row
// group maxima: layer 1 = two f32 MFMAs on accumulators seeded with the centre's shift + Q row, plus the neighbour's
// table piece gathered in accumulator layout; its relu converts straight into layer 2's B operand (the bf16 weight
// images order a 16-channel step the way an accumulator tile hands its rows to a lane: tile_dense.h), layer 2's
// accumulators into layer 3's, and the max over K starts from layer 3's accumulators (16-lane DPP groups) -- no LDS
// activation tile, no workgroup barrier after the weights are staged.  The arithmetic (seeds, the order of the three
// MFMAs of a product, signed-integer maxima, ReLU last) is sa_fused_kernel's, so the bits are the same.
// Work item = lcm(32, K) rows of one cloud (K = 48: three blocks = two centres): the group maxima of an item meet in a
// wave-private LDS strip, the item's centres leave from there.  Items of a cloud stay on one XCD (its table rows are
// gathered by every item of the cloud: one L2 should hold them).
constexpr int kSasWaves = 8;
constexpr int kSasCpi = 16;   // centres per item of the ragged form

// the three layers of one 32-row block, shared by the K-row and the ragged form.  In: the row's neighbour i, its centre
// point ci; out: y3[NCB3], layer 3 TRANSPOSED (lane (cout, h) holds the tokens 8 g + 4 h + q of its channel).
// s + (lane p of q's DPP row (16 lanes), row_newbcast; p is a constant after unrolling), ONE instruction: the add takes the
// broadcast as its DPP operand.  (Until round 6 an update_dpp builtin + an add: v_mov_b32 0 / v_mov_b32_dpp / half a
// v_pk_add_f32 per element, 160 instructions per 128-channel block for its 64 seeds; same bits.  pt1024's three K-row launches
// 4.145 -> 4.08 ms over six alternations, profiles/r06k_seed_dpp_ab.txt.)  The DPP operand is read two wait states after its
// last VALU write at the earliest: q comes from a memory load, the caller's s_nop covers the case of a copy.
__device__ __forceinline__ float add_row_bcast_f32(float s, float q, int p) {
  float r;
  switch (p & 15) {
#define PCR_RB(P) case P: asm("v_add_f32_dpp %0, %1, %2 row_newbcast:" #P " row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(q), "v"(s)); break;
    PCR_RB(0) PCR_RB(1) PCR_RB(2) PCR_RB(3) PCR_RB(4) PCR_RB(5) PCR_RB(6) PCR_RB(7)
    PCR_RB(8) PCR_RB(9) PCR_RB(10) PCR_RB(11) PCR_RB(12) PCR_RB(13) PCR_RB(14)
    default: asm("v_add_f32_dpp %0, %1, %2 row_newbcast:15 row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(q), "v"(s)); break;
#undef PCR_RB
  }
  return r;
}

template <int NCB, int NCB3, bool LO>
struct SasBlock {
  static constexpr int C = 32 * NCB, NS = 2 * NCB;
  __device__ static __forceinline__ void run(const float *xyz, const float *pq, int pqw, int qoff, bool has_q, int i, int ci,
                                             const float *s_sh1, const float *s_sh2, const float *s_sh3, const f32x4 *s_wa,
                                             const bf16x8 *s_w2, const bf16x8 *s_w3, int lane, f32x16 (&y3)[NCB3],
                                             unsigned long long *tr = nullptr, int *tok = nullptr, bool xt = false) {
    // xt (pcr_sa_params.pq_has_xyz): the tables carry the coordinate term and layer 1's shift -- no coordinates are read
    float dxv = 0.f, dyv = 0.f, dzv = 0.f;
    if (!xt) {
      dxv = xyz[i * 3] - xyz[ci * 3];
      dyv = xyz[i * 3 + 1] - xyz[ci * 3 + 1];
      dzv = xyz[i * 3 + 2] - xyz[ci * 3 + 2];
    }
    run_d(dxv, dyv, dzv, pq, pqw, qoff, has_q, i, ci, s_sh1, s_sh2, s_sh3, s_wa, s_w2, s_w3, lane, y3, tr, tok, xt);
  }
  // the same with the row's point - centre given (the ball query's row table holds it)
  __device__ static __forceinline__ void run_d(float dxv, float dyv, float dzv, const float *pq, int pqw, int qoff, bool has_q,
                                               int i, int ci, const float *s_sh1, const float *s_sh2, const float *s_sh3,
                                               const f32x4 *s_wa, const bf16x8 *s_w2, const bf16x8 *s_w3, int lane,
                                               f32x16 (&y3)[NCB3], unsigned long long *tr = nullptr, int *tok = nullptr,
                                               bool xt = false) {
#ifdef PCR_SA_TRACE_BUILD   // (diagnostic builds: tr = the caller's record of this block, marks 6 / 7 = layer 1 / layer 2 done)
#define PCR_BMARK(m) do { if (tr) tr[m] = __builtin_readcyclecounter(); } while (0)
#else
#define PCR_BMARK(m) do { (void)tr; } while (0)
#endif
    const int j = lane & 31, h = lane >> 5;
    auto cvec = [&](const float *base, int cb, int g) __attribute__((always_inline)) {
      return *reinterpret_cast<const f32x4 *>(base + 32 * cb + 8 * g + 4 * h);
    };
    const float b0 = h ? dyv : dxv, b1 = h ? 0.f : dzv;      // layer 1's B operand: k = h, 2 + h
    bf16x8 bh[NS], bl[NS];
    // the centre's Q row: the 16 lanes of a DPP row (one half h of one 16-row group: K is a multiple of 16) all want the
    // same 4 NCB pieces, so each of them fetches ONE (lane p: cout block p / 4, piece p % 4) and the accumulator seeds take
    // them by row broadcast -- one gather instruction per block instead of 4 NCB (the memory pipe, not the matrix pipe,
    // paced the K-row kernels: trace, tools/trace_stream.py)
    f32x4 qv = {0.f, 0.f, 0.f, 0.f};
    if (has_q) {
      const int p16 = lane & 15;
      if (p16 < 4 * NCB)
        qv = *reinterpret_cast<const f32x4 *>(pq + (size_t)ci * pqw + qoff + (p16 >> 2) * 32 + 8 * (p16 & 3) + 4 * h);
    }
    // ---- layer 1, one cout block at a time (the gathers of a block: 4 sixteen-byte pieces of the neighbour's row).
    // The 128-channel form has the registers (200 of its 256) to request ALL 16 pieces before the first block's MFMAs
    // instead of block by block -- four L2 round trips in a row were 30 % of its block (tools/trace_stream.py)
    // (has_q / pq are launch constants: the three cases are separate straight-line bodies behind wave-uniform branches --
    // as per-element selects they were 72 v_cndmask of a 64-channel block's 730 VALU instructions)
    constexpr bool kAllP = NCB >= 4;
    auto layer1 = [&](auto hq_tag, auto pq_tag) __attribute__((always_inline)) {
      constexpr bool HQ = decltype(hq_tag)::value, PQ = decltype(pq_tag)::value;
      f32x4 ppa[kAllP ? NCB : 1][4];
      if constexpr (kAllP && PQ) {
#pragma unroll
        for (int cb = 0; cb < NCB; cb++) {
          const float *pr = pq + (size_t)i * pqw + cb * 32 + 4 * h;
#pragma unroll
          for (int g = 0; g < 4; g++) ppa[cb][g] = *reinterpret_cast<const f32x4 *>(pr + 8 * g);
        }
#pragma unroll
        for (int cb = 0; cb < NCB; cb++)
#pragma unroll
          for (int g = 0; g < 4; g++) asm volatile("" : "+v"(ppa[cb][g]));   // (the loads land here, not before each use)
      }
      if constexpr (HQ) asm volatile("s_nop 1" : "+v"(qv));   // (add_row_bcast_f32: the DPP operand's wait states)
#pragma unroll
      for (int cb = 0; cb < NCB; cb++) {
        f32x4 pp[4];
        if constexpr (PQ) {
          const float *pr = pq + (size_t)i * pqw + cb * 32 + 4 * h;
#pragma unroll
          for (int g = 0; g < 4; g++) {
            if constexpr (kAllP) pp[g] = ppa[cb][g];
            else pp[g] = *reinterpret_cast<const f32x4 *>(pr + 8 * g);
          }
        }
        f32x16 acc;
#pragma unroll
        for (int g = 0; g < 4; g++) {
          const f32x4 s4 = cvec(s_sh1, cb, g);
#pragma unroll
          for (int q2 = 0; q2 < 4; q2++) {
            if constexpr (HQ) acc[4 * g + q2] = add_row_bcast_f32(s4[q2], qv[q2], 4 * cb + g);   // row_newbcast: lane 4 cb + g of the row
            else acc[4 * g + q2] = s4[q2];
          }
        }
        const f32x4 av = s_wa[cb * 64 + j * 2 + h];
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[0], b0, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[1], b1, acc, 0, 0, 0);
#pragma unroll
        for (int G = 0; G < 2; G++) {
          float v[8];
#pragma unroll
          for (int e = 0; e < 8; e++) {
            const int rr = 8 * G + e;
            float t = acc[rr];
            if constexpr (PQ) t = acc[rr] + pp[rr >> 2][rr & 3];
            v[e] = relu_bits(t);
          }
          bf_split8(v, bh[2 * cb + G], bl[2 * cb + G], LO);
        }
      }
    };
    // xt: a row of layer 1 is relu(P'[i] + Q'[c]) -- the table pieces gathered in accumulator layout plus the centre's row by
    // row broadcast, ONE v_add_f32_dpp per element: no coordinate loads, no seeds, none of the two f32 MFMAs per cout block
    // (512 matrix cycles of a 128-channel block) and no second add.  (Round 6, one process: pt1024's SA2 launch 1.375 -> 1.181 ms,
    // SA3 2.095 -> 1.909, profiles/r06m_xyz_tables_ab.txt; the bench batch's logits 2.9e-5 -> 2.5e-5 from the f32 path.)
    auto layer1_xt = [&]() __attribute__((always_inline)) {
      f32x4 ppa[kAllP ? NCB : 1][4];
      if constexpr (kAllP) {
#pragma unroll
        for (int cb = 0; cb < NCB; cb++) {
          const float *pr = pq + (size_t)i * pqw + cb * 32 + 4 * h;
#pragma unroll
          for (int g = 0; g < 4; g++) ppa[cb][g] = *reinterpret_cast<const f32x4 *>(pr + 8 * g);
        }
#pragma unroll
        for (int cb = 0; cb < NCB; cb++)
#pragma unroll
          for (int g = 0; g < 4; g++) asm volatile("" : "+v"(ppa[cb][g]));
      }
      asm volatile("s_nop 1" : "+v"(qv));   // (add_row_bcast_f32: the DPP operand's wait states)
#pragma unroll
      for (int cb = 0; cb < NCB; cb++) {
        f32x4 pp[4];
        const float *pr = pq + (size_t)i * pqw + cb * 32 + 4 * h;
#pragma unroll
        for (int g = 0; g < 4; g++) {
          if constexpr (kAllP) pp[g] = ppa[cb][g];
          else pp[g] = *reinterpret_cast<const f32x4 *>(pr + 8 * g);
        }
#pragma unroll
        for (int G = 0; G < 2; G++) {
          float v[8];
#pragma unroll
          for (int e = 0; e < 8; e++) {
            const int rr = 8 * G + e;
            v[e] = relu_bits(add_row_bcast_f32(pp[rr >> 2][rr & 3], qv[rr & 3], 4 * cb + (rr >> 2)));
          }
          bf_split8(v, bh[2 * cb + G], bl[2 * cb + G], LO);
        }
      }
    };
    if (xt) {
      layer1_xt();
    } else if (pq) {
      if (has_q) layer1(std::true_type{}, std::true_type{});
      else layer1(std::false_type{}, std::true_type{});
    } else {
      layer1(std::false_type{}, std::false_type{});
    }
    PCR_BMARK(6);
    // (round 6, measured and dropped: taking the token BEFORE layer 1 -- after its gathers have landed -- so that all of a
    // block's f32 VALU work runs inside the holder: pt1024 SA3 2.117 -> 2.200 ms, profiles/r06_inproc_ab.txt)
    mfma_token_acquire(tok, lane);
    PCR_BMARK(3);      // (K-row form: token taken; the ragged form's caller overwrites mark 3 with its own)
    // ---- layer 2 (normal orientation: its accumulators convert into layer 3's operand)
    {
      f32x16 y[NCB];
#pragma unroll
      for (int cb = 0; cb < NCB; cb++)
#pragma unroll
        for (int g = 0; g < 4; g++) {
          const f32x4 s4 = cvec(s_sh2, cb, g);
#pragma unroll
          for (int q2 = 0; q2 < 4; q2++) y[cb][4 * g + q2] = s4[q2];
        }
      const bf16x8 *wb = s_w2 + lane;
      // (round 6, measured and dropped: requesting step s2 + 1's weight units before the MFMAs of step s2 -- hi units
      // double-buffered, +16 registers -- made the three K-row launches of pt1024 0-5 % SLOWER: the other wave's MFMAs
      // already cover a wave's LDS round trips)
#pragma unroll
      for (int s2 = 0; s2 < NS; s2++) {
        bf16x8 wh[NCB], wl[NCB];
#pragma unroll
        for (int cb = 0; cb < NCB; cb++) {
          wh[cb] = wb[((s2 * NCB + cb) * 2) * 64];
          if constexpr (LO) wl[cb] = wb[((s2 * NCB + cb) * 2 + 1) * 64];
        }
        if constexpr (LO) {
#pragma unroll
          for (int cb = 0; cb < NCB; cb++) y[cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh[cb], bl[s2], y[cb], 0, 0, 0);
#pragma unroll
          for (int cb = 0; cb < NCB; cb++) y[cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wl[cb], bh[s2], y[cb], 0, 0, 0);
        }
#pragma unroll
        for (int cb = 0; cb < NCB; cb++) y[cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh[cb], bh[s2], y[cb], 0, 0, 0);
      }
#pragma unroll
      for (int cb = 0; cb < NCB; cb++)
#pragma unroll
        for (int G = 0; G < 2; G++) {
          float v[8];
#pragma unroll
          for (int e = 0; e < 8; e++) v[e] = relu_bits(y[cb][8 * G + e]);
          bf_split8(v, bh[2 * cb + G], bl[2 * cb + G], LO);
        }
    }
    PCR_BMARK(7);
#undef PCR_BMARK
    // ---- layer 3 TRANSPOSED (activations as the A operand, the same weight image as B)
    // (round 6, measured and dropped: these seeds from an LDS table that holds every shift four times over -- four
    // ds_read_b128 that land in the accumulator registers instead of sixteen v_mov per cout block: the three K-row launches of
    // pt1024 4.04 -> 4.19 ms, of pt4096 8.07 -> 8.48, profiles/r06k_seed3_lds_ab.txt.  A v_mov runs under the other wave's
    // MFMAs; the LDS round trip sits in front of this wave's first one.)
#pragma unroll
    for (int cb = 0; cb < NCB3; cb++) {
      const float sv = s_sh3[cb * 32 + j];
#pragma unroll
      for (int rr = 0; rr < 16; rr++) y3[cb][rr] = sv;
    }
    const bf16x8 *wb = s_w3 + lane;
#pragma unroll
    for (int s2 = 0; s2 < NS; s2++) {
      bf16x8 wh[NCB3], wl[NCB3];
#pragma unroll
      for (int cb = 0; cb < NCB3; cb++) {
        wh[cb] = wb[((s2 * NCB3 + cb) * 2) * 64];
        if constexpr (LO) wl[cb] = wb[((s2 * NCB3 + cb) * 2 + 1) * 64];
      }
      if constexpr (LO) {
#pragma unroll
        for (int cb = 0; cb < NCB3; cb++) y3[cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bl[s2], wh[cb], y3[cb], 0, 0, 0);
#pragma unroll
        for (int cb = 0; cb < NCB3; cb++) y3[cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh[s2], wl[cb], y3[cb], 0, 0, 0);
      }
#pragma unroll
      for (int cb = 0; cb < NCB3; cb++) y3[cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh[s2], wh[cb], y3[cb], 0, 0, 0);
    }
    mfma_token_release(tok, lane);
  }
};

// LDS layout shared by both forms: w2 | w3 | sh1 C | sh2 C | sh3 C3 | wa | per-wave area
template <int NCB, int NCB3>
struct SasLds {
  static constexpr int C = 32 * NCB, C3 = 32 * NCB3, NS = 2 * NCB;
  static constexpr int W2U = NS * NCB * 128, W3U = NS * NCB3 * 128;   // 16-byte units
  static constexpr size_t kFixed = (size_t)(W2U + W3U) * 16 + (size_t)(2 * C + C3) * 4 + (size_t)NCB * 64 * 16;
};

template <int NCB, int NCB3>
__device__ __forceinline__ void sas_stage(float *smem, const float *wp2, const float *wp3, const float *sh1, const float *sh2,
                                          const float *sh3, const float *wap, int nthr) {
  using L = SasLds<NCB, NCB3>;
  f32x4 *d2 = reinterpret_cast<f32x4 *>(smem), *d3 = d2 + L::W2U;
  float *s_sh = reinterpret_cast<float *>(d3 + L::W3U);
  f32x4 *s_wa = reinterpret_cast<f32x4 *>(s_sh + 2 * L::C + L::C3);
  const f32x4 *w2 = reinterpret_cast<const f32x4 *>(wp2), *w3 = reinterpret_cast<const f32x4 *>(wp3);
  const int tid = threadIdx.x;
  for (int e = tid; e < L::W2U; e += nthr) d2[e] = w2[e];
  for (int e = tid; e < L::W3U; e += nthr) d3[e] = w3[e];
  for (int e = tid; e < L::C; e += nthr) {
    s_sh[e] = sh1[e];
    s_sh[L::C + e] = sh2[e];
  }
  for (int e = tid; e < L::C3; e += nthr) s_sh[2 * L::C + e] = sh3[e];
  for (int e = tid; e < NCB * 64; e += nthr) s_wa[e] = reinterpret_cast<const f32x4 *>(wap)[e];
  __syncthreads();
}

template <int NCB, int NCB3, bool LO>
__global__ __launch_bounds__(64 * kSasWaves) __attribute__((amdgpu_waves_per_eu((NCB + NCB3 >= 6) ? 2 : 4, (NCB + NCB3 >= 6) ? 2 : 4)))
void sa_stream_kernel(Sa2Args a, int nblk_item, int ncen_item) {
  using L = SasLds<NCB, NCB3>;
  constexpr int C = L::C, C3 = L::C3;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const bf16x8 *s_w2 = reinterpret_cast<const bf16x8 *>(smem);
  const bf16x8 *s_w3 = s_w2 + L::W2U;
  const float *s_sh = reinterpret_cast<const float *>(s_w3 + L::W3U);
  const f32x4 *s_wa = reinterpret_cast<const f32x4 *>(s_sh + 2 * C + C3);
  float *s_gm = reinterpret_cast<float *>(smem) + L::kFixed / 4;   // [waves][6 groups][C3] group maxima of the wave's item
  const int tid = threadIdx.x, lane = tid & 63, j = lane & 31, h = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int *s_tok = reinterpret_cast<int *>(s_gm + kSasWaves * 6 * C3);   // [4 SIMDs] MFMA tokens
  if (tid < 4) s_tok[tid] = 0;
  sas_stage<NCB, NCB3>(smem, a.wp2, a.wp3, a.sh1, a.sh2, a.sh3, a.wap, 64 * kSasWaves);
  // on for the shapes that run two waves per SIMD (128-wide layers; measured -6..-9 % on pt1024's SA3 launch, same bits);
  // with four waves per SIMD (two workgroups, two tokens) the narrow shapes gain 0-1.5 %: off.  (PCR_SA_DBG bit 1024 of a
  // tuning build flips the choice; releasing the token for the relu / split between the two layers was measured too: no gain)
  int *tok = ((NCB + NCB3 >= 6) != ((a.dbg & 1024) != 0)) ? s_tok + pcr_simd_id() : nullptr;
  const int K = a.K, gpc = K >> 4;                      // 16-row groups per centre
  const int nitem = (a.S + ncen_item - 1) / ncen_item;  // items per cloud
  float *gm = s_gm + wave * 6 * C3;
  const bool has_q = a.pq && a.qoff >= 0;
  const bool xt = a.xt != 0 && has_q;
  // XCD-aware item order: workgroup w sits on XCD w % 8; the clouds b % 8 == x belong to XCD x
  const int xcd = blockIdx.x & 7, wrank = (blockIdx.x >> 3) * kSasWaves + wave, wstride = ((gridDim.x + 7 - xcd) >> 3) * kSasWaves;
  const int nq = ((a.B + 7 - xcd) >> 3) * nitem;                 // items of this XCD's clouds (the host keeps B x items < 2^31)
  // the wave's item walk without a division per item: (cloud rank on the XCD, item) advance by fixed steps
  const int step_b = wstride / nitem, step_i = wstride - step_b * nitem;
  int bq = wrank / nitem, item = wrank - bq * nitem;
  // ---- CLAIMED items (round 6; clouds of >= 1024 points, <= 64 channels: sas_claims).  Dealt items
  // (item = rank + n x stride) let a wave that misses fall behind and keep an older cloud's table alive: with 512 waves per
  // XCD the live span grows from half a cloud to three or four, and the 64-channel launch of pt4096 fetches every table
  // line ~4 times (8.2 GB per launch against 1.55 GB algorithmic; `TCC_HIT / TCC_MISS`).  Here the waves of
  // an XCD slot take items from a shared counter instead -- TWO at a time at agent scope (correct whatever XCD a workgroup
  // really runs on: the slot only decides which clouds it shares with whom), counters 4 KB apart, the next claim
  // requested when the current pair starts so that its memory-side latency hides behind two items.
  const bool claimed = a.claim != nullptr;
  int *ctr = claimed ? a.claim + xcd * 1024 : nullptr;
  auto claim2 = [&]() __attribute__((always_inline)) {
    int v = 0;
    if (lane == 0) v = __hip_atomic_fetch_add(ctr, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return v;
  };
  int q_first = wrank, q_end = 0, pend = 0;
  if (claimed) {
    q_first = __builtin_amdgcn_readfirstlane(claim2());
    q_end = q_first + 2;
    pend = claim2();
    bq = q_first / nitem;
    item = q_first - bq * nitem;
  }
  // a block's row indices {neighbour, centre point} are requested one block ahead (the next item's first block during
  // the current item's last): nothing else stands between a wave and its table gathers
  int i_pre = 0, ci_pre = 0;
  auto fetch_rows = [&](int bq2, int it2, int blk2) __attribute__((always_inline)) {
    const size_t b2 = (size_t)bq2 * 8 + xcd;
    const int c02 = it2 * ncen_item;
    const int nc2 = a.S - c02 < ncen_item ? a.S - c02 : ncen_item;
    int r2 = blk2 * 32 + j;
    r2 = r2 < nc2 * K ? r2 : nc2 * K - 1;               // padding rows repeat the last one (a max does not care)
    const int s2 = c02 + r2 / K;
    ci_pre = a.centre_idx ? a.centre_idx[b2 * a.S + s2] : s2;
    i_pre = a.idx[(b2 * a.S + c02) * (size_t)K + r2];
  };
  if (q_first < nq) fetch_rows(bq, item, 0);
#ifdef PCR_SA_TRACE_BUILD   // diagnostic builds only: waves 0 and 5 of a workgroup stamp the shader clock (one record per block)
  const bool tracing = (a.dbg & 256) && lane == 0 && (wave == 0 || wave == 5) && blockIdx.x < kTraceWgs / 2;
  unsigned long long *trace = g_rag_trace + (size_t)(2 * blockIdx.x + (wave ? 1 : 0)) * (2 + kTraceTiles * kTraceMarks);
  int trace_it = -(a.dbg >> 16);       // (PCR_SA_DBG bits 16..: blocks to skip before the kTraceTiles recorded ones)
#define PCR_SMARK(m)                                                                                   \
  do {                                                                                                 \
    if (tracing && trace_it >= 0 && trace_it < kTraceTiles) trace[2 + trace_it * kTraceMarks + (m)] = __builtin_readcyclecounter(); \
  } while (0)
#define PCR_STR() ((tracing && trace_it >= 0 && trace_it < kTraceTiles) ? trace + 2 + trace_it * kTraceMarks : nullptr)
#define PCR_SNEXT() trace_it++
#else
#define PCR_SMARK(m) do { } while (0)
#define PCR_STR() nullptr
#define PCR_SNEXT() do { } while (0)
#endif
  for (int qi = q_first, qn = 0; qi < nq; qi = qn) {
    asm volatile("" ::: "memory");
    PCR_SMARK(4);
    const size_t b = (size_t)bq * 8 + xcd;
    const int c0 = item * ncen_item;
    const int nc = a.S - c0 < ncen_item ? a.S - c0 : ncen_item;
    const int rows = nc * K;
    const float *xyz = a.xyz + b * a.N * 3;
    const float *pq = a.pq ? a.pq + b * a.N * (size_t)a.pqw : nullptr;
    const int bq_cur = bq, item_cur = item;
    if (!claimed) {
      qn = qi + wstride;
      item += step_i;                                   // the next item of this wave
      bq += step_b;
      if (item >= nitem) {
        item -= nitem;
        bq++;
      }
    } else {
      if (qi + 1 < q_end) {
        qn = qi + 1;
      } else {                                          // the pair is used up: take the claim requested when it started
        qn = __builtin_amdgcn_readfirstlane(pend);
        q_end = qn + 2;
        pend = claim2();
      }
      bq = qn / nitem;                                  // (qn >= nq: never used)
      item = qn - bq * nitem;
    }
    for (int blk = 0; blk < nblk_item; blk++) {
      if (blk * 32 >= rows) break;                      // (a partial last item: whole blocks of padding are skipped)
      PCR_SMARK(0);
#ifdef PCR_SA_TRACE_BUILD   // (mark 5 of the K-row form: the constant 100 MHz counter at the block top -> the clock the launch runs at)
      if (tracing && trace_it >= 0 && trace_it < kTraceTiles) trace[2 + trace_it * kTraceMarks + 5] = __builtin_amdgcn_s_memrealtime();
#endif
      const int i = i_pre, ci = ci_pre;
      if (blk + 1 < nblk_item && (blk + 1) * 32 < rows) fetch_rows(bq_cur, item_cur, blk + 1);
      else if (qn < nq) fetch_rows(bq, item, 0);
      f32x16 y[NCB3];
      SasBlock<NCB, NCB3, LO>::run(xyz, pq, a.pqw, a.qoff, has_q, i, ci, s_sh, s_sh + C, s_sh + 2 * C, s_wa, s_w2, s_w3, lane, y,
                                   PCR_STR(), tok, xt);
      PCR_SMARK(1);
      // the maximum over a 16-row group = a maximum over eight of the lane's OWN registers plus one exchange with its
      // partner lane (20 instructions per cout block; the token-per-lane form needs a 4-step DPP reduction of every
      // register: 128)
#pragma unroll
      for (int cb = 0; cb < NCB3; cb++) {
        int m0 = __float_as_int(y[cb][0]), m1 = __float_as_int(y[cb][8]);   // (signed maxima of the bit patterns, ReLU last)
#pragma unroll
        for (int rr = 1; rr < 8; rr++) {
          m0 = imax(m0, __float_as_int(y[cb][rr]));
          m1 = imax(m1, __float_as_int(y[cb][8 + rr]));
        }
        m0 = imax(m0, __shfl_xor(m0, 32, 64));
        m1 = imax(m1, __shfl_xor(m1, 32, 64));
        if (h == 0) {
          gm[(blk * 2) * C3 + cb * 32 + j] = __int_as_float(imax(m0, 0));
          gm[(blk * 2 + 1) * C3 + cb * 32 + j] = __int_as_float(imax(m1, 0));
        }
      }
      PCR_SMARK(2);
      PCR_SNEXT();
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // the item's centres: max over their groups, centre-major so that (B,S,c3) rows leave as whole lines
    for (int e = lane; e < nc * C3; e += 64) {
      const int c = e / C3, o = e - c * C3;
      const float *g = gm + (c * gpc) * C3 + o;
      float m = g[0];
      for (int k = 1; k < gpc; k++) m = fmaxf(m, g[k * C3]);
      if (a.out_pm) a.out[(b * a.S + c0 + c) * (size_t)C3 + o] = m;
      else a.out[(b * C3 + o) * (size_t)a.S + c0 + c] = m;
    }
    __builtin_amdgcn_wave_barrier();
  }
#undef PCR_SMARK
#undef PCR_STR
#undef PCR_SNEXT
}

#endif

// ------------------------------------------------- ragged SA MLP (ball-query duplicates) ----
// A ball-query row holds only cnt genuine neighbours; entries [cnt, K) repeat the first one
// (ball_query_cuda.cu:43-47), and a max over K does not care about repeats.  This variant runs the MLP
// on ceil2(max(cnt,1)) rows per centre instead of K:
//   sa_rag_plan_kernel  one thread per cloud walks the S counts and packs whole centres into tiles of
//                       32*TB rows -> per-cloud tile descriptors + tile count;
//   sa_rag_scan_kernel  one workgroup: exclusive scan of the tile counts over the clouds;
//   sa_rag_flatten_kernel  one wave per cloud copies its descriptors into ONE flat list (cloud, first centre, n);
//   sa_rag_rows_kernel  one wave per tile: the row table {neighbour index, dxyz} of the tile and the
//                       first row of each of its centres (the cnt -> prefix -> idx -> xyz chain of four
//                       dependent global loads, taken off the matrix kernel's critical path);
//   sa_rag_kernel       persistent: CUs x residency workgroups stride over the flat list (a grid of
//                       B x worst-case tiles would be 5/6 empty workgroups, which cost the chip a fifth of
//                       its workgroup slots); each tile is sa_fused_kernel's pipeline with a 4-lane
//                       (quad DPP) max; the row table of the NEXT tile is fetched during the matrix
//                       phases.  Output is bit-identical to the dense kernel's.
// Workspace (ints): nt[B] | total | desc[B][2*maxT] | pad4 | flat[B*maxT][4] | ctab[B*maxT][ROWS/4+4]
//                   | rowtab[B*maxT][ROWS][4]
struct RagArgs {
  int B, N, S, K, c1, c2, c3, maxT;
  const float *xyz;
  const int *idx, *cnt, *centre_idx;
  int *ws;
  const float *wa, *pq;
  const float *wap;         // packed (c1, 3) image of wa (layer 1 on the matrix core), or null
  const float *wap4;        // packed (c1, 4) image of [wa | shift of layer 1] (sa_wsplit_rag_kernel), or null
  const float *rowtab;      // the ball query's row table (pcr_ball_query_rows_f32), or null: sa_stream_rag_kernel<.., true>
  int pqw;
  int dbg;                  // PCR_SA_DBG ablation mask (diagnostics only; 0 in production)
  int out_pm;               // out is (B,S,c3)
  const float *wp2, *wp3, *sh1, *sh2, *sh3;
  float *out;
};

// rows of a centre are padded to a multiple of kRagG (the max over a centre's rows starts with a DPP max inside
// groups of kRagG lanes): 2 instead of 4 costs twice the LDS for the group maxima and saves a quarter of the rows
// of sparsely populated balls (mean 2.9 hits: 4.5 -> 3.4 rows per centre)
constexpr int kRagG = 2;
__host__ __device__ inline int rag_ceil(int x) { return (x + kRagG - 1) & ~(kRagG - 1); }
constexpr int kRagCT = 4;   // ints per tile in ctab: ends mask (2 x 32 bits), group count, pad

__host__ __device__ inline size_t rag_desc_off(int B) { return ((size_t)B + 2) & ~(size_t)1; }   // (8-byte aligned pairs)
__host__ __device__ inline size_t rag_flat_off(int B, int maxT) {
  return (rag_desc_off(B) + (size_t)B * 2 * maxT + 3) & ~(size_t)3;
}
__host__ __device__ inline size_t rag_ctab_off(int B, int maxT) {
  return rag_flat_off(B, maxT) + (size_t)B * maxT * 4;
}
__host__ __device__ inline size_t rag_rowtab_off(int B, int maxT, int rows) {
  (void)rows;
  return rag_ctab_off(B, maxT) + (size_t)B * maxT * kRagCT;
}
__host__ __device__ inline size_t rag_ws_ints(int B, int maxT, int rows) {
  return rag_rowtab_off(B, maxT, rows) + (size_t)B * maxT * rows * 4;
}

// one WAVE per cloud: the counts arrive 64 at a time with one coalesced load, the greedy walk itself runs
// on the scalar unit (v_readlane with a uniform lane index), lane 0 stores the descriptors
__global__ __launch_bounds__(256) void sa_rag_plan_kernel(RagArgs a, int rows_per_tile) {
  const int lane = threadIdx.x & 63;
  const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= a.B) return;
  const int *cnt = a.cnt ? a.cnt + (size_t)b * a.S : nullptr;   // no counts: every row of idx is genuine (kNN groups)
  int *tl = a.ws + rag_desc_off(a.B) + (size_t)b * 2 * a.maxT;
  int t = 0, used = 0, first = 0;
  for (int base = 0; base < a.S; base += 64) {
    int c = (cnt && base + lane < a.S) ? cnt[base + lane] : (cnt ? 1 : a.K);
    c = c < 1 ? 1 : (c > a.K ? a.K : c);
    const int g = rag_ceil(c);
    const int nv = a.S - base < 64 ? a.S - base : 64;
    for (int l = 0; l < nv; l++) {
      const int gl = __builtin_amdgcn_readlane(g, l);
      if (used + gl > rows_per_tile) {
        if (lane == 0) {
          tl[2 * t] = first;
          tl[2 * t + 1] = base + l - first;
        }
        t++;
        first = base + l;
        used = 0;
      }
      used += gl;
    }
  }
  if (lane == 0) {
    tl[2 * t] = first;
    tl[2 * t + 1] = a.S - first;
    a.ws[b] = t + 1;
  }
}

// exclusive scan of the per-cloud tile counts (one workgroup; four clouds per thread and iteration), in place:
// ws[b] <- first flat index of cloud b, ws[B] <- total
__global__ __launch_bounds__(1024) void sa_rag_scan_kernel(int B, int *ws) {
  __shared__ int wtot[16];
  __shared__ int carry;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) carry = 0;
  __syncthreads();
  for (int base = 0; base < B; base += 4096) {
    int v[4];
    int sum = 0;
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const int b = base + tid * 4 + u;
      v[u] = b < B ? ws[b] : 0;
      sum += v[u];
    }
    int incl = sum;                          // wave scans, then one scan over the 16 wave totals: two barriers
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const int t = __shfl_up(incl, off, 64);
      if (lane >= off) incl += t;
    }
    if (lane == 63) wtot[wave] = incl;
    __syncthreads();
    if (wave == 0) {
      int w = lane < 16 ? wtot[lane] : 0;
#pragma unroll
      for (int off = 1; off < 16; off <<= 1) {
        const int t = __shfl_up(w, off, 64);
        if (lane >= off) w += t;
      }
      if (lane < 16) wtot[lane] = w;         // inclusive totals
    }
    __syncthreads();
    int excl = incl - sum + (wave ? wtot[wave - 1] : 0) + carry;
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const int b = base + tid * 4 + u;
      if (b < B) ws[b] = excl;
      excl += v[u];
    }
    __syncthreads();
    if (tid == 1023) carry = excl;
    __syncthreads();
  }
  if (tid == 0) ws[B] = carry;
}

// one wave per cloud: its descriptors (first centre, n centres) -> the flat list entries (cloud, first, n, -)
__global__ __launch_bounds__(256) void sa_rag_flatten_kernel(int B, int maxT, int *ws) {
  const int lane = threadIdx.x & 63;
  const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= B) return;
  const int lo = ws[b], hi = b + 1 < B ? ws[b + 1] : ws[B];
  const int2 *dsrc = reinterpret_cast<const int2 *>(ws + rag_desc_off(B) + (size_t)b * 2 * maxT);
  int4 *flat = reinterpret_cast<int4 *>(ws + rag_flat_off(B, maxT));
  for (int j = lane; j < hi - lo; j += 64) {
    const int2 d = dsrc[j];
    flat[lo + j] = make_int4(b, d.x, d.y, 0);
  }
}

// one wave per tile (4 tiles per workgroup), grid-stride over the flat list
template <int ROWS>
__global__ __launch_bounds__(256) void sa_rag_rows_kernel(RagArgs a) {
  constexpr int MAXC = ROWS / kRagG, CT = kRagCT;   // MAXC <= 64: one lane per centre
  __shared__ int s_off[4][MAXC + 1], s_cnt[4][MAXC];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int total = a.ws[a.B];
  const int4 *flat = reinterpret_cast<const int4 *>(a.ws + rag_flat_off(a.B, a.maxT));
  int *ctab = a.ws + rag_ctab_off(a.B, a.maxT);
  f32x4 *rowtab = reinterpret_cast<f32x4 *>(a.ws + rag_rowtab_off(a.B, a.maxT, ROWS));
  for (int tile = blockIdx.x * 4 + w; tile < total; tile += gridDim.x * 4) {
    const int4 td = flat[tile];
    const size_t b = (size_t)td.x;
    const int first = td.y, nc = td.z;      // 1 <= nc <= MAXC <= 64 lanes
    int n = 1;
    if (lane < nc) {
      n = a.cnt ? a.cnt[b * a.S + first + lane] : a.K;
      n = n < 1 ? 1 : (n > a.K ? a.K : n);
    }
    const int g = lane < nc ? rag_ceil(n) : 0;
    int incl = g;                            // inclusive prefix over the lanes of the wave
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const int t = __shfl_up(incl, off, 64);
      if (lane >= off) incl += t;
    }
    if (lane < nc) {
      s_off[w][lane] = incl - g;
      s_cnt[w][lane] = n;
    }
    const int used_rows = __shfl(incl, nc - 1, 64);
    if (lane == 0) s_off[w][nc] = used_rows;
    {  // bit q of `ends`: row group q is the last group of its centre (what the max-pool epilogue scans)
      const int last = (incl / kRagG) - 1;
      unsigned lo32 = (lane < nc && last < 32) ? 1u << last : 0u;
      unsigned hi32 = (lane < nc && last >= 32) ? 1u << (last - 32) : 0u;
#pragma unroll
      for (int off = 32; off >= 1; off >>= 1) {
        lo32 |= __shfl_xor(lo32, off, 64);
        hi32 |= __shfl_xor(hi32, off, 64);
      }
      if (lane == 0) {
        ctab[(size_t)tile * CT] = (int)lo32;
        ctab[(size_t)tile * CT + 1] = (int)hi32;
        ctab[(size_t)tile * CT + 2] = used_rows / kRagG;
      }
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xc07f);      // lgkmcnt(0): the wave's own LDS writes have landed
    const int used = s_off[w][nc];
    const float *xyz = a.xyz + b * a.N * 3;
    for (int r = lane; r < ROWS; r += 64) {
      f32x4 v = {__int_as_float(-1), 0.f, 0.f, 0.f};
      if (r < used) {
        int lo = 0, hi = nc - 1;             // last centre whose first row is <= r
        while (lo < hi) {
          const int mid = (lo + hi + 1) >> 1;
          if (s_off[w][mid] <= r) lo = mid; else hi = mid - 1;
        }
        const int c = lo, k = r - s_off[w][c];
        const int s = first + c;
        const int ci = a.centre_idx ? a.centre_idx[b * a.S + s] : s;
        const int i = a.idx[(b * a.S + s) * a.K + (k < s_cnt[w][c] ? k : 0)];
        v[0] = __int_as_float(i);
        v[1] = xyz[i * 3] - xyz[ci * 3];
        v[2] = xyz[i * 3 + 1] - xyz[ci * 3 + 1];
        v[3] = xyz[i * 3 + 2] - xyz[ci * 3 + 2];
      }
      rowtab[(size_t)tile * ROWS + r] = v;
    }
    __builtin_amdgcn_wave_barrier();
  }
}


#if PCR_SA_PREC != 0
// Ragged form (ball-query groups with hit counts, mode 1): a centre contributes its first rag_ceil(max(cnt, 1)) rows
// (the rest repeat row 0: a max does not care).  Item = kSasCpi consecutive centres of a cloud: the wave scans their row
// counts, lays the rows out back to back (row -> (centre, k) map in a wave-private LDS strip) and runs them through the
// same block routine, 32 at a time.  Rows of a centre come in pairs (counts are even), and with layer 3 transposed a
// pair is two registers of one lane: their maximum goes to a wave-private LDS strip (one row per pair of the block), and
// the lanes -- now one or two channels each -- walk the block's 16 pairs in row order with a running maximum that
// leaves for the output whenever a centre closes (signed maxima of bit patterns from +0: order-independent, the ReLU
// included) -- no tile plan, no descriptors.
typedef int i32x2 __attribute__((ext_vector_type(2)));
// TAB (round 4): the rows' {neighbour, point - centre} come from the ball query's row table (one 16-byte load per row
// at an address that depends on nothing but the item: it is requested a block ahead, the next item's first block and
// counts during the current item) instead of the cnt -> row map -> idx -> xyz chain of dependent loads, which left the
// matrix pipe idle ~1000 cycles per 32-row block (two waves per SIMD cannot hide it).  Same rows, same values, same bits.
// WAVES: 8, or 12 where the per-wave strips leave room (the table form of every shape at K <= 32): three waves per SIMD.
// A block is ~2.6 k cycles of MFMAs and about as many of everything else (splits, LDS traffic, the reduction over pairs,
// the item prologue); two waves per SIMD left the matrix pipe idle half of the time (trace: tools/trace_stream.py).
template <int NCB, int NCB3, bool LO, bool TAB = false, int WAVES = kSasWaves>
__global__ __launch_bounds__(64 * WAVES) __attribute__((amdgpu_waves_per_eu(WAVES / 4, WAVES / 4)))
void sa_stream_rag_kernel(RagArgs a) {
  using L = SasLds<NCB, NCB3>;
  constexpr int C = L::C, C3 = L::C3, CPI = kSasCpi;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const bf16x8 *s_w2 = reinterpret_cast<const bf16x8 *>(smem);
  const bf16x8 *s_w3 = s_w2 + L::W2U;
  const float *s_sh = reinterpret_cast<const float *>(s_w3 + L::W3U);
  const f32x4 *s_wa = reinterpret_cast<const f32x4 *>(s_sh + 2 * C + C3);
  const int tid = threadIdx.x, lane = tid & 63, j = lane & 31, h = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int K = a.K, maxrows = CPI * K;
  int *s_w = reinterpret_cast<int *>(smem) + L::kFixed / 4 + wave * sas_rag_wave_ints(C3, K, TAB);
  int *pbuf = s_w;                 // [16 pairs][C3] maxima of the current block's row pairs (bit patterns)
  int *rmap = s_w + CPI * C3;      // [maxrows] row -> centre | k << 8 (indexed form)
  unsigned char *eflag = reinterpret_cast<unsigned char *>(rmap + (TAB ? 0 : maxrows));   // [maxrows / 2] pair closes a centre
  constexpr int CPL = C3 >= 128 ? C3 / 64 : 1;   // channels of the lane in the reduction over pairs
  static_assert(CPL <= 2, "wider layers do not fit this kernel's LDS anyway");
  const bool lane_on = lane * CPL < C3;
  sas_stage<NCB, NCB3>(smem, a.wp2, a.wp3, a.sh1, a.sh2, a.sh3, a.wap, 64 * WAVES);
  for (int e = lane; e < (maxrows / 2 + 3) / 4; e += 64) reinterpret_cast<int *>(eflag)[e] = 0;
  const int nitem = (a.S + CPI - 1) / CPI;
  const int xcd = blockIdx.x & 7, wrank = (blockIdx.x >> 3) * WAVES + wave, wstride = ((gridDim.x + 7 - xcd) >> 3) * WAVES;
  const int nq = ((a.B + 7 - xcd) >> 3) * nitem;       // (the host keeps B x items below 2^31)
  // the wave's item walk without a division per item: (cloud rank on the XCD, item) advance by fixed steps
  const int step_b = wstride / nitem, step_i = wstride - step_b * nitem;
  int bq = wrank / nitem, item = wrank - bq * nitem;
  // (TAB) what the next item needs first, requested an item ahead: its centres' counts and its first block of rows
  const f32x4 *const tab = reinterpret_cast<const f32x4 *>(a.rowtab);
  int cn_pre = 0;
  f32x4 e_pre = {0.f, 0.f, 0.f, 0.f};
  auto prefetch_item = [&](bool live, int bq2, int it2) __attribute__((always_inline)) {
    if (live) {
      const size_t b2 = (size_t)bq2 * 8 + xcd;
      const int lc = lane < CPI ? lane : CPI - 1;
      cn_pre = it2 * CPI + lc < a.S ? a.cnt[b2 * a.S + it2 * CPI + lc] : 0;
      e_pre = tab[(b2 * nitem + it2) * (size_t)maxrows + j];
    }
  };
  if constexpr (TAB) prefetch_item(wrank < nq, bq, item);
#ifdef PCR_SA_TRACE_BUILD   // diagnostic builds only: waves 0 and 5 of a workgroup stamp the shader clock (one record per block)
  const bool tracing = (a.dbg & 256) && lane == 0 && (wave == 0 || wave == 5) && blockIdx.x < kTraceWgs / 2;
  unsigned long long *trace = g_rag_trace + (size_t)(2 * blockIdx.x + (wave ? 1 : 0)) * (2 + kTraceTiles * kTraceMarks);
  int trace_it = 0;
#define PCR_SMARK(m)                                                                                   \
  do {                                                                                                 \
    if (tracing && trace_it < kTraceTiles) trace[2 + trace_it * kTraceMarks + (m)] = __builtin_readcyclecounter(); \
  } while (0)
#define PCR_STR() ((tracing && trace_it < kTraceTiles) ? trace + 2 + trace_it * kTraceMarks : nullptr)
#define PCR_SNEXT() trace_it++
#else
#define PCR_SMARK(m) do { } while (0)
#define PCR_STR() nullptr
#define PCR_SNEXT() do { } while (0)
#endif
  for (int qi = wrank; qi < nq; qi += wstride) {
    asm volatile("" ::: "memory");
    PCR_SMARK(4);
    const size_t b = (size_t)bq * 8 + xcd;
    const int c0 = item * CPI;
    const int nc = a.S - c0 < CPI ? a.S - c0 : CPI;
    const float *xyz = a.xyz + b * a.N * 3;
    const float *pq = a.pq ? a.pq + b * a.N * (size_t)a.pqw : nullptr;
    const f32x4 *rt = TAB ? tab + (b * nitem + item) * (size_t)maxrows : nullptr;
    float *const obase = a.out_pm ? a.out + (b * a.S + c0) * (size_t)C3 : a.out + b * C3 * (size_t)a.S + c0;
    f32x4 e_nb = e_pre;
    // rows per centre (lanes 0 .. CPI-1), their prefix sums, the row map
    int n = 0;
    if (lane < nc) {
      const int cn = TAB ? cn_pre : a.cnt[b * a.S + c0 + lane];
      n = rag_ceil(cn > 1 ? cn : 1);
      n = n < K ? n : K;
    }
    // the next item of this wave
    item += step_i;
    bq += step_b;
    if (item >= nitem) {
      item -= nitem;
      bq++;
    }
    if constexpr (TAB) prefetch_item(qi + wstride < nq, bq, item);
    int incl = n;   // inclusive prefix over the 16-lane row (DPP row shifts: lanes >= 16 hold zeros and are not read)
    incl += __builtin_amdgcn_update_dpp(0, incl, 0x111, 0xF, 0xF, true);   // row_shr:1
    incl += __builtin_amdgcn_update_dpp(0, incl, 0x112, 0xF, 0xF, true);   // row_shr:2
    incl += __builtin_amdgcn_update_dpp(0, incl, 0x114, 0xF, 0xF, true);   // row_shr:4
    incl += __builtin_amdgcn_update_dpp(0, incl, 0x118, 0xF, 0xF, true);   // row_shr:8
    const int R = __builtin_amdgcn_readlane(incl, CPI - 1);
    if (lane < nc) eflag[(incl >> 1) - 1] = 1;          // the centre's last pair
    if constexpr (!TAB) {
      const int start = incl - n;
      for (int k = 0; k < n; k++) rmap[start + k] = lane | (k << 8);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const int nb = (R + 31) >> 5;
    int cur[CPL], cc = 0;            // the open centre's running maxima, the number of centres closed so far
#pragma unroll
    for (int e = 0; e < CPL; e++) cur[e] = 0;
    PCR_SMARK(5);
    for (int blk = 0; blk < nb; blk++) {
      f32x16 y[NCB3];
      PCR_SMARK(0);
      // which of the block's pairs close a centre (lane p < 16 looks at pair p and takes the flag down again)
      const int np = (R >> 1) - blk * 16 < 16 ? (R >> 1) - blk * 16 : 16;
      bool is_end = false;
      if (lane < np) {
        is_end = eflag[blk * 16 + lane] != 0;
        if (is_end) eflag[blk * 16 + lane] = 0;
      }
      if constexpr (TAB) {
        const f32x4 e = e_nb;
        if (blk + 1 < nb) e_nb = rt[(blk + 1) * 32 + j];   // (whole blocks are written: rows past R are zero entries)
        SasBlock<NCB, NCB3, LO>::run_d(e[1], e[2], e[3], pq, a.pqw, -1, false, __float_as_int(e[0]), 0, s_sh, s_sh + C,
                                       s_sh + 2 * C, s_wa, s_w2, s_w3, lane, y, PCR_STR());
      } else {
        int r = blk * 32 + j;
        r = r < R ? r : R - 1;
        const int mp = rmap[r];
        const int c = mp & 0xFF, k = mp >> 8;
        const int s = c0 + c;
        const int ci = a.centre_idx ? a.centre_idx[b * a.S + s] : s;
        const int i = a.idx[(b * a.S + s) * (size_t)K + k];
        SasBlock<NCB, NCB3, LO>::run(xyz, pq, a.pqw, -1, false, i, ci, s_sh, s_sh + C, s_sh + 2 * C, s_wa, s_w2, s_w3, lane, y);
      }
      PCR_SMARK(1);
      // pairs of rows (tokens 8 g + 4 h + 2 p, + 1) belong to one centre: pair 4 g + 2 h + p of the block puts its maximum
      // into row `pair` of pbuf with plain stores (the first version sent it to the centre's row with LDS integer atomics:
      // 32 per lane and block, each far slower than a store on an LDS unit that eight waves share)
#pragma unroll
      for (int g = 0; g < 4; g++)
#pragma unroll
        for (int pr = 0; pr < 2; pr++) {
          int *prow = pbuf + (4 * g + 2 * h + pr) * C3 + j;
#pragma unroll
          for (int cb = 0; cb < NCB3; cb++)
            prow[cb * 32] = imax(__float_as_int(y[cb][4 * g + 2 * pr]), __float_as_int(y[cb][4 * g + 2 * pr + 1]));
        }
      const unsigned ends = (unsigned)__ballot(is_end);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      PCR_SMARK(2);
      // the lane's channels over the block's pairs in row order: a running maximum (from +0: the ReLU, signed maxima of
      // the bit patterns) that leaves for the output whenever a centre closes and carries over to the next block otherwise
      int v[16][CPL];
#pragma unroll
      for (int p2 = 0; p2 < 16; p2++) {
        const int *src = pbuf + p2 * C3 + (lane_on ? lane * CPL : 0);
        if constexpr (CPL == 2) {
          const i32x2 t = *reinterpret_cast<const i32x2 *>(src);
          v[p2][0] = t[0];
          v[p2][1] = t[1];
        } else {
          v[p2][0] = src[0];
        }
      }
#pragma unroll
      for (int p2 = 0; p2 < 16; p2++) {
        if (p2 < np) {
#pragma unroll
          for (int e = 0; e < CPL; e++) cur[e] = imax(cur[e], v[p2][e]);
          if ((ends >> p2) & 1u) {
            if (lane_on) {
              if (a.out_pm) {
                float *dst = obase + cc * C3 + lane * CPL;
                if constexpr (CPL == 2) *reinterpret_cast<f32x2 *>(dst) = f32x2{__int_as_float(cur[0]), __int_as_float(cur[1])};
                else dst[0] = __int_as_float(cur[0]);
              } else {
#pragma unroll
                for (int e = 0; e < CPL; e++) obase[(lane * CPL + e) * (size_t)a.S + cc] = __int_as_float(cur[e]);
              }
            }
#pragma unroll
            for (int e = 0; e < CPL; e++) cur[e] = 0;
            cc++;
          }
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();   // (the next block's pair maxima overwrite pbuf)
      PCR_SMARK(3);
      PCR_SNEXT();
    }
  }
#undef PCR_SMARK
#undef PCR_STR
#undef PCR_SNEXT
}
#endif


#if PCR_SA_PREC != 0
// ---- cout-split ragged kernel (round 4): the layers whose weight images do not fit LDS beside anything else (SSG SA2:
// 128 / 128 / 256 = 192 KB of hi + lo images) and whose tile form therefore streamed 192 KB of weights from L2 for every
// 64-row tile (13 GB per ssg1024 launch; the dense phases ran at 0.45 of their MFMA floor).  Here the WEIGHTS never move:
// a workgroup has eight waves, wave w keeps cout block w % NCB of layer 2 and cout block w % NCB3 of layer 3 in its
// registers for the whole (persistent) kernel -- 64 + 64 VGPRs at 128 / 128 / 256 -- and the ACTIVATIONS of a tile of
// 32 RB rows (RB = 8 / NCB row blocks) travel through LDS as bf images (tile_dense.h), whose pieces are exactly the MFMA
// operands: two ds_read_b128 per three MFMAs, nothing else.  The work of a tile:
//   A  layer 1: wave (cb = w % NCB, rb = w / NCB) holds the four 16-byte table pieces of its rows' neighbours in
//      accumulator layout, two f32 MFMAs add Wa dxyz to the shift seeds, ReLU + split -> image X1;
//   B  layer 2: the same (cb, rb): NS steps of {2 LDS reads, 3 MFMAs} against the resident weights -> image X2;
//   C  layer 3 TRANSPOSED (activations as the A operand): wave w, cout block w % NCB3, row blocks w / NCB3 + k G3; lane
//      (cout, h) then holds tokens 8 g + 4 h + q of its channel: a row pair is two registers of one lane, and its
//      maximum goes to row `pair` of obuf with one plain LDS store;
//   D  every centre is reduced over its own pairs (signed maxima of the bit patterns from +0: the ReLU) and leaves as
//      whole 16-byte pieces.
// Eight waves in lockstep would run these phases one after the other, the matrix pipe idle during A, D and both
// epilogues (first version: 1.5 ms per ssg1024 launch, ablations: MFMA phases 0.65 ms, everything else 0.9).  So the
// tiles are SOFTWARE-PIPELINED inside every wave, two barriers per tile:
//   step 1 (i):  B(i)'s MFMA steps, between them the pair maxima of tile i - 1 -> obuf | X2 | barrier
//   step 2 (i):  C(i)'s MFMA steps, between them gather(i + 1), D(i - 1), A(i + 1) -> the other X1 buffer, the row entry
//                of tile i + 2 | barrier
// (one piece of side work per MFMA triple, pinned by scheduling barriers: the vector unit issues while the matrix pipe
// is busy).  Tiles, row tables and segment masks are sa_rag_kernel's (the same pre-kernels); every row goes through the
// same instruction sequence wherever it sits, so ragged and K-row evaluation agree bit for bit.
template <int NCB, int NCB3, bool LO>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2)))
void sa_wsplit_rag_kernel(RagArgs a) {
  constexpr int NS = 2 * NCB, RB = 8 / NCB, ROWS = 32 * RB, NG = ROWS / kRagG, CT = kRagCT;
  constexpr int G3 = 8 / NCB3, NRB3 = RB / G3, C = 32 * NCB, C3 = 32 * NCB3;
  static_assert(NCB * RB == 8 && NCB3 * G3 == 8 && NRB3 * G3 == RB, "eight waves cover every (cout block, row block)");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float *x1 = smem;                                    // 2 x bf image [C / 8 pieces][hi, lo][ROWS] 16-byte units
  float *x2 = x1 + 2 * C * ROWS;                       // (hi + lo of C x ROWS = the bytes of the f32 tile)
  int *obuf = reinterpret_cast<int *>(x2 + C * ROWS);  // [NG row pairs][C3] pair maxima of the tile before (bit patterns)
  float *s_sh2 = reinterpret_cast<float *>(obuf + NG * C3);
  int *coff = reinterpret_cast<int *>(s_sh2 + C);      // [NG + 1] first row pair of every centre of that tile
  const int tid = threadIdx.x, lane = tid & 63, j = lane & 31, h = lane >> 5;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int cb2 = w % NCB, rb2 = w / NCB;              // layers 1 / 2: this wave's tile
  const int cb3 = w % NCB3, rb3 = w / NCB3;            // layer 3: cout block, first row block (then + G3)
  for (int e = tid; e < C; e += 512) s_sh2[e] = a.sh2[e];
  // resident weights (the images' unit of (step s, cout block cb, part) is ((s nCB + cb) 2 + part) 64 + lane)
  bf16x8 w2h[NS], w2l[NS], w3h[NS], w3l[NS];
  {
    const bf16x8 *i2 = reinterpret_cast<const bf16x8 *>(a.wp2) + lane, *i3 = reinterpret_cast<const bf16x8 *>(a.wp3) + lane;
#pragma unroll
    for (int s2 = 0; s2 < NS; s2++) {
      w2h[s2] = i2[((s2 * NCB + cb2) * 2) * 64];
      w3h[s2] = i3[((s2 * NCB3 + cb3) * 2) * 64];
      if constexpr (LO) {
        w2l[s2] = i2[((s2 * NCB + cb2) * 2 + 1) * 64];
        w3l[s2] = i3[((s2 * NCB3 + cb3) * 2 + 1) * 64];
      }
    }
  }
  // layer 1's A operand, k = h and 2 + h of [wa | shift]: the B operand is (dx, dy, dz, 1), so the shift arrives with
  // the coordinate term (no seed reads; the accumulator starts from the gathered table pieces)
  const f32x4 av = reinterpret_cast<const f32x4 *>(a.wap4)[cb2 * 64 + j * 2 + h];
  const float sv3 = a.sh3[cb3 * 32 + j];
  // XCD-aware tile ranges, as sa_rag_kernel: the tiles of a cloud gather rows of one table -> one L2
  const int n_all = a.ws[a.B];
  const bool xaware = gridDim.x >= 8;
  const int xcd = blockIdx.x & 7;
  const int t_step = xaware ? ((int)gridDim.x + 7 - xcd) >> 3 : (int)gridDim.x;
  const int t_lo = xaware ? (int)((long long)n_all * xcd / 8) : 0;
  const int total = xaware ? (int)((long long)n_all * (xcd + 1) / 8) : n_all;
  const int t_first = xaware ? t_lo + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
  const int4 *flat = reinterpret_cast<const int4 *>(a.ws + rag_flat_off(a.B, a.maxT));
  const int *ctab = a.ws + rag_ctab_off(a.B, a.maxT);
  const f32x4 *rowtab = reinterpret_cast<const f32x4 *>(a.ws + rag_rowtab_off(a.B, a.maxT, ROWS));
  const bf16x8 *x2u = reinterpret_cast<const bf16x8 *>(x2);
  const int r1 = rb2 * 32 + j;                         // this lane's row in layers 1 / 2
  f32x4 rv = {__int_as_float(-1), 0.f, 0.f, 0.f};
  // layer 1's accumulator doubles as the gather destination: the table pieces land in it, the shift seeds are added
  // when they have, the two coordinate MFMAs accumulate on top ((P + shift) + Wa dxyz)
  f32x16 acc1;
#pragma unroll
  for (int rr = 0; rr < 16; rr++) acc1[rr] = 0.f;
  // buffer addressing (scalar base + one 32-bit lane offset): 64-bit lane addresses of the row table, the feature
  // table and the output were hoisted out of the tile loop as invariants and spilled
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  const __amdgpu_buffer_rsrc_t rrow = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<f32x4 *>(rowtab), 0, (int)((size_t)n_all * ROWS * 16 > 0x7FFFFFFFull ? 0x7FFFFFFF : (size_t)n_all * ROWS * 16), 0x00020000);
  const int vo_row = r1 * 16, vo_pq = (cb2 * 32 + 4 * h) * 4;
  auto fetch_row = [&](int tile) {
    if (tile < total)
      rv = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rrow, vo_row, tile * (ROWS * 16), 0));
  };
  auto gather = [&](int tile, int cloud) {             // table pieces of the row in rv
    if (tile < total && !a.pq) {
#pragma unroll
      for (int rr = 0; rr < 16; rr++) acc1[rr] = 0.f;
    }
    if (tile < total && a.pq) {
      const __amdgpu_buffer_rsrc_t rpq = __builtin_amdgcn_make_buffer_rsrc(
          const_cast<float *>(a.pq) + (size_t)cloud * a.N * a.pqw, 0, a.N * a.pqw * 4, 0x00020000);
      const int i = __float_as_int(rv[0]);
      const int vo = (i < 0 ? 0 : i) * (a.pqw * 4) + vo_pq;
#pragma unroll
      for (int g = 0; g < 4; g++) {
        const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rpq, vo + 32 * g, 0, 0));
#pragma unroll
        for (int q = 0; q < 4; q++) acc1[4 * g + q] = v[q];
      }
    }
  };
  // layer 1 of this wave's (cb2, rb2) tile, in pieces (one per MFMA triple of the phase it hides under)
  auto a_mfma = [&]() {
    const float b0 = h ? rv[2] : rv[1], b1 = h ? 1.f : rv[3];
    acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[0], b0, acc1, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[1], b1, acc1, 0, 0, 0);
  };
  // a quarter of bf_store_tile: accumulator rows 8 gp + 4 hf .. + 3 = elements 4 hf .. + 3 of piece 4 cb2 + 2 gp + h
  // (8 bytes of the piece's hi unit, 8 of its lo unit)
  auto a_store = [&](float *img, int gp, int hf) {
    typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
    bf16x4 *u = reinterpret_cast<bf16x4 *>(img);
    bf16x4 hi, lo;
#pragma unroll
    for (int q = 0; q < 2; q++) {
      const f32x2 v = {relu_bits(acc1[8 * gp + 4 * hf + 2 * q]), relu_bits(acc1[8 * gp + 4 * hf + 2 * q + 1])};
      const bf16x2 h2 = __builtin_convertvector(v, bf16x2);
      hi[2 * q] = h2[0];
      hi[2 * q + 1] = h2[1];
      if constexpr (LO) {
        const bf16x2 l2 = __builtin_convertvector(bf_residual2(v, h2), bf16x2);
        lo[2 * q] = l2[0];
        lo[2 * q + 1] = l2[1];
      }
    }
    const int P = 4 * cb2 + 2 * gp + h;
    u[2 * ((2 * P) * ROWS + r1) + hf] = hi;
    if constexpr (LO) u[2 * ((2 * P + 1) * ROWS + r1) + hf] = lo;
  };
  // pair maxima of layer 3 (tile i) -> obuf; pm[k][4 g + 2 h' ..]: see phase C
  int pm[NRB3][8];
  // Row pair q = 16 rb + 4 g + 2 h + pr holds tokens 8 g + 4 h + 2 pr (+ 1) of row block rb.  Its maximum goes to
  // obuf[q][cout] with ONE plain ds_write_b32 (the lane's address is fixed, q's compile-time part is the instruction's
  // offset field) -- the first version sent it to the centre's row with an LDS integer max: ~45 cycles of wave time per
  // instruction, sixteen per tile.  The centres' pair ranges come from the tile's `ends` mask (bit q: pair q closes a
  // centre): the lane of wave 0 whose bit is set knows the centre it closes (the set bits below it) and writes that
  // centre's end = the next centre's start into coff; phase D then reduces every centre over its own pairs.
  auto c_epi = [&](int k, int g, int *obase) {
#pragma unroll
    for (int pr = 0; pr < 2; pr++) obase[(16 * k * G3 + 4 * g + pr) * C3] = pm[k][2 * g + pr];
  };
  auto c_offsets = [&](unsigned ends_lo, unsigned ends_hi) {   // wave 0
    if (w == 0) {
      const unsigned bits = NG <= 32 ? ends_lo : (lane < 32 ? ends_lo : ends_hi);
      const int below = NG <= 32 || lane < 32 ? __popc(ends_lo & ((1u << (lane & 31)) - 1u))
                                              : __popc(ends_lo) + __popc(ends_hi & ((1u << (lane & 31)) - 1u));
      if (lane < NG && ((bits >> (lane & 31)) & 1u)) coff[below + 1] = lane + 1;
      if (lane == 0) coff[0] = 0;
    }
  };
  auto d_out = [&](int cloud, int first, int nc) {     // the tile's centres: max over their pairs, then the ReLU (>= +0)
    typedef int i32x4 __attribute__((ext_vector_type(4)));
    const i32x4 *ob4 = reinterpret_cast<const i32x4 *>(obuf);
    const __amdgpu_buffer_rsrc_t rout = __builtin_amdgcn_make_buffer_rsrc(
        a.out_pm ? a.out + ((size_t)cloud * a.S + first) * (size_t)C3 : a.out + (size_t)cloud * C3 * a.S, 0,
        a.out_pm ? nc * C3 * 4 : C3 * a.S * 4, 0x00020000);
    for (int e = tid; e < nc * (C3 / 4); e += 512) {
      const int c = e / (C3 / 4), o4 = e - c * (C3 / 4);
      const int qa = coff[c], qb = coff[c + 1];
      i32x4 m = {0, 0, 0, 0};
      for (int q = qa; q < qb; q += 4) {               // four pairs per round trip (a clamped re-read leaves a max alone)
        i32x4 v[4];
#pragma unroll
        for (int t = 0; t < 4; t++) v[t] = ob4[(q + t < qb ? q + t : qb - 1) * (C3 / 4) + o4];
#pragma unroll
        for (int t = 0; t < 4; t++)
#pragma unroll
          for (int u = 0; u < 4; u++) m[u] = imax(m[u], v[t][u]);
      }
      if (a.out_pm) {
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, m), rout, e * 16, 0, 0);
      } else {
#pragma unroll
        for (int u = 0; u < 4; u++)
          __builtin_amdgcn_raw_buffer_store_b32((unsigned)m[u], rout, ((4 * o4 + u) * a.S + first + c) * 4, 0, 0);
      }
    }
  };
#define PCR_SB() __builtin_amdgcn_sched_barrier(0)
#ifdef PCR_SA_TRACE_BUILD   // diagnostic builds only: wave 0 and wave 7 stamp the shader clock at the step boundaries
  const bool tracing = (a.dbg & 256) && lane == 0 && (w == 0 || w == 7) && blockIdx.x < kTraceWgs / 2;
  unsigned long long *trace = g_rag_trace + (size_t)(2 * blockIdx.x + (w ? 1 : 0)) * (2 + kTraceTiles * kTraceMarks);
  int trace_it = 0;
#define PCR_WMARK(m)                                                                                   \
  do {                                                                                                 \
    if (tracing && trace_it < kTraceTiles) trace[2 + trace_it * kTraceMarks + (m)] = __builtin_readcyclecounter(); \
  } while (0)
#else
#define PCR_WMARK(m) do { } while (0)
#endif
  // ---- prologue: tile 0's layer 1, tile 1's row entry
  int4 td_prev = make_int4(0, 0, 0, 0), td_cur = make_int4(0, 0, 0, 0), td_next = make_int4(0, 0, 0, 0);
  unsigned pe_lo = 0u, pe_hi = 0u;                     // previous tile's segment mask / group count (its pair maxima)
  int p_ngr = 0;
  bool have_prev = false;
  if (t_first < total) {
    td_cur = flat[t_first];
    fetch_row(t_first);
    gather(t_first, td_cur.x);
    a_mfma();
    a_store(x1, 0, 0);
    a_store(x1, 0, 1);
    a_store(x1, 1, 0);
    a_store(x1, 1, 1);
    if (t_first + t_step < total) td_next = flat[t_first + t_step];
    fetch_row(t_first + t_step);
  }
  __syncthreads();
  int par = 0;
  for (int tile = t_first; tile < total; tile += t_step, par ^= 1) {
    const bool have_next = tile + t_step < total;
    const unsigned e_lo = (unsigned)ctab[(size_t)tile * CT], e_hi = (unsigned)ctab[(size_t)tile * CT + 1];
    const int ngr = ctab[(size_t)tile * CT + 2];       // row groups (pairs) in use
    // the descriptor of the tile after next.  Its fields are wave-uniform and the compiler turns the load into
    // `global_load ; s_waitcnt vmcnt(0) ; v_readfirstlane` on the spot -- an exposed L2 round trip at the top of EVERY tile
    // for values that are needed two tiles later.  They stay in vector registers (the empty asm at the bottom of the loop
    // is their first use) and move to scalars there, a whole tile after the request.  (Measured: no change of the launch
    // time -- the other wave of the SIMD covered that wait; kept because it costs nothing.)
    // (loaded unconditionally from a clamped index: behind an `if` the merge with the zero default is a register copy
    // right after the load, with the same wait)
    const int nn_i = tile + 2 * t_step < total ? tile + 2 * t_step : total - 1;
    const int4 t4 = flat[nn_i];
    int nn_x = t4.x, nn_y = t4.y, nn_z = t4.z;
    const bf16x8 *x1u = reinterpret_cast<const bf16x8 *>(x1 + par * C * ROWS);
    float *x1n = x1 + (par ^ 1) * C * ROWS;
    // ---- step 1: layer 2 of (cb2, rb2); between its MFMA triples the pair maxima of the previous tile
    PCR_WMARK(0);
    int *obase = obuf + (16 * rb3 + 2 * h) * C3 + cb3 * 32 + j;
    gather(tile + t_step, td_next.x);                  // (its row entry was requested most of a tile ago; the pieces land
                                                       //  in layer 1's accumulator during this step)
    if (have_prev) c_offsets(pe_lo, pe_hi);
    {
      f32x16 y;
#pragma unroll
      for (int g = 0; g < 4; g++) {
        const f32x4 s4 = *reinterpret_cast<const f32x4 *>(s_sh2 + 32 * cb2 + 8 * g + 4 * h);
#pragma unroll
        for (int q = 0; q < 4; q++) y[4 * g + q] = s4[q];
      }
      const bf16x8 *xb = x1u + 2 * h * ROWS + r1;      // piece 2 s + h of step s: unit (2 (2 s + h) + part) ROWS + row
      constexpr int NE = NRB3 * 4;                      // pieces of the previous tile's epilogue
      // per step: a piece of side work, the NEXT step's operands (they have this step's MFMAs to land), the MFMAs
      bf16x8 xh = xb[0], xl;
      if constexpr (LO) xl = xb[ROWS];
#pragma unroll
      for (int s2 = 0; s2 < NS; s2++) {
        if (have_prev && !(a.dbg & 1)) {     // (a.dbg: PCR_SA_DBG ablation mask of tuning builds, 0 in production)
#pragma unroll
          for (int e = (s2 * NE) / NS; e < ((s2 + 1) * NE) / NS; e++) c_epi(e >> 2, e & 3, obase);
        }
        PCR_SB();
        bf16x8 nh, nl;
        if (s2 + 1 < NS) {
          nh = xb[(4 * (s2 + 1)) * ROWS];
          if constexpr (LO) nl = xb[(4 * (s2 + 1) + 1) * ROWS];
        }
        PCR_SB();
        if (!(a.dbg & 4)) {
          if constexpr (LO) {
            y = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w2h[s2], xl, y, 0, 0, 0);
            y = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w2l[s2], xh, y, 0, 0, 0);
          }
          y = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w2h[s2], xh, y, 0, 0, 0);
        }
        PCR_SB();
        if (s2 + 1 < NS) {
          xh = nh;
          if constexpr (LO) xl = nl;
        }
      }
      PCR_WMARK(1);
      bf_store_tile<LO>(x2, ROWS, y, cb2, rb2, j, h);
    }
    PCR_WMARK(2);
    __syncthreads();
    PCR_WMARK(3);
    // ---- step 2: layer 3 (transposed) of cout block cb3, row blocks rb3 + k G3; between its MFMA triples layer 1 of the
    // next tile and the previous tile's output
    {
#pragma unroll
      for (int k = 0; k < NRB3; k++) {                 // (one row block at a time: 16 accumulator registers, not 32)
        f32x16 y3;
#pragma unroll
        for (int rr = 0; rr < 16; rr++) y3[rr] = sv3;
        const bf16x8 *xa = x2u + 2 * h * ROWS + (rb3 + k * G3) * 32 + j;
        bf16x8 ah = xa[0], al;
        if constexpr (LO) al = xa[ROWS];
#pragma unroll
        for (int s2 = 0; s2 < NS; s2++) {
          const int slot = k * NS + s2;
          if (a.dbg & 2) {
          } else if (slot == 0 || slot == 1) {
            // (the two waves of a SIMD take their share of the previous tile's output one triple apart: one of them
            // always has matrix work while the other waits for its LDS reads)
            if (have_prev && slot == (w >> 2)) d_out(td_prev.x, td_prev.y, td_prev.z);
          } else if (have_next) {
            // layer 1 of the next tile (its table pieces landed during step 1), early in the step: the row entry of the
            // tile after it then has the rest of the step to arrive, and the step's tail stays free of vector work
            if (slot == 2) {
              a_mfma();
              fetch_row(tile + 2 * t_step);              // (rv is free once the two MFMAs hold their operands)
            }
            if (slot == 4) a_store(x1n, 0, 0);
            if (slot == 5) a_store(x1n, 0, 1);
            if (slot == 6) a_store(x1n, 1, 0);
            if (slot == 7) a_store(x1n, 1, 1);
          }
          PCR_SB();
          bf16x8 nh, nl;
          if (s2 + 1 < NS) {
            nh = xa[(4 * (s2 + 1)) * ROWS];
            if constexpr (LO) nl = xa[(4 * (s2 + 1) + 1) * ROWS];
          }
          PCR_SB();
          if (!(a.dbg & 8)) {
            if constexpr (LO) {
              y3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, w3h[s2], y3, 0, 0, 0);
              y3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, w3l[s2], y3, 0, 0, 0);
            }
            y3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, w3h[s2], y3, 0, 0, 0);
          }
          PCR_SB();
          if (s2 + 1 < NS) {
            ah = nh;
            if constexpr (LO) al = nl;
          }
        }
#pragma unroll
        for (int g = 0; g < 4; g++)
#pragma unroll
          for (int pr = 0; pr < 2; pr++)
            pm[k][2 * g + pr] = imax(__float_as_int(y3[4 * g + 2 * pr]), __float_as_int(y3[4 * g + 2 * pr + 1]));
      }
    }
    PCR_WMARK(4);
    __syncthreads();
    PCR_WMARK(5);
#ifdef PCR_SA_TRACE_BUILD
    trace_it++;
#endif
    td_prev = td_cur;
    td_cur = td_next;
    asm volatile("" : "+v"(nn_x), "+v"(nn_y), "+v"(nn_z));
    td_next = make_int4(__builtin_amdgcn_readfirstlane(nn_x), __builtin_amdgcn_readfirstlane(nn_y),
                        __builtin_amdgcn_readfirstlane(nn_z), 0);
    pe_lo = e_lo;
    pe_hi = e_hi;
    p_ngr = ngr;
    have_prev = true;
  }
  // ---- drain: the last tile's pair maxima and output
  if (have_prev) {
    int *obase = obuf + (16 * rb3 + 2 * h) * C3 + cb3 * 32 + j;
    c_offsets(pe_lo, pe_hi);
#pragma unroll
    for (int e = 0; e < NRB3 * 4; e++) c_epi(e >> 2, e & 3, obase);
    __syncthreads();
    d_out(td_prev.x, td_prev.y, td_prev.z);
  }
#undef PCR_SB
}
#endif

// PCR_SA_TRACE=<file> (diagnostics only): wave 0 of every workgroup stamps the shader clock at the phase
// boundaries of its first kTraceTiles tiles; the host dumps the buffer after the launch

// W1 != 0: layer 1 on the matrix core as well.  relu(Wa dxyz + P[i] + shift) is a one-k-block dense call on the
// [dx;dy;dz;0..] rows of the tile (seeded with the shift) whose epilogue adds the table pieces, which each lane
// gathers directly in the accumulator layout (token = lane, four runs of four couts): ~100 instructions per tile
// instead of ~500 of VALU / LDS work.  (Host picks it when c1 has at most four cout blocks and, with a table, TB = 2.)
template <int TB, int NR, int W2, int W3, int NR2 = NR, int W1 = 0, int PREC = 0>
__global__ __launch_bounds__(kThreads) void sa_rag_kernel(RagArgs a) {
  constexpr int ROWS = 32 * TB, RP = ROWS + 1, NG = ROWS / kRagG, CT = kRagCT;
  constexpr bool kL1M = W1 != 0;
  constexpr bool kImg = PREC != 0 && kL1M;   // activations between the layers as a bf image (tile_dense.h)
  constexpr bool kLoImg = PREC == 1;
  const int C3P = ceil32(a.c3);           // gmax is [NG row groups][C3P]
  constexpr int QS = kThreads / ROWS;     // channel quads advance by QS per item: a thread keeps ONE row
  constexpr int NI = 8;                   // 16-byte table pieces a thread holds in registers
  const bool tracing = (a.dbg & 256) && threadIdx.x == 0 && blockIdx.x < kTraceWgs;
  unsigned long long *trace = g_rag_trace + (size_t)blockIdx.x * (2 + kTraceTiles * kTraceMarks);
  int trace_it = 0;
  if (tracing) {
    trace[0] = __builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11));    // HW_REG_HW_ID, 32 bits
    trace[1] = __builtin_amdgcn_s_getreg((20) | (0 << 6) | (31 << 11));   // HW_REG_XCC_ID
  }
#ifdef PCR_SA_TRACE_BUILD   // diagnostic builds only (the stamps cost registers in the hot kernel)
#define PCR_MARK(m)                                                                                    \
  do {                                                                                                 \
    if (tracing && trace_it < kTraceTiles) trace[2 + trace_it * kTraceMarks + (m)] = __builtin_readcyclecounter(); \
  } while (0)
#else
#define PCR_MARK(m) do { (void)tracing; (void)trace; } while (0)
#endif
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int c1 = a.c1, c2 = a.c2, c3 = a.c3;
  // XCD-aware tile ranges: workgroup ids are dealt round-robin over the 8 XCDs (own L2 each); the flat tile list is
  // ordered by cloud, so XCD x (ids = x mod 8) takes the contiguous eighth [t_lo, t_end) of it and its workgroups stride
  // through that range -- the tiles of one cloud, which gather rows of the same table, then share one L2
  const int n_all = a.ws[a.B];
  const bool xaware = gridDim.x >= 8;      // (fewer than eight workgroups: plain striding, every range needs an owner)
  const int xcd = blockIdx.x & 7;
  const int t_step = xaware ? ((int)gridDim.x + 7 - xcd) >> 3 : (int)gridDim.x;
  const int t_lo = xaware ? (int)((long long)n_all * xcd / 8) : 0;
  const int total = xaware ? (int)((long long)n_all * (xcd + 1) / 8) : n_all;
  const int t_first = xaware ? t_lo + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
  const int4 *flat = reinterpret_cast<const int4 *>(a.ws + rag_flat_off(a.B, a.maxT));
  const int *ctab = a.ws + rag_ctab_off(a.B, a.maxT);
  const f32x4 *rowtab = reinterpret_cast<const f32x4 *>(a.ws + rag_rowtab_off(a.B, a.maxT, ROWS));
  const int rowsC = c1 > ceil32(c2) ? c1 : ceil32(c2);
  float *buf = smem;                                          // [rowsC][RP]
  float *gmax = buf + rowsC * RP;                             // [NG][ceil32(c3)], 16-byte aligned
  float *s_wa = gmax + NG * C3P;                              // [c1][3] dxyz weights and [ceil32(c1)] shift of
  float *s_sh1 = s_wa + 3 * c1;                               //   layer 1, staged once per (persistent) workgroup
  float *s_sh2 = s_sh1 + ceil32(c1);                          // accumulator seeds of layers 2 / 3 (zero-padded)
  float *s_sh3 = s_sh2 + ceil32(c2);
  float *sdx8 = s_sh3 + C3P;                                  // [8][RP] (kL1M): dx, dy, dz of the tile's rows, 5 zero rows
  const int tid = threadIdx.x;
  for (int e = tid; e < 3 * c1; e += kThreads) s_wa[e] = a.wa[e];
  for (int e = tid; e < ceil32(c1); e += kThreads) s_sh1[e] = e < c1 ? a.sh1[e] : 0.f;
  if constexpr (kL1M)
    for (int e = tid; e < 8 * RP; e += kThreads) sdx8[e] = 0.f;
  for (int e = tid; e < ceil32(c2); e += kThreads) s_sh2[e] = a.sh2[e];
  for (int e = tid; e < C3P; e += kThreads) s_sh3[e] = a.sh3[e];
  // weight fragments of the NEXT dense call, requested as soon as the previous call's k-loop is over (explicit
  // shapes only): the L2 round trip then never sits between a barrier and the first MFMA
  constexpr bool kRing = PCR_RING && W2 != 0 && W3 != 0 && PREC == 0;   // (the bf16 tile streams its own ring)
  f32x4 ring2[PCR_PF][DenseShape<NR2, W2>::nr], ring3[PCR_PF][DenseShape<NR, W3>::nr];
  constexpr int kBfSel = TB == 2 ? PCR_BF_RING_RAG2 : PCR_BF_RING_RAG4;
  constexpr bool kBfR2 = PREC != 0 && W2 != 0 && W3 != 0 && (kBfSel & 1);
  constexpr bool kBfR3 = PREC != 0 && W2 != 0 && W3 != 0 && (kBfSel & 2);
  BfRingOf<NR2, W2> bring2;
  BfRingOf<NR, W3> bring3;
  auto load_ring2 = [&]() {
    if constexpr (kRing && PCR_RING == 1) tile_dense_ring_load<DenseShape<NR2, W2>::nr, DenseShape<NR2, W2>::ways>(a.wp2, c1, ceil32(c2), ring2);
    if constexpr (kBfR2) bf_ring_load2<PREC, NR2, W2>(a.wp2, c1, ceil32(c2), bring2);
  };
  auto load_ring3 = [&]() {
    if constexpr (kRing) tile_dense_ring_load<DenseShape<NR, W3>::nr, DenseShape<NR, W3>::ways>(a.wp3, ceil8(c2), C3P, ring3);
    if constexpr (kBfR3) bf_ring_load2<PREC, NR, W3>(a.wp3, ceil8(c2), C3P, bring3);
  };
  load_ring2();
  const int r = tid % ROWS, q0 = tid / ROWS;   // this thread's row of every tile, its first channel quad
  const int nq = c1 >> 2;                      // channel quads of layer 1
  const int ni = (nq - q0 + QS - 1) / QS;      // items of this thread: quads q0, q0 + QS, ...
  // PREF: the whole layer-1 gather of the NEXT tile (<= NI pieces per thread) is in flight during layer 3
  const bool pref = a.pq && ni <= NI;
  // per-tile state in registers: the thread's row entry {neighbour index, dxyz}, its table pieces, the tile's
  // segment mask (bit q: row group q closes a centre; wave-uniform)
  f32x4 rv = {__int_as_float(-1), 0.f, 0.f, 0.f};
  f32x4 p4[NI];
  unsigned ends_lo = 0u, ends_hi = 0u, nxt_lo = 0u, nxt_hi = 0u;
  auto fetch_row = [&](int tile) {
    if (tile < total) {
      rv = rowtab[(size_t)tile * ROWS + r];
      nxt_lo = (unsigned)ctab[(size_t)tile * CT];
      nxt_hi = (unsigned)ctab[(size_t)tile * CT + 1];
    }
  };
  const int lane = tid & 63, wv = tid >> 6;
  auto gather = [&](int tile) {   // table pieces of the row held in rv (tile's cloud from the flat list)
    if (tile < total) {
      const size_t bt = (size_t)flat[tile].x;
      if constexpr (kL1M) {
        // accumulator layout (TB = 2, one cout block per wave): tokens l31 and 32 + l31 (their row entries are
        // this lane's and lane ^ 32's), couts 32 wave + 8 g + 4 h .. + 3
        const int own = __float_as_int(rv[0]), other = __shfl_xor(own, 32, 64);
        const int h = lane >> 5;
        const int i0 = h ? other : own, i1 = h ? own : other;
        const float *pr0 = a.pq + (bt * a.N + (size_t)(i0 < 0 ? 0 : i0)) * a.pqw + wv * 32 + 4 * h;
        const float *pr1 = a.pq + (bt * a.N + (size_t)(i1 < 0 ? 0 : i1)) * a.pqw + wv * 32 + 4 * h;
#pragma unroll
        for (int g = 0; g < 4; g++) {
          p4[g] = *reinterpret_cast<const f32x4 *>(pr0 + 8 * g);
          p4[4 + g] = *reinterpret_cast<const f32x4 *>(pr1 + 8 * g);
        }
      } else {
        const float *prow = a.pq + (bt * a.N + (size_t)(__float_as_int(rv[0]) < 0 ? 0 : __float_as_int(rv[0]))) * a.pqw;
#pragma unroll
        for (int u = 0; u < NI; u++) {
          const int oq = q0 + u * QS;
          p4[u] = *reinterpret_cast<const f32x4 *>(prow + 4 * (oq < nq ? oq : 0));
        }
      }
    }
  };
  auto stash_dxyz = [&]() {   // (kL1M) the row entry in rv -> B operand rows of layer 1
    if (q0 == 0) {
      sdx8[r] = rv[1];
      sdx8[RP + r] = rv[2];
      sdx8[2 * RP + r] = rv[3];
    }
  };
  int par = 0;
  __syncthreads();            // (sdx8 zeroed)
  fetch_row(t_first);
  ends_lo = nxt_lo;
  ends_hi = nxt_hi;
  if constexpr (kL1M) stash_dxyz();
  if (pref || (kL1M && a.pq)) gather(t_first);
  __syncthreads();
  for (int tile = t_first; tile < total; tile += t_step, par ^= 1) {
  const int4 td = flat[tile];
  const size_t b = (size_t)td.x;
  const int first = td.y;
  PCR_MARK(0);
  if constexpr (kL1M) {
    const bool hasp = a.pq != nullptr;
    tile_dense2<TB, 1, W1, true>(sdx8, 8, a.wap, ceil32(c1), false,
                                 [&](const f32x16 &acc, int cb, int tb, int l31, int h) {
      if constexpr (kImg) {
        f32x16 v16 = acc;
        if (hasp) {
#pragma unroll
          for (int rr = 0; rr < 16; rr++) v16[rr] += (tb == 0 ? p4[rr >> 2] : p4[4 + (rr >> 2)])[rr & 3];
        }
        bf_store_tile<kLoImg>(buf, ROWS, v16, cb, tb, l31, h);
      } else {
      float *dst = buf + (cb * 32 + 4 * h) * RP + tb * 32 + l31;
#pragma unroll
      for (int rr = 0; rr < 16; rr++) {
        float v = acc[rr];
        if (hasp) v += (tb == 0 ? p4[rr >> 2] : p4[4 + (rr >> 2)])[rr & 3];   // (TB = 2 whenever there is a table)
        dst[((rr & 3) + 8 * (rr >> 2)) * RP] = relu_bits(v);
      }
      }
    }, s_sh1);
  } else
  if (!(a.dbg & 1)) {  // layer 1 (BatchNorm scale folded into wa / P, shift added here): row r, quads q0 + u QS
    const bool live = __float_as_int(rv[0]) >= 0;
    const float dx = rv[1], dy = rv[2], dz = rv[3];
    const float *prow = a.pq ? a.pq + (b * a.N + (size_t)(live ? __float_as_int(rv[0]) : 0)) * a.pqw : nullptr;
    for (int u0 = 0; u0 < ni; u0 += NI) {
      if (!pref && prow) {
#pragma unroll
        for (int u = 0; u < NI; u++) {
          const int oq = q0 + (u0 + u) * QS;
          p4[u] = *reinterpret_cast<const f32x4 *>(prow + 4 * (oq < nq ? oq : 0));
        }
      }
#pragma unroll
      for (int u = 0; u < NI; u++) {
        const int oq = q0 + (u0 + u) * QS;
        if (oq < nq) {
          const int o = oq << 2;
          const float *w = s_wa + o * 3;
          f32x4 v = {0.f, 0.f, 0.f, 0.f};
          if (live) {
#pragma unroll
            for (int j = 0; j < 4; j++) {
              const float t = fmaf(w[3 * j + 2], dz, fmaf(w[3 * j + 1], dy, fmaf(w[3 * j], dx, s_sh1[o + j])));
              v[j] = relu_bits(prow ? t + p4[u][j] : t);
            }
          }
#pragma unroll
          for (int j = 0; j < 4; j++) buf[(o + j) * RP + r] = v[j];
        }
      }
    }
  }
  PCR_MARK(1);
  __syncthreads();
  PCR_MARK(2);
  fetch_row(tile + t_step);   // next tile's row entry: lands during layer 2
  if (!(a.dbg & 2)) {
  if constexpr (kImg)
  tile_dense2p<PREC, TB, NR2, W2, true, true>(buf, c1, a.wp2, ceil32(c2), true,
                                              [&](const f32x16 &acc, int cb, int tb, int l31, int h) {
    bf_store_tile<kLoImg>(buf, ROWS, acc, cb, tb, l31, h);
  }, s_sh2, nullptr, load_ring3, 0, kBfR2 ? &bring2 : nullptr);
  else if constexpr (PREC != 0)
  tile_dense2p<PREC, TB, NR2, W2>(buf, c1, a.wp2, ceil32(c2), true,
                                  [&](float v, int o, int t) { buf[o * RP + t] = relu_bits(v); }, s_sh2,
                                  nullptr, load_ring3, 0, kBfR2 ? &bring2 : nullptr);
  else
  tile_dense2p<PREC, TB, NR2, W2>(buf, c1, a.wp2, ceil32(c2), true,
                                  [&](float v, int o, int t) { buf[o * RP + t] = relu_bits(v); }, s_sh2,
                                  kRing && PCR_RING == 1 ? ring2 : nullptr, load_ring3);
  }
  PCR_MARK(3);
  if (pref || (kL1M && a.pq)) gather(tile + t_step);   // next tile's table pieces: land during layer 3
  __syncthreads();
  PCR_MARK(4);
  if (!(a.dbg & 4))
  tile_dense2p<PREC, TB, NR, W3, true, kImg>(buf, ceil8(c2), a.wp3, ceil32(c3), false,
                                            [&](const f32x16 &acc, int cb, int tb, int l31, int h) {
    // row-pair maxima -> gmax[group][cout]: the 16 accumulator rows of a lane are four runs of four consecutive
    // couts, so the first lane of every pair stores four 16-byte pieces
    // (signed maxima of the bit patterns; the ReLU is the scan's max with 0: see relu_bits)
    static_assert(kRagG == 2, "the DPP step below pairs lanes l and l ^ 1");
    f32x4 g4[4];
#pragma unroll
    for (int rr = 0; rr < 16; rr++) {
      int v = __float_as_int(acc[rr]);
      v = imax(v, dpp_i32<0xB1>(v));    // lane ^ 1
      g4[rr >> 2][rr & 3] = __int_as_float(v);
    }
    if ((l31 & 1) == 0) {
      float *gq = gmax + (tb * 16 + (l31 >> 1)) * C3P + cb * 32 + 4 * h;
#pragma unroll
      for (int g = 0; g < 4; g++) *reinterpret_cast<f32x4 *>(gq + 8 * g) = g4[g];
    }
  }, s_sh3, kRing ? ring3 : nullptr, load_ring2, 0, kBfR3 ? &bring3 : nullptr);
  PCR_MARK(5);
  if constexpr (kL1M) stash_dxyz();   // next tile's dxyz rows (layer 1 of this tile is long done)
  __syncthreads();
  PCR_MARK(6);
  if (!(a.dbg & 8)) {
    // one cout per thread; all quad maxima of the tile are read first (independent, conflict-free LDS reads),
    // then scanned with the wave-uniform `ends` mask: values are >= 0 after the ReLU, so 0 starts a segment
    if (tid < c3) {   // c3 <= 256 = kThreads
      const int o = tid;
      const float *g = gmax + o;
      // a wave issues one instruction every few cycles whatever its kind: the per-step work is kept to a max,
      // a bit test and (at a centre's last group) one store through a running 32-bit offset.  The mask has no bit
      // beyond the tile's last group, so whatever the unused groups hold is never stored.
      float *base = a.out_pm ? a.out + (b * a.S + first) * c3 : a.out + b * c3 * a.S + first;
      unsigned off = a.out_pm ? (unsigned)o : (unsigned)o * (unsigned)a.S;
      const unsigned step = a.out_pm ? (unsigned)c3 : 1u;
      int m = 0;   // signed max of the bit patterns starting from +0 = max over the groups, then ReLU (relu_bits)
#pragma unroll
      for (int half = 0; half < NG / 32; half++) {
        const unsigned ends = (unsigned)__builtin_amdgcn_readfirstlane((int)(half ? ends_hi : ends_lo));
        float gv[32];
#pragma unroll
        for (int q = 0; q < 32; q++) gv[q] = g[(half * 32 + q) * C3P];
#pragma unroll
        for (int q = 0; q < 32; q++) {
          m = imax(m, __float_as_int(gv[q]));
          if ((ends >> q) & 1u) {
            base[off] = __int_as_float(m);
            off += step;
            m = 0;
          }
        }
      }
    }
  }
  ends_lo = nxt_lo;
  ends_hi = nxt_hi;
  PCR_MARK(7);
  trace_it++;
  // no barrier needed here: the next tile's layer 1 writes buf (dead since layer 3), its layer 3 writes gmax
  // only after two more barriers
  }
}

#if PCR_SA_PREC == 0
// y (B,L,cout) POINT-major = W x for x (B,cin,L) channel-major; cout <= 256, no activation.
struct DensePmArgs {
  const float *x, *wp;
  float *y;
  int cin, cout, L, x_pm;
  const float *xyz, *wxyz;  // pcr_dense_pm_xyz_f32: (B,L,3) and (cout,4) {wx, wy, wz, bias}, or null
  int q_rows, q_off;        // pcr_dense_pm_xyz_f32: tokens >= q_rows get the couts [0, q_off) only (q_rows = L: all couts)
};

// NR: cout-block rounds per wave (2 when cout > 128).  X and Y share one LDS buffer (barrier between the
// k-loop and the epilogue), so a 128 -> 128 table needs 33 KB and four workgroups fit a CU.
// PREC: 0 = f32-input MFMA; 1 / 2 = the bf16 matrix core (split / plain; wp = the bf16 image, windows of 256 couts are
// 8 cout blocks = 1024 sixteen-byte units further in)
template <int NR, int PREC = 0>
__global__ __launch_bounds__(kThreads) void dense_pm_kernel(DensePmArgs a) {
  constexpr int TB = 2, T = 64, RP = 65;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  // couts beyond 256 (the tables of 256- / 512-channel SA layers) go in windows of 256 along grid.z: the window's rows
  // of the packed image start 8 floats per cout further in, the k-block stride stays that of the whole image
  const int cinP = ceil8(a.cin), cout = a.cout;
  const int w0 = blockIdx.z * 256;
  int wc = cout - w0 < 256 ? cout - w0 : 256;
  float *X = smem;   // [max(cinP, ceil32(wc))][RP]
  const size_t b = blockIdx.y;
  const int t0 = blockIdx.x * T;
  // (pcr_dense_pm_xyz_f32 with q_rows < L: the Q half of the SA tables is read for CENTRES only -- the first q_rows points
  // under prefix sampling -- so the tiles behind them compute and store the P half alone)
  if (a.xyz && t0 >= a.q_rows) {
    if (w0 >= a.q_off) return;
    if (w0 + wc > a.q_off) wc = a.q_off - w0;
  }
  if (a.x_pm) load_tile_pm(X, RP, a.x + b * a.cin * a.L, a.cin, cinP, a.L, t0, T);
  else load_tile(X, RP, a.x + b * a.cin * a.L, a.cin, cinP, a.L, t0, T);
  // pcr_dense_pm_xyz_f32: the tile's coordinates and the window's rows of {wx, wy, wz, bias}, behind the activation buffer
  // (the launcher sized it), for the store phase
  float *s_xyz = X + (cinP > ceil32(wc) ? cinP : ceil32(wc)) * RP;
  f32x4 *s_wx = reinterpret_cast<f32x4 *>(s_xyz + 3 * T);     // [wc] the window's rows {wx, wy, wz, bias}
  if (a.xyz) {
    if (threadIdx.x < 3 * T) {
      const int t = threadIdx.x / 3;
      s_xyz[threadIdx.x] = t0 + t < a.L ? a.xyz[(b * a.L + t0) * 3 + threadIdx.x] : 0.f;
    }
    if (threadIdx.x < wc) s_wx[threadIdx.x] = reinterpret_cast<const f32x4 *>(a.wxyz)[w0 + threadIdx.x];
  }
  __syncthreads();
  tile_dense2p<PREC, TB, NR>(X, cinP, a.wp + (size_t)w0 * (PREC == 0 ? 8 : 16), ceil32(wc), true,
                             [&](float v, int o, int t) { X[o * RP + t] = v; }, nullptr, nullptr, DenseNoHook(), ceil32(cout));
  __syncthreads();
  float *out = a.y + (b * a.L + t0) * (size_t)cout + w0;
  if ((cout & 3) == 0) {
    // 16-byte stores; item e = (token e / Q, cout quad e % Q) advances by kThreads: one division, then increments
    const int Q = wc >> 2, total = Q * T;
    const int dt = kThreads / Q, dq = kThreads - dt * Q;
    int t = threadIdx.x / Q, q = threadIdx.x - t * Q;
    for (int e = threadIdx.x; e < total; e += kThreads) {
      if (t0 + t < a.L) {
        const float *xs = X + 4 * q * RP + t;
        f32x4 v = {xs[0], xs[RP], xs[2 * RP], xs[3 * RP]};
        if (a.xyz) {
          const float px = s_xyz[3 * t], py = s_xyz[3 * t + 1], pz = s_xyz[3 * t + 2];
#pragma unroll
          for (int c = 0; c < 4; c++) {   // (scalar fmas, pinned: see dense_pm_xyz_res_kernel)
            const f32x4 w = s_wx[4 * q + c];
            float acc = v[c] + w[3];
            asm volatile("" : "+v"(acc));
            acc = __builtin_fmaf(w[2], pz, acc);
            asm volatile("" : "+v"(acc));
            acc = __builtin_fmaf(w[1], py, acc);
            asm volatile("" : "+v"(acc));
            v[c] = __builtin_fmaf(w[0], px, acc);
          }
        }
        *reinterpret_cast<f32x4 *>(out + (size_t)t * cout + 4 * q) = v;
      }
      t += dt;
      q += dq;
      if (q >= Q) { q -= Q; t++; }
    }
  } else {
    for (int e = threadIdx.x; e < wc * T; e += kThreads) {
      const int t = e / wc, c = e - t * wc;
      if (t0 + t < a.L) out[(size_t)t * cout + c] = X[c * RP + t];
    }
  }
}

// The same tables for POINT-major input on the bf16 matrix core, cin = 16 NPI in {32, 64, 128}, cout = 16 NPO in {64, 128},
// whole 64-token tiles of the flattened (B L) token axis (round 5).  dense_pm_kernel is a one-shot workgroup per tile:
// 32 KB in, 32 KB out and the 64 KB weight image streamed from L2 behind every tile, its load -> k-loop -> store phases
// end to end with nothing of the next tile in flight (ssg1024's SA2 table: 2.1 GB in 0.88 ms).  Here a PERSISTENT
// workgroup keeps every step of its waves' weight rows in registers (tile_dense_bf_impl's RES ring: 8 steps x hi / lo =
// 64 VGPRs), so the k-loop issues no vector-memory load at all -- which is what lets the NEXT tile's rows, requested
// into registers before the k-loop, arrive behind it (a wave's loads retire in order: with a weight ring in the loop
// the first refill would wait for them).  Same per-tile arithmetic in the same order: bit-identical to dense_pm_kernel.
template <int NS, int NPI, int NPO>
__global__ __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(NPI == 8 ? 2 : 4, NPI == 8 ? 2 : 4)))
void dense_pm_res_kernel(DensePmArgs a, long ntile) {
  constexpr int TB = 2, T = 64, RP = 65, PF = 8;
  constexpr int CIN = 16 * NPI, COUT = 16 * NPO, Q = CIN / 4, Q2 = COUT / 4;
  constexpr int LQ = NPI == 8 ? 5 : (NPI == 4 ? 4 : 3), LQ2 = NPO == 8 ? 5 : 4;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float *X = smem;   // [max(CIN, COUT)][RP]
  const int tid = threadIdx.x;
  BfRing<PF, 1> ring;
  bf_ring_load<1, 1, PF, NS>(a.wp, CIN, ceil32(COUT), ring);
  const int tq = tid >> LQ, q = tid & (Q - 1);          // piece u of a thread: token tq + u (256 / Q), channel quad q
  const int tq2 = tid >> LQ2, q2 = tid & (Q2 - 1);
  f32x4 v[NPI];
  auto request = [&](long tile) __attribute__((always_inline)) {
    const float *src = a.x + ((size_t)tile * T + tq) * CIN + 4 * q;
#pragma unroll
    for (int u = 0; u < NPI; u++) v[u] = *reinterpret_cast<const f32x4 *>(src + (size_t)u * (kThreads / Q) * CIN);
  };
  auto epi = [&](float r, int o, int t) { X[o * RP + t] = r; };
  long tile = blockIdx.x;
  if (tile < ntile) request(tile);
  for (; tile < ntile; tile += gridDim.x) {
#pragma unroll
    for (int u = 0; u < NPI; u++) {
      float *d = X + 4 * q * RP + tq + u * (kThreads / Q);
      d[0] = v[u][0];
      d[RP] = v[u][1];
      d[2 * RP] = v[u][2];
      d[3 * RP] = v[u][3];
    }
    if (tile + gridDim.x < ntile) request(tile + gridDim.x);
    __syncthreads();
    tile_dense_bf_impl<TB, 1, 1, false, NS, decltype(epi), PF, DenseNoHook, false, true>(
        X, CIN, a.wp, ceil32(COUT), true, epi, nullptr, DenseNoHook(), 0, &ring);
    __syncthreads();
    float *out = a.y + ((size_t)tile * T + tq2) * COUT + 4 * q2;
#pragma unroll
    for (int u = 0; u < NPO; u++) {
      const float *xs = X + 4 * q2 * RP + tq2 + u * (kThreads / Q2);
      *reinterpret_cast<f32x4 *>(out + (size_t)u * (kThreads / Q2) * COUT) = f32x4{xs[0], xs[RP], xs[2 * RP], xs[3 * RP]};
    }
    __syncthreads();
  }
}

// The tables WITH the coordinate term (pcr_dense_pm_xyz_f32) in the persistent form, for CHANNEL-major input -- what the
// attention blocks hand the Point-Transformer's SA layers: cin = 16 NPI in {32, 64}, cout = 16 NPO in {128, 256}, clouds of
// whole 64-token tiles.  As dense_pm_res_kernel: every step of a wave's weight rows in registers (NPI steps x hi / lo x one
// or two cout blocks), the NEXT tile's rows and coordinates requested into registers before the k-loop; the window's
// {wx, wy, wz, bias} rows of the thread's cout quad are read once per workgroup.  (The one-shot dense_pm_kernel ran these
// launches latency-bound: 0.19 / 0.22 ms for pt1024's two tables against 0.11 each at 5 TB/s.)  Same per-tile arithmetic in the
// same order as dense_pm_kernel with the term in its store phase: bit-identical tables.
template <int NS, int NPI, int NPO>
__global__ __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(NPO == 16 ? 2 : 3, NPO == 16 ? 2 : 3)))
void dense_pm_xyz_res_kernel(DensePmArgs a, int tpc, long ntile) {
  constexpr int TB = 2, T = 64, RP = 65, PF = NPI, NR = NPO == 16 ? 2 : 1;
  constexpr int CIN = 16 * NPI, COUT = 16 * NPO, Q2 = COUT / 4, LQ2 = NPO == 16 ? 6 : 5;
  static_assert((PF & 1) == 0 && (NPO == 8 || NPO == 16), "cin in {32, 64}, cout in {128, 256}");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float *X = smem;                                        // [COUT][RP]
  float *s_xyz = X + COUT * RP;                           // [3 T]
  const int tid = threadIdx.x;
  BfRing<PF, NR> ring;
  bf_ring_load<NR, 1, PF, NS>(a.wp, CIN, ceil32(COUT), ring);
  const int ch0 = tid >> 4, tq4 = tid & 15;               // piece u of a thread: channel ch0 + 16 u, tokens 4 tq4 .. + 3
  const int tq2 = tid >> LQ2, q2 = tid & (Q2 - 1);        // store phase: token tq2 + u (256 / Q2), cout quad q2
  f32x4 wq[4];
  {
    const f32x4 *wr = reinterpret_cast<const f32x4 *>(a.wxyz) + 4 * q2;
#pragma unroll
    for (int c = 0; c < 4; c++) wq[c] = wr[c];
  }
  f32x4 v[NPI];
  float vx = 0.f;
  auto request = [&](long tile) __attribute__((always_inline)) {
    const long b = tile / tpc;
    const int t0 = (int)(tile - b * tpc) * T;
    const float *src = a.x + ((size_t)b * CIN + ch0) * a.L + t0 + 4 * tq4;
#pragma unroll
    for (int u = 0; u < NPI; u++) v[u] = *reinterpret_cast<const f32x4 *>(src + (size_t)16 * u * a.L);
    if (tid < 3 * T) vx = a.xyz[((size_t)b * a.L + t0) * 3 + tid];
  };
  auto epi = [&](float r, int o, int t) { X[o * RP + t] = r; };
  long tile = blockIdx.x;
  if (tile < ntile) request(tile);
  for (; tile < ntile; tile += gridDim.x) {
#pragma unroll
    for (int u = 0; u < NPI; u++) {
      float *d = X + (ch0 + 16 * u) * RP + 4 * tq4;
      d[0] = v[u][0];
      d[1] = v[u][1];
      d[2] = v[u][2];
      d[3] = v[u][3];
    }
    if (tid < 3 * T) s_xyz[tid] = vx;
    if (tile + gridDim.x < ntile) request(tile + gridDim.x);
    __syncthreads();
    tile_dense_bf_impl<TB, NR, 1, false, NS, decltype(epi), PF, DenseNoHook, false, true>(
        X, CIN, a.wp, ceil32(COUT), true, epi, nullptr, DenseNoHook(), 0, &ring);
    __syncthreads();
    const long b = tile / tpc;
    const int t0 = (int)(tile - b * tpc) * T;
    // (tokens behind q_rows: the couts [0, q_off) only -- the Q half is read for centres alone)
    if (!(t0 >= a.q_rows && 4 * q2 >= a.q_off)) {
      float *out = a.y + ((size_t)b * a.L + t0 + tq2) * COUT + 4 * q2;
#pragma unroll
      for (int u = 0; u < NPO; u++) {
        const int t = tq2 + u * (kThreads / Q2);
        const float *xs = X + 4 * q2 * RP + t;
        f32x4 r = {xs[0], xs[RP], xs[2 * RP], xs[3 * RP]};
        const float px = s_xyz[3 * t], py = s_xyz[3 * t + 1], pz = s_xyz[3 * t + 2];
        // SCALAR fmas, pinned -- here and in dense_pm_kernel's store phase.  Left to the vectoriser this chain becomes
        // v_pk_fma_f32 pairs over (c, c + 1) and (t, t + 8) whose 64-bit operands are assembled by v_mov_b32 pairs right in
        // front of them, and that form returns a wrong half of ONE packed result in lanes 48-63 now and then, a different
        // place every run: ~3e-4 of the elements in this kernel, ~1e-8 in the one-shot kernel (tools/stress_tables.py against
        // torch fp64; profiles/r06t_stress_tables.txt).  Not an LDS race (no prefetch, weights reloaded per tile, an extra
        // barrier + sleep in front of this phase, no weight ring: each changed nothing; the dense part, the bias and xyz = 0
        // were always exact; every operand register is written only by its loads), and the same signature in two kernels of
        // different structure that goes away with the packed instructions points at that instruction sequence, not at the
        // data flow.  Root cause not found (hipcc 7.2 / gfx950); the general shape -- a packed read of a pair a 32-bit
        // instruction has just written -- is everywhere in this library (every bf16 split) and exact there.  With scalar chains:
        // 0 differing of 1.3e10 elements between the two kernels (tools/stress_tables.py).
#pragma unroll
        for (int c = 0; c < 4; c++) {
          float acc = r[c] + wq[c][3];
          asm volatile("" : "+v"(acc));
          acc = __builtin_fmaf(wq[c][2], pz, acc);
          asm volatile("" : "+v"(acc));
          acc = __builtin_fmaf(wq[c][1], py, acc);
          asm volatile("" : "+v"(acc));
          r[c] = __builtin_fmaf(wq[c][0], px, acc);
        }
        *reinterpret_cast<f32x4 *>(out + (size_t)u * (kThreads / Q2) * COUT) = r;
      }
    }
    __syncthreads();
  }
}

#endif   // PCR_SA_PREC == 0 (table kernel)

}  // namespace

template <int TB, int NR, int W2, int W3, int NR2 = NR, int RKB = 0>
static void sa2_launch_one(const Sa2Args &a, bool maxe, size_t lds, hipStream_t st, dim3 grid) {
  // bf16 forms: layers whose first layer runs on the matrix core (NR2 == 1; the launcher has checked a.l1m) keep their
  // activations as bf images
  constexpr bool kImgK = kPrec != 0 && NR2 == 1;
  if (maxe) {
    static bool ok = allow_big_lds(sa_fused_kernel<TB, NR, W2, W3, true, NR2, RKB, kPrec, kImgK>);
    (void)ok;
    hipLaunchKernelGGL((sa_fused_kernel<TB, NR, W2, W3, true, NR2, RKB, kPrec, kImgK>), grid, dim3(kThreads), lds, st, a);
  } else if constexpr (kPrec == 0) {
    static bool ok = allow_big_lds(sa_fused_kernel<TB, NR, W2, W3, false, NR2, RKB, kPrec>);
    (void)ok;
    hipLaunchKernelGGL((sa_fused_kernel<TB, NR, W2, W3, false, NR2, RKB, kPrec>), grid, dim3(kThreads), lds, st, a);
  }
}

// wsel: 1 / 2 / 4 when both MFMA layers have the same cout class (specialised bodies), else 0
// nr2 / nr: cout-block rounds of layer 2 / of the wider of the two layers
template <int TB>
static int sa2_launch_tb(const Sa2Args &a, int nr, int nr2, int w2, int w3, bool maxe, size_t lds, hipStream_t st,
                         dim3 grid) {
  const int wsel = kPrec != 0 ? 10 * w2 + w3 : (w2 == w3 ? w2 : 0);
  // resident weight fragments for 32-channel layers (2 x 16 VGPRs).  For 64-channel layers the 2 x 32 VGPRs cost
  // more residency than the saved L2 round trips are worth (measured: 6.4 -> 7.9 ms on the 64/64/64 layer).
  const bool narrow4 = a.c1 <= 32 && a.c2 <= 32;
  if constexpr (kPrec != 0) {
    // bf16 forms: explicit shapes, max-from-accumulators epilogue, at most two cout-block rounds (the callers checked);
    // w23 = 10 * (ways of layer 2) + (ways of layer 3)
    (void)narrow4;
    if (!maxe || nr > 2) return -1;
    if (nr2 == 1 && !a.l1m) return -1;     // (the one-round instantiations are the bf-image ones: layer 1 on the matrix core)
    if (wsel == 44) sa2_launch_one<TB, 1, 4, 4>(a, maxe, lds, st, grid);
    else if (wsel == 22) sa2_launch_one<TB, 1, 2, 2>(a, maxe, lds, st, grid);
    else if (wsel == 21) sa2_launch_one<TB, 1, 2, 1>(a, maxe, lds, st, grid);
    else if (wsel == 11 && nr == 1) sa2_launch_one<TB, 1, 1, 1>(a, maxe, lds, st, grid);
    else if (wsel == 11 && nr2 == 1) sa2_launch_one<TB, 2, 1, 1, 1>(a, maxe, lds, st, grid);
    else if (wsel == 11) sa2_launch_one<TB, 2, 1, 1>(a, maxe, lds, st, grid);
    else return -1;
    return 0;
  } else {
  if (nr == 4) {
    if constexpr (TB <= 2) sa2_launch_one<TB, 4, 1, 1, 4>(a, maxe, lds, st, grid);
    else return -1;
  } else if (wsel == 4 && narrow4) sa2_launch_one<TB, 1, 4, 4, 1, 4>(a, maxe, lds, st, grid);
  else if (wsel == 4) sa2_launch_one<TB, 1, 4, 4>(a, maxe, lds, st, grid);
  else if (wsel == 2) sa2_launch_one<TB, 1, 2, 2>(a, maxe, lds, st, grid);
  else if (wsel == 1 && nr == 1) sa2_launch_one<TB, 1, 1, 1>(a, maxe, lds, st, grid);
  else if (wsel == 1 && nr2 == 1) sa2_launch_one<TB, 2, 1, 1, 1>(a, maxe, lds, st, grid);
  else if (wsel == 1) sa2_launch_one<TB, 2, 1, 1>(a, maxe, lds, st, grid);
  else if (nr == 1) sa2_launch_one<TB, 1, 0, 0>(a, maxe, lds, st, grid);
  else if (nr2 == 1) sa2_launch_one<TB, 2, 0, 0, 1>(a, maxe, lds, st, grid);
  else sa2_launch_one<TB, 2, 0, 0>(a, maxe, lds, st, grid);
  return 0;
  }
}

extern "C" int pcr_dense_pm_f32(const float *, const float *, float *, int, int, int, int, int, pcr_stream_t);

// diagnostics only (PCR_SA_TRACE): synchronises, appends one launch's phase stamps to the file
static void rag_dump_trace(const char *path, const char *tag, int wgs) {
  static int launches = 0;
  static const int skip = pcr_tune_int("PCR_SA_TRACE_SKIP");   // (a freshly started process runs its first launches cold)
  launches++;
  if (launches <= skip || launches > skip + 8) return;   // a few launches are enough
  static std::vector<unsigned long long> host(kTraceWgs * (2 + kTraceTiles * kTraceMarks));
  if (hipDeviceSynchronize() != hipSuccess) return;
  if (hipMemcpyFromSymbol(host.data(), HIP_SYMBOL(g_rag_trace), host.size() * sizeof(unsigned long long)) != hipSuccess)
    return;
  FILE *f = fopen(path, "a");
  if (!f) return;
  const int n = wgs < kTraceWgs ? wgs : kTraceWgs;
  fprintf(f, "launch %d kernel %s wgs %d\n", launches, tag, wgs);
  for (int w = 0; w < n; w++) {
    const unsigned long long *t = host.data() + (size_t)w * (2 + kTraceTiles * kTraceMarks);
    fprintf(f, "wg %d hwid %llu xcc %llu", w, t[0], t[1]);
    for (int i = 0; i < kTraceTiles * kTraceMarks; i++) fprintf(f, " %llu", t[2 + i]);
    fprintf(f, "\n");
  }
  fclose(f);
}

// fast path; returns -1 when the configuration is not covered (caller falls back to sa_mlp_kernel)
#if PCR_SA_PREC == 0
int pcr_sa2_try_bf3(const pcr_sa_params *p, pcr_stream_t st);   // sa_kernels_bf3.hip / sa_kernels_bf1.hip
int pcr_sa2_try_bf1(const pcr_sa_params *p, pcr_stream_t st);
#endif

#if PCR_SA_PREC != 0
// shapes of the wave-autonomous forms (sa_stream_kernel / sa_stream_rag_kernel).  Ball-query layers (mode 1) can run
// either form depending on whether hit counts are given, and the tests hold the two to the same bits: such a layer
// streams only when BOTH forms fit, so both always share one arithmetic.
static bool sas_k_ok(int K) {
  for (int nb = 1; nb <= 3; nb++)
    if ((32 * nb) % K == 0) return (K & 15) == 0;
  return false;
}
static size_t sas_fixed_lds(const pcr_sa_params &p) {
  const int ncb = p.c1 >> 5, ncb3 = p.c3 >> 5;
  return ((size_t)(2 * ncb) * ncb * 128 + (size_t)(2 * ncb) * ncb3 * 128) * 16 + (size_t)(2 * p.c1 + p.c3) * 4 + (size_t)ncb * 64 * 16;
}
static bool sas_shape_ok(const pcr_sa_params &p, bool ragged) {
  static const int no_stream = pcr_tune_int("PCR_SA_NO_STREAM");   // diagnostics
  if (no_stream || p.B < 1 || !p.wa_packed || p.c1 != p.c2 || !(p.c3 == p.c2 || p.c3 == 2 * p.c2) ||
      !(p.c1 == 32 || p.c1 == 64 || p.c1 == 128) || !sas_k_ok(p.K))
    return false;
  const size_t lds_k = sas_fixed_lds(p) + (size_t)kSasWaves * 6 * p.c3 * 4 + 16;
  const size_t lds_r = sas_fixed_lds(p) + (size_t)kSasWaves * sas_rag_wave_ints(p.c3, p.K, false) * 4;
  const size_t cap = (size_t)160 * 1024;
  if ((long)p.B * p.S >= 0x7FFFFFFFl) return false;    // (the kernels count items in 32 bits)
  if (ragged || p.mode == 1) return lds_k <= cap && lds_r <= cap;
  return lds_k <= cap;
}
#endif

static int sa2_try(const pcr_sa_params &p, pcr_stream_t st_) {
  hipStream_t st = pcr_s(st_);
  if (!p.wa || !p.wps[0] || !p.wps[1] || !p.shift_pad[0] || !p.shift_pad[1] || (p.D && (!p.wpq || !p.pq_ws)))
    return -1;
#if PCR_SA_PREC == 0
  // precision 1 / 2: layers 2 and 3 on the bf16 matrix core (their own translation units); shapes those units do not
  // instantiate come back with -1 and run here in f32, which is never less accurate than what was asked for
  if (p.precision != 0 && p.wps_bf[0] && p.wps_bf[1]) {
    const int rc = p.precision == 1 ? pcr_sa2_try_bf3(&p, st_) : pcr_sa2_try_bf1(&p, st_);
    if (rc >= 0) return rc;
  }
  if (!p.idx) return -1;
  const float *const wl2 = p.wps[0], *const wl3 = p.wps[1];
#else
  if (!p.wps_bf[0] || !p.wps_bf[1] || (p.c1 & 31) || (p.c2 & 31)) return -1;
  const float *const wl2 = p.wps_bf[0], *const wl3 = p.wps_bf[1];
  if (!p.idx && !(p.row_tab && p.cnt && p.tile_ws && p.mode == 1 && sas_shape_ok(p, true))) return -1;
#endif
  pcr_note_arith(kPrec);   // (every launch below runs layers 2 / 3 in this unit's arithmetic)
  if ((p.c1 & 7) || p.c1 > 512 || p.c2 > 512 || p.c3 > 512) return -1;
  const int pqw = p.mode == 0 ? 2 * p.c1 : p.c1;
  if (p.D && pqw > 1024) return -1;
  // the persistent, register-pipelined kernel: ball-query groups with hit counts (only the distinct rows are
  // evaluated).  (It also runs count-less featureless layers -- all K rows -- but the K-row kernel with resident
  // weights is faster there: 2.2 vs 2.9 ms on the 32-channel kNN layer, its row tables are 0.5 GB of extra traffic.)
  // (the cout-split kernel's shape also takes count-less launches -- all K rows of every centre -- through the same tile
  // plan, so that ragged and K-row evaluation of that layer share one kernel and one arithmetic)
  const bool wsplit_shape = kPrec != 0 && p.c1 == 128 && p.c2 == 128 && p.c3 == 256 && p.wa_shift_packed && p.K <= 64 &&
                            (size_t)p.B * ((p.S + (64 / rag_ceil(p.K)) - 1) / (64 / rag_ceil(p.K))) * 64 * 16 < 0x7FFFFFFFull;
  if (p.tile_ws && (p.cnt || wsplit_shape) && p.mode == 1 && p.c1 <= 256 && p.c2 <= 256 && p.c3 <= 256) {
    const int n2r = ceil32(p.c2) >> 5, n3r = ceil32(p.c3) >> 5;
    const int nrr = (n2r > 4 || n3r > 4) ? 2 : 1;
    const int tb = nrr == 2 ? 2 : 4;
    const int ROWS = 32 * tb;
    bool shape_ok = true;
    if (kPrec != 0) {   // the bf16 units hold the explicit-shape instantiations only
      const int v2 = n2r >= 3 ? 1 : (n2r == 2 ? 2 : 4), v3 = n3r >= 3 ? 1 : (n3r == 2 ? 2 : 4);
      shape_ok = tb == 2 ? (v2 == 1 && v3 == 1)
                         : ((v2 == 1 && v3 == 1) || (v2 == 2 && v3 == 1) || (v2 == 2 && v3 == 2) || (v2 == 4 && v3 == 4));
    }
    if (!shape_ok) return -1;
#if PCR_SA_PREC != 0
    {
      // wave-autonomous ragged form (shape-only choice, the same shapes as the K-row form so that both paths share one
      // arithmetic): equal widths of layers 1 / 2, c3 = c2 or 2 c2, weights + per-wave areas within LDS
      static const int no_stream = pcr_tune_int("PCR_SA_NO_STREAM");
      const int ncb = p.c1 >> 5, ncb3 = p.c3 >> 5;
      const size_t fixed = ((size_t)(2 * ncb) * ncb * 128 + (size_t)(2 * ncb) * ncb3 * 128) * 16 + (size_t)(2 * p.c1 + p.c3) * 4 +
                           (size_t)ncb * 64 * 16;
      (void)no_stream;
      if (sas_shape_ok(p, true)) {
        if (p.D && !p.pq_ready) {
          const int rc = pcr_dense_pm_f32(p.feat, p.wpq, p.pq_ws, p.B, p.D, p.c1, p.N, p.feat_point_major, st_);
          if (rc != PCR_OK) return rc == PCR_ERR_INVALID ? -1 : rc;
        }
        RagArgs r;
        r.B = p.B; r.N = p.N; r.S = p.S; r.K = p.K; r.c1 = p.c1; r.c2 = p.c2; r.c3 = p.c3; r.maxT = 0;
        r.xyz = p.xyz; r.idx = p.idx; r.cnt = p.cnt; r.centre_idx = p.centre_idx; r.ws = p.tile_ws;
        r.wa = p.wa; r.pq = p.D ? p.pq_ws : nullptr; r.pqw = p.c1;
        r.wap = p.wa_packed;
        r.rowtab = p.row_tab;
        if (!p.row_tab && (!p.idx || !p.cnt)) return PCR_ERR_INVALID;
        r.wp2 = wl2; r.wp3 = wl3; r.sh1 = p.shift[0]; r.sh2 = p.shift_pad[0]; r.sh3 = p.shift_pad[1];
        static const char *strace = pcr_tune_str("PCR_SA_TRACE");
        r.dbg = strace ? 256 : 0; r.out = p.out; r.out_pm = p.out_point_major;
        static const int ncu = [] {
          int dev = 0, n = 0;
          if (hipGetDevice(&dev) != hipSuccess ||
              hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n < 8)
            n = 256;
          return n;
        }();
        const long items = (long)p.B * ((p.S + kSasCpi - 1) / kSasCpi);
        const size_t cap = (size_t)160 * 1024;
        const size_t lds_r8 = fixed + (size_t)8 * sas_rag_wave_ints(p.c3, p.K, r.rowtab != nullptr) * 4;
        const size_t lds_r12 = fixed + (size_t)12 * sas_rag_wave_ints(p.c3, p.K, true) * 4;
        static const int no12 = pcr_tune_int("PCR_SA_NO_W12");   // diagnostics
        const bool w12 = r.rowtab && lds_r12 <= cap && !no12;    // three waves per SIMD (see the kernel)
        const int nw = w12 ? 12 : 8;
        long wgs = (items + nw - 1) / nw;
        if (wgs > ncu) wgs = ncu;
        wgs = (wgs + 7) / 8 * 8;
        const dim3 gg((unsigned)wgs), bb(64 * nw);
        constexpr bool kLoS = kPrec == 1;
#define PCR_SASR(NCBv, NCB3v)                                                                 \
  do {                                                                                        \
    static bool ok = allow_big_lds(sa_stream_rag_kernel<NCBv, NCB3v, kLoS>);                  \
    static bool okt = allow_big_lds(sa_stream_rag_kernel<NCBv, NCB3v, kLoS, true>);           \
    static bool okw = allow_big_lds(sa_stream_rag_kernel<NCBv, NCB3v, kLoS, true, 12>);       \
    (void)ok; (void)okt; (void)okw;                                                           \
    if (w12) hipLaunchKernelGGL((sa_stream_rag_kernel<NCBv, NCB3v, kLoS, true, 12>), gg, bb, lds_r12, st, r); \
    else if (r.rowtab) hipLaunchKernelGGL((sa_stream_rag_kernel<NCBv, NCB3v, kLoS, true>), gg, bb, lds_r8, st, r); \
    else hipLaunchKernelGGL((sa_stream_rag_kernel<NCBv, NCB3v, kLoS>), gg, bb, lds_r8, st, r); \
  } while (0)
        if (ncb == 1 && ncb3 == 1) PCR_SASR(1, 1);
        else if (ncb == 1) PCR_SASR(1, 2);
        else if (ncb3 == 2) PCR_SASR(2, 2);
        else PCR_SASR(2, 4);
#undef PCR_SASR
        if (strace) rag_dump_trace(strace, "stream", (int)(2 * wgs));
        if (hipGetLastError() != hipSuccess) return PCR_ERR_LAUNCH;
        return PCR_OK;
      }
    }
#endif
    if (!p.idx) return -1;
    if (p.K <= ROWS) {
      const int per_tile = ROWS / rag_ceil(p.K);               // whole centres a tile holds in the worst case
      RagArgs r;
      r.B = p.B; r.N = p.N; r.S = p.S; r.K = p.K; r.c1 = p.c1; r.c2 = p.c2; r.c3 = p.c3;
      r.maxT = (p.S + per_tile - 1) / per_tile;
      r.xyz = p.xyz; r.idx = p.idx; r.cnt = p.cnt; r.centre_idx = p.centre_idx; r.ws = p.tile_ws;
      r.wa = p.wa; r.pq = p.D ? p.pq_ws : nullptr; r.pqw = p.c1;
      r.wap = p.wa_packed;
      r.wap4 = p.wa_shift_packed;
      r.wp2 = wl2; r.wp3 = wl3; r.sh1 = p.shift[0]; r.sh2 = p.shift_pad[0]; r.sh3 = p.shift_pad[1];
      static const int rdbg = pcr_tune_int("PCR_SA_DBG");
      static const char *rtrace = pcr_tune_str("PCR_SA_TRACE");
      r.dbg = rdbg | (rtrace ? 256 : 0);
      r.out = p.out;
      r.out_pm = p.out_point_major;
      const int rowsCr = p.c1 > ceil32(p.c2) ? p.c1 : ceil32(p.c2);
      const size_t lds = ((size_t)rowsCr * (ROWS + 1) +
                          (size_t)ceil32(p.c3) * (ROWS / kRagG) + 3 * (size_t)p.c1 + ceil32(p.c1) + ceil32(p.c2) +
                          ceil32(p.c3) + 8 * (ROWS + 1)) *
                         sizeof(float);
      if (lds <= 150 * 1024) {
        if (p.D && !p.pq_ready) {
          const int rc = pcr_dense_pm_f32(p.feat, p.wpq, p.pq_ws, p.B, p.D, p.c1, p.N, p.feat_point_major, st_);
          if (rc != PCR_OK) return rc == PCR_ERR_INVALID ? -1 : rc;
        }
        hipLaunchKernelGGL(sa_rag_plan_kernel, dim3((p.B + 3) / 4), dim3(256), 0, st, r, ROWS);
        hipLaunchKernelGGL(sa_rag_scan_kernel, dim3(1), dim3(1024), 0, st, p.B, r.ws);
        hipLaunchKernelGGL(sa_rag_flatten_kernel, dim3((p.B + 3) / 4), dim3(256), 0, st, p.B, r.maxT, r.ws);
        {
          long long wgs = ((long long)p.B * r.maxT + 3) / 4;
          if (wgs > 4096) wgs = 4096;
          if (tb == 2) hipLaunchKernelGGL(sa_rag_rows_kernel<64>, dim3((unsigned)wgs), dim3(256), 0, st, r);
          else hipLaunchKernelGGL(sa_rag_rows_kernel<128>, dim3((unsigned)wgs), dim3(256), 0, st, r);
        }
        if (hipGetLastError() != hipSuccess) return PCR_ERR_LAUNCH;
        static const int n_cu = [] {
          int dev = 0, n = 0;
          if (hipGetDevice(&dev) != hipSuccess ||
              hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n < 1)
            n = 256;
          return n;
        }();
        const long long max_tiles = (long long)p.B * r.maxT;
        const int w2 = n2r >= 3 ? 1 : (n2r == 2 ? 2 : 4), w3 = n3r >= 3 ? 1 : (n3r == 2 ? 2 : 4);
#if PCR_SA_PREC != 0
        {
          // cout-split form with register-resident weights (shape-only choice): 128 / 128 / 256 on 64-row tiles
          static const int no_wsplit = pcr_tune_int("PCR_SA_NO_WSPLIT");   // diagnostics
          if (!no_wsplit && tb == 2 && wsplit_shape) {
            constexpr bool kLoW = kPrec == 1;
            static bool okw = allow_big_lds(sa_wsplit_rag_kernel<4, 8, kLoW>);
            (void)okw;
            const size_t lds_w = ((size_t)3 * 128 * 64 + (size_t)32 * 256 + 2 * 128 + 64) * sizeof(float);   // X1 x 2, X2, obuf, seeds, coff
            long long want = n_cu;                                           // persistent: one 8-wave workgroup per CU
            if (want > max_tiles) want = max_tiles;
            hipLaunchKernelGGL((sa_wsplit_rag_kernel<4, 8, kLoW>), dim3((unsigned)want), dim3(512), lds_w, st, r);
            if (rtrace) rag_dump_trace(rtrace, "wsplit", (int)(2 * want));
            if (hipGetLastError() != hipSuccess) return PCR_ERR_LAUNCH;
            return PCR_OK;
          }
        }
#endif
#define PCR_RAG(TBv, NRv, A2, A3, NR2v, W1v)                                                             \
  do {                                                                                                   \
    auto kern = sa_rag_kernel<TBv, NRv, A2, A3, NR2v, W1v, kPrec>;                                       \
    static bool ok = allow_big_lds(kern);                                                                \
    (void)ok;                                                                                            \
    static size_t occ_lds = 0;                                                                           \
    static int occ = 0;   /* resident workgroups per CU for this LDS size (registers and LDS) */          \
    if (occ_lds != lds) {                                                                                \
      int n = 0;                                                                                         \
      if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, kern, kThreads, lds) != hipSuccess || n < 1)  \
        n = 1;                                                                                           \
      occ = n;                                                                                           \
      occ_lds = lds;                                                                                     \
    }                                                                                                    \
    long long want = (long long)n_cu * occ;                                                              \
    if (want > max_tiles) want = max_tiles;                                                              \
    hipLaunchKernelGGL(kern, dim3((unsigned)want), dim3(kThreads), lds, st, r);                          \
    if (rtrace) rag_dump_trace(rtrace, #TBv "," #NRv, (int)want);                                        \
  } while (0)
        const int n1r = ceil32(p.c1) >> 5;
        const int w1 = n1r >= 3 ? 1 : (n1r == 2 ? 2 : 4);
        const bool l1m = p.wa_packed && n1r <= 4;   // layer 1 on the matrix core
        if (tb == 2) {
          const bool narrow2 = n2r <= 4;   // layer 2 needs one cout-block round only
          if (w2 == 1 && w3 == 1 && narrow2 && l1m && w1 == 1) PCR_RAG(2, 2, 1, 1, 1, 1);
          else if (w2 == 1 && w3 == 1 && narrow2) PCR_RAG(2, 2, 1, 1, 1, 0);
          else if (w2 == 1 && w3 == 1) PCR_RAG(2, 2, 1, 1, 2, 0);
          else if constexpr (kPrec == 0) {
            if (narrow2) PCR_RAG(2, 2, 0, 0, 1, 0);
            else PCR_RAG(2, 2, 0, 0, 2, 0);
          }
        } else {
          if (w2 == 1 && w3 == 1) PCR_RAG(4, 1, 1, 1, 1, 0);
          else if (w2 == 2 && w3 == 1 && l1m && w1 == 2 && !p.D) PCR_RAG(4, 1, 2, 1, 1, 2);
          else if (w2 == 2 && w3 == 1) PCR_RAG(4, 1, 2, 1, 1, 0);
          else if (w2 == 2 && w3 == 2) PCR_RAG(4, 1, 2, 2, 1, 0);
          else if (w2 == 4 && w3 == 4) PCR_RAG(4, 1, 4, 4, 1, 0);
          else if constexpr (kPrec == 0) PCR_RAG(4, 1, 0, 0, 1, 0);
        }
#undef PCR_RAG
        if (hipGetLastError() != hipSuccess) return PCR_ERR_LAUNCH;
        return PCR_OK;
      }
    }
  }
  const bool maxe = (p.K & 15) == 0;
  int rowsC = p.c1 > ceil32(p.c2) ? p.c1 : ceil32(p.c2);
  if (!maxe && ceil32(p.c3) > rowsC) rowsC = ceil32(p.c3);
  auto lds_bytes = [&](int tb, int cpw) {
    size_t stage = (size_t)6 * 32 * tb + (size_t)cpw * p.c1 + 32;   // (sq read in 16-byte pieces up to ceil32(c1))
    const size_t gm = maxe ? (size_t)ceil32(p.c3) * 2 * tb : 0;
    if (gm > stage) stage = gm;
    return ((size_t)rowsC * (32 * tb + 1) + stage) * sizeof(float);
  };
  const int n2 = ceil32(p.c2) >> 5, n3 = ceil32(p.c3) >> 5;
  const int nmin = n2 < n3 ? n2 : n3;
  const int ways = nmin >= 3 ? 1 : (nmin == 2 ? 2 : 4);
  // cout-block rounds per wave: 2 for 129-256 couts, 4 for 257-512 (the 1.5M / 7M Point-Transformer configs'
  // 256- / 512-channel layers, backbone_net.py:43-46); the four-round form is instantiated for both layers together
  const int nr = (n2 > 8 || n3 > 8) ? 4 : ((n2 > 4 || n3 > 4) ? 2 : 1);
  const int nr2 = nr == 4 ? 4 : (n2 > 4 ? 2 : 1);
  if (kPrec != 0) {   // the bf16 units: max-from-accumulators epilogue, explicit shapes, at most two cout-block rounds
    const int v2 = n2 >= 3 ? 1 : (n2 == 2 ? 2 : 4), v3 = n3 >= 3 ? 1 : (n3 == 2 ? 2 : 4);
    if (!maxe || nr > 2 || !((v2 == v3) || (v2 == 2 && v3 == 1))) return -1;
  }
  // Tile choice.  Measured on MI355X (re-fitted after the epilogue / k-loop work, which halved what
  // a low residency costs): time per row ~ padding x wave imbalance x (1 + 1.0 / resident workgroups per CU); residency is bounded by LDS (160 KiB, 2 KiB granules),
  // by registers (accumulator tiles + ~70 VGPRs against 512 per SIMD lane) and by 8 workgroups.
  int best_cpw = 0, best_tb = 0;
  double best_cost = 1e30;
  const int cpw_max = 192 / p.K > 0 ? 192 / p.K : 1;
  for (int cpw = 1; cpw <= cpw_max; cpw++) {
    const int tb = (cpw * p.K + 31) / 32;
    if (tb > 6 || (nr == 4 && tb > 2)) continue;
    const size_t lds = lds_bytes(tb, cpw);
    if (lds > 150 * 1024) continue;
    const int tbw = (tb + ways - 1) / ways;
    const int regs = nr * tbw * 16 + 70;
    if (regs > 250) continue;
    int wgs = (int)((160 * 1024) / ((lds + 2047) / 2048 * 2048));
    const int by_regs = 512 / ((regs + 7) / 8 * 8);
    if (by_regs < wgs) wgs = by_regs;
    if (wgs > 8) wgs = 8;
    if (wgs < 1) continue;
    const double pad = (double)(32 * tb) / (double)(cpw * p.K);
    const double imb = (double)(tbw * ways) / (double)tb;
    const double cost = pad * imb * (1.0 + 1.0 / wgs);
    if (cost < best_cost - 1e-9) { best_cost = cost; best_cpw = cpw; best_tb = tb; }
  }
  static const int force_cpw = pcr_tune_int("PCR_SA_CPW");   // tuning aid
  if (force_cpw > 0) {
    const int tb = (force_cpw * p.K + 31) / 32;
    if (tb <= (nr == 4 ? 2 : 6) && lds_bytes(tb, force_cpw) <= 150 * 1024) { best_cpw = force_cpw; best_tb = tb; }
  }
  if (!best_cpw) return -1;
  if (p.D && !p.pq_ready) {
    const int rc = pcr_dense_pm_f32(p.feat, p.wpq, p.pq_ws, p.B, p.D, pqw, p.N, p.feat_point_major, st_);
    if (rc != PCR_OK) return rc == PCR_ERR_INVALID ? -1 : rc;
  }
  Sa2Args a;
  a.B = p.B; a.N = p.N; a.S = p.S; a.K = p.K; a.c1 = p.c1; a.c2 = p.c2; a.c3 = p.c3; a.CPW = best_cpw;
  a.xyz = p.xyz; a.idx = p.idx; a.centre_idx = p.centre_idx; a.wa = p.wa;
  a.pq = p.D ? p.pq_ws : nullptr;
  a.pqw = pqw;
  a.qoff = p.mode == 0 ? p.c1 : -1;
  const int dbg = pcr_tune_int("PCR_SA_DBG");    // (0 in production; a tuning build re-reads it per launch: in-process A/Bs)
  a.dbg = dbg;
  a.claim = nullptr;
  a.xt = 0;
  a.wp2 = wl2; a.wp3 = wl3;
  a.sh1 = p.shift[0]; a.sh2 = p.shift_pad[0]; a.sh3 = p.shift_pad[1];
  a.out = p.out;
  a.out_pm = p.out_point_major;
  a.wap = p.wa_packed;
  {
    static const int no_l1m = pcr_tune_int("PCR_SA_NO_L1M");   // diagnostics
    const int n1 = ceil32(p.c1) >> 5;
    const int w1 = n1 >= 3 ? 1 : (n1 == 2 ? 2 : 4);
    const int w2c = n2 >= 3 ? 1 : (n2 == 2 ? 2 : 4), w3c = n3 >= 3 ? 1 : (n3 == 2 ? 2 : 4);
    // (the kernel deals layer 1's tiles with layer 2's compile-time shape: explicit shapes only, same class)
    // (f32 forms are instantiated for equal classes of layers 2 and 3 only; the bf16 units also hold (2, 1))
    a.l1m = (!no_l1m && p.wa_packed && n1 <= 4 && n2 <= 4 && w1 == w2c && (kPrec != 0 || w2c == w3c) && (p.c1 & 3) == 0) ? 1 : 0;
  }
#if PCR_SA_PREC != 0
  {
    // wave-autonomous form (shape-only choice): equal widths of 32 / 64 / 128, K in whole 16-row groups, at most three
    // 32-row blocks per item (K = 16, 32, 48, 64, 96)
    static const int no_stream = pcr_tune_int("PCR_SA_NO_STREAM");   // diagnostics
    const int ncb = p.c1 >> 5, ncb3 = p.c3 >> 5;
    int nblk_item = 0, ncen_item = 0;
    // (the LARGEST item of whole centres within three blocks: K = 32 / 16 take three blocks = 3 / 6 centres per item, so the
    // per-item work -- the maxima's fold and store, the item walk -- is paid once per 96 rows)
    static const int small_items = pcr_tune_int("PCR_SAS_SMALL_ITEMS");   // diagnostics: the smallest item instead
    for (int nb = 3; nb >= 1; nb--)
      if ((32 * nb) % p.K == 0 && (!nblk_item || small_items)) { nblk_item = nb; ncen_item = 32 * nb / p.K; }
    const size_t fixed = ((size_t)(2 * ncb) * ncb * 128 + (size_t)(2 * ncb) * ncb3 * 128) * 16 + (size_t)(2 * p.c1 + p.c3) * 4 +
                         (size_t)ncb * 64 * 16;
    const size_t lds_s = fixed + (size_t)kSasWaves * 6 * p.c3 * 4 + 16;   // (+ the four MFMA tokens)
    (void)no_stream;
    if (maxe && nblk_item && sas_shape_ok(p, false)) {
      static const int ncu = [] {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess ||
            hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n < 8)
          n = 256;
        return n;
      }();
      const long items = (long)p.B * ((p.S + ncen_item - 1) / ncen_item);
      long wgs = (items + kSasWaves - 1) / kSasWaves;
      const long resident = (long)ncu * ((ncb + ncb3 >= 6 || lds_s > (size_t)80 * 1024) ? 1 : 2);
      if (wgs > resident) wgs = resident;
      wgs = (wgs + 7) / 8 * 8;                         // every XCD gets workgroups (the item order is per XCD)
      const dim3 gg((unsigned)wgs), bb(64 * kSasWaves);
      constexpr bool kLoS = kPrec == 1;
      static const char *ktrace = pcr_tune_str("PCR_SA_TRACE");
      if (ktrace) a.dbg |= 256;
      // claimed items (shape-only: pcr_sa_claim_ws_ints says when; PCR_SA_DBG bit 4096 of a tuning build switches them off)
      a.xt = (p.pq_has_xyz && p.D && p.pq_ready && p.mode == 0) ? 1 : 0;
      if (p.pq_has_xyz && !a.xt) return PCR_ERR_INVALID;
      a.claim = (p.claim_ws && sas_claims(p.c1, p.c3, p.N) && !(a.dbg & 4096)) ? p.claim_ws : nullptr;
      if (a.claim && hipMemsetAsync(a.claim, 0, (size_t)kSasClaimInts * sizeof(int), st) != hipSuccess) return PCR_ERR_LAUNCH;
#define PCR_SAS(NCBv, NCB3v)                                                                  \
  do {                                                                                        \
    static bool ok = allow_big_lds(sa_stream_kernel<NCBv, NCB3v, kLoS>);                      \
    (void)ok;                                                                                 \
    hipLaunchKernelGGL((sa_stream_kernel<NCBv, NCB3v, kLoS>), gg, bb, lds_s, st, a, nblk_item, ncen_item); \
  } while (0)
      if (ncb == 1 && ncb3 == 1) PCR_SAS(1, 1);
      else if (ncb == 1) PCR_SAS(1, 2);
      else if (ncb == 2 && ncb3 == 2) PCR_SAS(2, 2);
      else if (ncb == 2) PCR_SAS(2, 4);
      else PCR_SAS(4, 4);
#undef PCR_SAS
      if (ktrace) rag_dump_trace(ktrace, "krow", (int)(2 * wgs));
      if (hipGetLastError() != hipSuccess) return PCR_ERR_LAUNCH;
      return PCR_OK;
    }
  }
#endif
  if (p.pq_has_xyz) return PCR_ERR_INVALID;   // (tables with the coordinate term: the K-row kernel above is their only reader)
  const size_t lds = lds_bytes(best_tb, best_cpw);
  dim3 grid((p.S + best_cpw - 1) / best_cpw, p.B);
  const int w2 = n2 >= 3 ? 1 : (n2 == 2 ? 2 : 4), w3 = n3 >= 3 ? 1 : (n3 == 2 ? 2 : 4);
  int lrc = 0;
  switch (best_tb) {
    case 1: lrc = sa2_launch_tb<1>(a, nr, nr2, w2, w3, maxe, lds, st, grid); break;
    case 2: lrc = sa2_launch_tb<2>(a, nr, nr2, w2, w3, maxe, lds, st, grid); break;
    case 3: lrc = sa2_launch_tb<3>(a, nr, nr2, w2, w3, maxe, lds, st, grid); break;
    case 4: lrc = sa2_launch_tb<4>(a, nr, nr2, w2, w3, maxe, lds, st, grid); break;
    case 5: lrc = sa2_launch_tb<5>(a, nr, nr2, w2, w3, maxe, lds, st, grid); break;
    default: lrc = sa2_launch_tb<6>(a, nr, nr2, w2, w3, maxe, lds, st, grid); break;
  }
  if (lrc < 0) return -1;
  if (hipGetLastError() != hipSuccess) return PCR_ERR_LAUNCH;
  return PCR_OK;
}

#if PCR_SA_PREC == 1
int pcr_sa2_try_bf3(const pcr_sa_params *p, pcr_stream_t st) { return sa2_try(*p, st); }
#elif PCR_SA_PREC == 2
int pcr_sa2_try_bf1(const pcr_sa_params *p, pcr_stream_t st) { return sa2_try(*p, st); }
#else

// 1: launches of this shape WITHOUT hit counts (all K rows of every group) also run on the tile plan and want the
// workspace of pcr_sa_tile_ws_ints -- the cout-split kernel's shape, sa2_try's `wsplit_shape` (bf16 modes)
// does a ball-query layer with hit counts of this shape read the ball query's row table (pcr_ball_query_rows_f32)?
// = the wave-autonomous ragged form runs it (sas_shape_ok, shape only)
PCR_EXPORT int pcr_sa_uses_row_table(int c1, int c2, int c3, int K, int precision) {
  if (precision == 0 || c1 != c2 || !(c3 == c2 || c3 == 2 * c2) || !(c1 == 32 || c1 == 64 || c1 == 128)) return 0;
  bool kok = false;
  for (int nb = 1; nb <= 3; nb++)
    if ((32 * nb) % K == 0) { kok = (K & 15) == 0; break; }
  if (!kok) return 0;
  const int ncb = c1 >> 5, ncb3 = c3 >> 5;
  {   // (sa2_try asks for a tile-kernel instantiation of the shape before it looks at the wave-autonomous form)
    const int v2 = ncb >= 3 ? 1 : (ncb == 2 ? 2 : 4), v3 = ncb3 >= 3 ? 1 : (ncb3 == 2 ? 2 : 4);
    if (!((v2 == 1 && v3 == 1) || (v2 == 2 && v3 == 1) || (v2 == 2 && v3 == 2) || (v2 == 4 && v3 == 4))) return 0;
  }
  const size_t fixed = ((size_t)(2 * ncb) * ncb * 128 + (size_t)(2 * ncb) * ncb3 * 128) * 16 + (size_t)(2 * c1 + c3) * 4 +
                       (size_t)ncb * 64 * 16;
  const size_t cap = (size_t)160 * 1024;
  return fixed + (size_t)8 * 6 * c3 * 4 <= cap && fixed + (size_t)8 * sas_rag_wave_ints(c3, K, false) * 4 <= cap;
}

PCR_EXPORT int pcr_sa_krow_uses_tiles(int c1, int c2, int c3, int K, int precision) {
  return precision != 0 && c1 == 128 && c2 == 128 && c3 == 256 && K >= 1 && K <= 64;
}

// sa2_try's K-row dispatch for the shapes whose tables may carry the coordinate term (pcr_sa_params.pq_has_xyz)
PCR_EXPORT int pcr_sa_tables_take_xyz(int mode, int D, int c1, int c2, int c3, int K, int precision) {
  static const int no_stream = pcr_tune_int("PCR_SA_NO_STREAM"), no_xt = pcr_tune_int("PCR_SA_NO_XYZ_TABLES");   // diagnostics
  if (no_stream || no_xt || precision == 0 || mode != 0 || D < 1 || c1 != c2 || c2 != c3 || !(c1 == 32 || c1 == 64 || c1 == 128))
    return 0;
  if (K < 16 || (K & 15) || !(32 % K == 0 || 64 % K == 0 || 96 % K == 0)) return 0;
  const int ncb = c1 >> 5;
  const size_t fixed = ((size_t)(2 * ncb) * ncb * 128 * 2) * 16 + (size_t)(3 * c1) * 4 + (size_t)ncb * 64 * 16;
  return fixed + (size_t)8 * 6 * c3 * 4 + 16 <= (size_t)160 * 1024;
}

PCR_EXPORT long pcr_sa_claim_ws_ints(int c1, int c2, int c3, int K, int N, int precision) {
  // the wave-autonomous K-row form's shapes (equal widths, whole 16-row groups) in a bf16 mode, on large clouds
  if (precision == 0 || c1 != c2 || c2 != c3 || (K & 15) || K < 16) return 0;
  return sas_claims(c1, c3, N) ? (long)kSasClaimInts : 0;
}

PCR_EXPORT long pcr_sa_tile_ws_ints(int B, int S, int K, int c2, int c3) {
  if (B < 1 || S < 1 || K < 1 || c2 < 1 || c3 < 1) return 0;
  const int rows = (ceil32(c2) > 128 || ceil32(c3) > 128) ? 64 : 128;   // sa2_try's tile choice
  if (K > rows) return 0;
  const int per_tile = rows / rag_ceil(K);
  const int maxT = (S + per_tile - 1) / per_tile;
  return (long)rag_ws_ints(B, maxT, rows);
}

static int dense_pm_launch(const float *x, const float *wp, float *y, int B, int cin, int cout, int L, int x_point_major,
                           int precision, pcr_stream_t stream, const float *xyz = nullptr, const float *wxyz = nullptr,
                           int q_rows = 0, int q_off = 0) {
  if (!x || !wp || !y || B < 0 || cin < 1 || cout < 1 || cout > 1024 || L < 1 || precision < 0 || precision > 2)
    return PCR_ERR_INVALID;
  if ((xyz != nullptr) != (wxyz != nullptr) || (xyz && (cout & 3))) return PCR_ERR_INVALID;
  if (xyz && (q_rows < 0 || q_rows > L || (q_rows != L && (q_rows & 63)) || q_off < 4 || q_off > cout || (q_off & 3))) return PCR_ERR_INVALID;
  if (cout > 256 && (cout & 3)) return PCR_ERR_INVALID;   // windows of 256 couts keep the 16-byte store path
  if (B == 0) return PCR_OK;
  if (B > 65535) return PCR_ERR_INVALID;
  pcr_note_arith(precision);
  DensePmArgs d{x, wp, y, cin, cout, L, x_point_major, xyz, wxyz, q_rows, q_off};
  {
    // point-major in and out, whole tiles, the weight rows of a wave in registers: the persistent form (shape-only choice)
    static const int no_res = pcr_tune_int("PCR_DENSE_PM_NO_RES");   // diagnostics
    const long ntok = (long)B * L;
    if (!no_res && !xyz && precision != 0 && x_point_major && (cin == 32 || cin == 64 || cin == 128) && (cout == 64 || cout == 128) &&
        ntok % 64 == 0 && (reinterpret_cast<size_t>(x) & 15) == 0 && (reinterpret_cast<size_t>(y) & 15) == 0) {
      static const int ncu = [] {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess ||
            hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n < 1)
          n = 256;
        return n;
      }();
      const long ntile = ntok / 64;
      const long res = (long)ncu * (cin == 128 ? 2 : 4);   // (the kernel's registers: two or four workgroups per CU)
      const long wgs = ntile < res ? ntile : res;
      const size_t lds_r = (size_t)(cin > cout ? cin : cout) * 65 * sizeof(float);
#define PCR_PMR(NSv, NPIv, NPOv)                                                                      \
  hipLaunchKernelGGL((dense_pm_res_kernel<NSv, NPIv, NPOv>), dim3((unsigned)wgs), dim3(kThreads), lds_r, \
                     pcr_s(stream), d, ntile)
#define PCR_PMR_O(NSv, NPIv)                             \
  do {                                                   \
    if (cout == 128) PCR_PMR(NSv, NPIv, 8);              \
    else PCR_PMR(NSv, NPIv, 4);                          \
  } while (0)
#define PCR_PMR_I(NSv)                                   \
  do {                                                   \
    if (cin == 128) PCR_PMR_O(NSv, 8);                   \
    else if (cin == 64) PCR_PMR_O(NSv, 4);               \
    else PCR_PMR_O(NSv, 2);                              \
  } while (0)
      if (precision == 1) PCR_PMR_I(3);
      else PCR_PMR_I(1);
#undef PCR_PMR_I
#undef PCR_PMR_O
#undef PCR_PMR
      PCR_CHECK_LAUNCH();
      return PCR_OK;
    }
  }
  if (xyz && !x_point_major && (cin == 32 || cin == 64) && (cout == 128 || cout == 256) && (L & 63) == 0 &&
      (reinterpret_cast<size_t>(x) & 15) == 0 && (reinterpret_cast<size_t>(y) & 15) == 0 &&
      (reinterpret_cast<size_t>(wxyz) & 15) == 0) {
    // channel-major input, whole tiles per cloud: the persistent form with the coordinate term (shape-only choice)
    static const int no_res = pcr_tune_int("PCR_DENSE_PM_NO_RES");   // diagnostics
    if (!no_res) {
      static const int ncu = [] {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess ||
            hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n < 1)
          n = 256;
        return n;
      }();
      const int tpc = L / 64;
      const long ntile = (long)B * tpc;
      const long res = (long)ncu * (cout == 256 ? 2 : 3);     // (the kernel's registers: two or three workgroups per CU)
      const long wgs = ntile < res ? ntile : res;
      const size_t lds_r = ((size_t)cout * 65 + 3 * 64) * sizeof(float);
#define PCR_PMX(NSv, NPIv, NPOv)                                                                            \
  do {                                                                                                      \
    static bool ok = allow_big_lds(dense_pm_xyz_res_kernel<NSv, NPIv, NPOv>);                               \
    (void)ok;                                                                                               \
    hipLaunchKernelGGL((dense_pm_xyz_res_kernel<NSv, NPIv, NPOv>), dim3((unsigned)wgs), dim3(kThreads), lds_r, \
                       pcr_s(stream), d, tpc, ntile);                                                       \
  } while (0)
#define PCR_PMX_S(NSv)                                      \
  do {                                                      \
    if (cin == 32 && cout == 128) PCR_PMX(NSv, 2, 8);       \
    else if (cin == 32) PCR_PMX(NSv, 2, 16);                \
    else if (cout == 128) PCR_PMX(NSv, 4, 8);               \
    else PCR_PMX(NSv, 4, 16);                               \
  } while (0)
      if (precision == 1) PCR_PMX_S(3);
      else PCR_PMX_S(1);
#undef PCR_PMX_S
#undef PCR_PMX
      PCR_CHECK_LAUNCH();
      return PCR_OK;
    }
  }
  const int wmax = cout < 256 ? cout : 256;
  const int rows = ceil8(cin) > ceil32(wmax) ? ceil8(cin) : ceil32(wmax);
  size_t lds = (size_t)rows * 65 * sizeof(float) + (xyz ? (size_t)(3 * 64 + 4 * wmax) * sizeof(float) : 0);
  if (lds > (size_t)kMaxDynLds) return PCR_ERR_INVALID;
  const dim3 grid((L + 63) / 64, B, (cout + 255) / 256);
#define PCR_PM(NRv, PRv)                                                                          \
  do {                                                                                            \
    static bool ok = allow_big_lds(dense_pm_kernel<NRv, PRv>);                                    \
    (void)ok;                                                                                     \
    hipLaunchKernelGGL((dense_pm_kernel<NRv, PRv>), grid, dim3(kThreads), lds, pcr_s(stream), d); \
  } while (0)
  if (cout > 128) {
    if (precision == 0) PCR_PM(2, 0);
    else if (precision == 1) PCR_PM(2, 1);
    else PCR_PM(2, 2);
  } else {
    if (precision == 0) PCR_PM(1, 0);
    else if (precision == 1) PCR_PM(1, 1);
    else PCR_PM(1, 2);
  }
#undef PCR_PM
  PCR_CHECK_LAUNCH();
  return PCR_OK;
}

PCR_EXPORT int pcr_dense_pm_f32(const float *x, const float *wp, float *y, int B, int cin, int cout, int L,
                                int x_point_major, pcr_stream_t stream) {
  return dense_pm_launch(x, wp, y, B, cin, cout, L, x_point_major, 0, stream);
}

PCR_EXPORT int pcr_dense_pm_prec_f32(const float *x, const float *wp_bf, float *y, int B, int cin, int cout, int L,
                                     int x_point_major, int precision, pcr_stream_t stream) {
  if (precision != 1 && precision != 2) return PCR_ERR_INVALID;
  return dense_pm_launch(x, wp_bf, y, B, cin, cout, L, x_point_major, precision, stream);
}

PCR_EXPORT int pcr_dense_pm_xyz_f32(const float *x, const float *wp_bf, const float *xyz, const float *wxyz, float *y, int B,
                                    int cin, int cout, int L, int x_point_major, int precision, int q_rows, int q_off,
                                    pcr_stream_t stream) {
  if ((precision != 1 && precision != 2) || !xyz || !wxyz) return PCR_ERR_INVALID;
  return dense_pm_launch(x, wp_bf, y, B, cin, cout, L, x_point_major, precision, stream, xyz, wxyz, q_rows, q_off);
}

PCR_EXPORT int pcr_sa_mlp_f32(const pcr_sa_params *pp, pcr_stream_t stream) {
  if (!pp) return PCR_ERR_INVALID;
  const pcr_sa_params &p = *pp;
  if (p.B < 0 || p.N < 1 || p.S < 0 || p.K < 1 || p.D < 0 || p.c1 < 1 || p.c2 < 1 || p.c3 < 1 ||
      !p.xyz || (!p.idx && !(p.row_tab && p.cnt)) || !p.out || (p.D && !p.feat) || (p.mode != 0 && p.mode != 1))
    return PCR_ERR_INVALID;
  for (int l = 0; l < 3; l++)
    if (!p.wp[l] || !p.scale[l] || !p.shift[l]) return PCR_ERR_INVALID;
  if (p.B == 0 || p.S == 0) return PCR_OK;
  if (p.B > 65535) return PCR_ERR_INVALID;
  // tables with the coordinate term (ABI 17): edge mode with features, built by the caller, a shape the query accepts
  if (p.pq_has_xyz && (p.mode != 0 || !p.D || !p.pq_ready || !p.pq_ws ||
                       !pcr_sa_tables_take_xyz(p.mode, p.D, p.c1, p.c2, p.c3, p.K, p.precision)))
    return PCR_ERR_INVALID;
  const int fast = sa2_try(p, stream);
  if (fast >= 0) return fast;
  if (p.pq_has_xyz) return PCR_ERR_INVALID;   // (only the wave-autonomous K-row kernel reads such tables)
  if (!p.idx) return PCR_ERR_INVALID;   // (a row table without an index tensor: only the kernel that reads the table will do)
  pcr_note_arith(PCR_PREC_F32);
  SaArgs a;
  a.p = p;
  a.C0 = 3 + (p.mode == 0 ? 2 * p.D : p.D);
  a.C0P = ceil8(a.C0);
  a.rowsA = a.C0P > ceil8(p.c2) ? a.C0P : ceil8(p.c2);
  a.rowsB = ceil8(p.c1) > p.c3 ? ceil8(p.c1) : p.c3;
  // centres per workgroup: as many as keep rows <= 128 (at least one) and LDS <= 150 KiB
  int cpw = 128 / p.K;
  if (cpw < 1) cpw = 1;
  size_t lds = 0;
  for (;; cpw--) {
    a.CPW = cpw;
    a.TB = (cpw * p.K + 31) / 32;
    a.RP = 32 * a.TB + 1;
    lds = ((size_t)(a.rowsA + a.rowsB) * a.RP + 32 * a.TB) * sizeof(float);
    if (lds <= 150 * 1024 || cpw == 1) break;
  }
  if (lds > (size_t)kMaxDynLds) return PCR_ERR_INVALID;
  static bool ok = allow_big_lds(sa_mlp_kernel);
  (void)ok;
  hipLaunchKernelGGL(sa_mlp_kernel, dim3((p.S + a.CPW - 1) / a.CPW, p.B), dim3(kThreads), lds,
                     pcr_s(stream), a);
  PCR_CHECK_LAUNCH();
  return PCR_OK;
}
#endif   // PCR_SA_PREC == 0 (C-ABI entry points)
