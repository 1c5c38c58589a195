// Training-mode pieces of the grouped set-abstraction layer (models/pointnet2_utils.py:242-288, 333-357) that are
// not plain dense layers: the first MLP layer evaluated from per-point tables (forward and backward) and the max
// over K with its argmax (forward) / the BatchNorm-backward sums of the pooled gradient.
//
// Layer 1 is linear in its input row [dxyz, f_c, f_i - f_c], so y1 = Wa dxyz + P[i] + Q[c] + b with the per-POINT
// tables P = Wf f, Q = (Wc - Wf) f (one dense layer over the N points of the cloud instead of S*K rows); its
// backward sends dy1 back to the tables: dP[i] = sum of dy1 over the rows that gathered point i, dQ[c] = sum over
// the K rows of centre c -- a scatter, done here WITHOUT float atomics: every (channel, point) accumulator has one
// owner thread that adds its rows in row order, so the gradients are bit-reproducible.
#include <type_traits>

#include "tile_dense.h"

namespace {

__host__ __device__ inline int l1_chunk(int N, int c1) {   // channels per workgroup: the chunk's table slice (CS x N) lives in LDS
  int cs = 32;
  while (cs > 1 && (size_t)cs * N * 4 > 32 * 1024) cs >>= 1;
  return cs < c1 ? cs : c1;
}

struct L1Args {
  const float *xyz;        // (B,N,3)
  const int *idx;          // (B,S,K)
  const float *tab;        // (B,2*c1,N): rows [0,c1) = P, rows [c1,2c1) = Q; null when the layer has no point features
  const float *wa, *bias;  // (c1,3), (c1)
  float *y;                // (B,c1,S*K)
  float *stats;            // partials [B * chunks][2][ceil32(CS)]... see l1_fwd: [b][chunk][2][CS]
  int N, S, K, c1, CS;
};

template <int CTRL>
__device__ __forceinline__ float dpp_f32(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float wave_sum_f(float v) {   // four DPP adds inside the 16-lane rows + four v_readlane
  v += dpp_f32<0xB1>(v);
  v += dpp_f32<0x4E>(v);
  v += dpp_f32<0x141>(v);
  v += dpp_f32<0x140>(v);
  const int b = __float_as_int(v);     // (v_readlane moves 32-bit patterns: the builtin is typed int)
  return (__int_as_float(__builtin_amdgcn_readlane(b, 0)) + __int_as_float(__builtin_amdgcn_readlane(b, 16))) +
         (__int_as_float(__builtin_amdgcn_readlane(b, 32)) + __int_as_float(__builtin_amdgcn_readlane(b, 48)));
}

// one workgroup per (cloud, chunk of CS channels); rows r = s*K + k across the lanes (coalesced idx reads / y writes)
template <int CS>
__global__ __launch_bounds__(kThreads) void sa_l1_fwd_kernel(L1Args a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int N = a.N, S = a.S, K = a.K, c1 = a.c1, L = S * K;
  float *Pl = smem;                                   // [CS][N]
  float *Ql = Pl + (a.tab ? CS * N : 0);              // [CS][S]: the centres' columns of Q (a row's 32 loads of it sat
                                                      // in front of its stores: one L2 round trip per row iteration)
  float *xl = Ql + (a.tab ? CS * S : 0);              // [N][3]
  float *wl = xl + 3 * N;                             // [CS][4]: wa, bias
  float *red = wl + 4 * CS;                           // [4 waves][2][CS]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const size_t b = blockIdx.y;
  const int c0 = blockIdx.x * CS;
  const float *tab = a.tab ? a.tab + b * 2 * c1 * N : nullptr;
  if (tab) {   // (batches of eight loads in flight: a load -> LDS-store loop pays the full latency per element)
    constexpr int U = 8;
    for (int e0 = tid; e0 < CS * N; e0 += U * kThreads) {
      float v[U];
#pragma unroll
      for (int u = 0; u < U; u++) {
        const int e = e0 + u * kThreads;
        const int c = e / N, i = e - c * N;
        v[u] = (e < CS * N && c0 + c < c1) ? tab[(size_t)(c0 + c) * N + i] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < U; u++) {
        const int e = e0 + u * kThreads;
        if (e < CS * N) Pl[e] = v[u];
      }
    }
  }
  if (tab) {
    const float *Qg = tab + (size_t)(c1 + c0) * N;
    constexpr int U = 8;
    for (int e0 = tid; e0 < CS * S; e0 += U * kThreads) {
      float v[U];
#pragma unroll
      for (int u = 0; u < U; u++) {
        const int e = e0 + u * kThreads;
        const int c = e / S, s = e - c * S;
        v[u] = (e < CS * S && c0 + c < c1) ? Qg[(size_t)c * N + s] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < U; u++) {
        const int e = e0 + u * kThreads;
        if (e < CS * S) Ql[e] = v[u];
      }
    }
  }
  for (int e = tid; e < 3 * N; e += kThreads) xl[e] = a.xyz[b * N * 3 + e];
  for (int e = tid; e < CS; e += kThreads) {
    const int c = c0 + e;
    const bool ok = c < c1;
    wl[4 * e] = ok ? a.wa[3 * c] : 0.f;
    wl[4 * e + 1] = ok ? a.wa[3 * c + 1] : 0.f;
    wl[4 * e + 2] = ok ? a.wa[3 * c + 2] : 0.f;
    wl[4 * e + 3] = ok && a.bias ? a.bias[c] : 0.f;
  }
  __syncthreads();
  float ssum[CS], ssq[CS];
#pragma unroll
  for (int c = 0; c < CS; c++) ssum[c] = ssq[c] = 0.f;
  const int *idx = a.idx + b * L;
  float *y = a.y + (b * c1 + c0) * L;
  // (two instantiations of the row loop: the table-less first layer must not carry the table's registers and selects)
  auto rows = [&](auto has_tab) {
    constexpr bool TAB = decltype(has_tab)::value;
    int inext = tid < L ? idx[tid] : 0;
    for (int r = tid; r < L; r += kThreads) {
      const int s = r / K, i = inext;
      if (r + kThreads < L) inext = idx[r + kThreads];   // (the next row's index travels while this row is written)
      const float dx = xl[3 * i] - xl[3 * s], dy = xl[3 * i + 1] - xl[3 * s + 1], dz = xl[3 * i + 2] - xl[3 * s + 2];
      // the centre's table column first, all CS loads in flight: read inside the store loop each one waits behind the
      // previous store to y (the compiler cannot rule out that y aliases the table)
#pragma unroll
      for (int c = 0; c < CS; c++) {
        if (c0 + c < c1) {
          float v = fmaf(wl[4 * c + 2], dz, fmaf(wl[4 * c + 1], dy, fmaf(wl[4 * c], dx, wl[4 * c + 3])));
          if constexpr (TAB) v += Pl[c * N + i] + Ql[c * S + s];
          y[(size_t)c * L + r] = v;
          ssum[c] += v;
          ssq[c] += v * v;
        }
      }
    }
  };
  if (tab) rows(std::true_type());
  else rows(std::false_type());
  if (!a.stats) return;
#pragma unroll
  for (int c = 0; c < CS; c++) {
    const float s = wave_sum_f(ssum[c]), q = wave_sum_f(ssq[c]);
    if (lane == 0) {
      red[(wave * 2) * CS + c] = s;
      red[(wave * 2 + 1) * CS + c] = q;
    }
  }
  __syncthreads();
  if (tid < 2 * CS) {
    const int st = tid / CS, c = tid - st * CS;
    const float v = ((red[(0 * 2 + st) * CS + c] + red[(1 * 2 + st) * CS + c]) + red[(2 * 2 + st) * CS + c]) +
                    red[(3 * 2 + st) * CS + c];
    // partial layout [cloud][2][ceil32(c1)]: every chunk of a cloud writes its own channels
    const int CP = ceil32(c1);
    if (c0 + c < c1) a.stats[(b * 2 + st) * CP + c0 + c] = v;
  }
}

#ifndef PCR_L1_TR
#define PCR_L1_TR 64
#endif
constexpr int kL1Tr = PCR_L1_TR;   // 64 or 128 rows per staged tile of sa_l1_bwd_kernel (round 5, measured: 128 halves the barriers
                                    // but needs 193 registers -- 0.29 -> 0.41 ms on the K = 48 layers, 0.183 -> 0.164 on K = 32: 64 stays)

struct L1BwdArgs {
  const float *xyz;
  const int *idx;
  const float *g, *y;          // (B,c1,S*K): masked gradient of the BatchNorm output and the raw layer output
  const float *ka, *kb, *kc;   // BatchNorm backward: dy = ka g + kb y + kc
  float *dtab;                 // (B,2*c1,N) or null
  float *dwa;                  // partials [B][c1][4]: d wa (3), d bias
  int N, S, K, c1, CS;
};

// thread (cl = tid % CS, part = tid / CS): owner of channel cl's accumulators for the points / centres / rows
// congruent to `part` modulo the number of parts; rows are visited in increasing order by every owner
template <int CS>
__global__ __launch_bounds__(kThreads) void sa_l1_bwd_kernel(L1BwdArgs a) {
  constexpr int PARTS = kThreads / CS, TR = kL1Tr;   // rows per staged tile
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int N = a.N, S = a.S, K = a.K, c1 = a.c1, L = S * K;
  const int NP = N | 1, SP = S | 1;            // odd pitches: the lanes of a wave differ in cl, not in i / s
  float *dP = smem;                            // [CS][NP]
  float *dQ = dP + (a.dtab ? CS * NP : 0);     // [CS][SP]
  float *xl = dQ + (a.dtab ? CS * SP : 0);     // [N][3]
  float *dyt = xl + 3 * N;                     // [CS][TR + 1]
  float *dxt = dyt + CS * (TR + 1);            // [TR][3]
  int *it = reinterpret_cast<int *>(dxt + 3 * TR);   // [TR] neighbour index, [TR] centre
  int *st = it + TR;
  float *comb = reinterpret_cast<float *>(st + TR);  // [PARTS][CS][4]
  unsigned long long *msk = reinterpret_cast<unsigned long long *>(      // [PARTS][2] row masks, 8-byte aligned
      (reinterpret_cast<uintptr_t>(comb + PARTS * CS * 4) + 7) & ~(uintptr_t)7);
  const int tid = threadIdx.x;
  const int cl = tid % CS, part = tid / CS;
  const size_t b = blockIdx.y;
  const int c0 = blockIdx.x * CS;
  const bool live = c0 + cl < c1;
  if (a.dtab)
    for (int e = tid; e < CS * NP + CS * SP; e += kThreads) dP[e] = 0.f;   // (dQ follows dP)
  for (int e = tid; e < 3 * N; e += kThreads) xl[e] = a.xyz[b * N * 3 + e];
  const int *idx = a.idx + b * L;
  const float *g = a.g + (b * c1 + c0) * L, *y = a.y + (b * c1 + c0) * L;
  float w0 = 0.f, w1 = 0.f, w2 = 0.f, wb = 0.f;
  // the tile's (channel, row) elements of this thread: e = tid + u * 256, u < NE; the NEXT tile's g / y are fetched into
  // registers before the current tile is scanned, so the global round trip hides behind the scan
  constexpr int NE = (CS * TR + kThreads - 1) / kThreads;
  float pg[NE], py[NE];
  int pi = 0;
  auto fetch = [&](int t0) {
    const int nr = L - t0 < TR ? L - t0 : TR;
    if (tid < nr) pi = idx[t0 + tid];
#pragma unroll
    for (int u = 0; u < NE; u++) {
      const int e = tid + u * kThreads;
      const int c = e / TR, rr = e - c * TR;
      const bool ok = e < CS * TR && rr < nr && c0 + c < c1;
      const size_t o = ok ? (size_t)c * L + t0 + rr : 0;
      pg[u] = g[o];
      py[u] = y[o];
    }
  };
  if (L > 0) fetch(0);
  __syncthreads();
  for (int t0 = 0; t0 < L; t0 += TR) {
    const int nr = L - t0 < TR ? L - t0 : TR;
    if (tid < nr) {
      const int r = t0 + tid, i = pi, s = r / K;
      it[tid] = i;
      st[tid] = s;
      dxt[3 * tid] = xl[3 * i] - xl[3 * s];
      dxt[3 * tid + 1] = xl[3 * i + 1] - xl[3 * s + 1];
      dxt[3 * tid + 2] = xl[3 * i + 2] - xl[3 * s + 2];
    }
#pragma unroll
    for (int u = 0; u < NE; u++) {
      const int e = tid + u * kThreads;
      const int c = e / TR, rr = e - c * TR;
      float v = 0.f;
      if (rr < nr && c0 + c < c1) v = a.ka[c0 + c] * pg[u] + a.kb[c0 + c] * py[u] + a.kc[c0 + c];
      if (e < CS * TR) dyt[c * (TR + 1) + rr] = v;
    }
    // per owner class p: which rows of this tile gather a point congruent to p (one ballot by wave 0): an owner then
    // visits ITS rows only (~1/PARTS of them), in row order
    if (tid < 64) {
#pragma unroll
      for (int w64 = 0; w64 < TR / 64; w64++) {
        const int iv = tid + 64 * w64 < nr ? it[tid + 64 * w64] : -1;
#pragma unroll
        for (int p = 0; p < PARTS; p++) {
          const unsigned long long mp = __ballot(iv >= 0 && (iv % PARTS) == p);
          if (tid == 0) msk[2 * p + w64] = mp;
        }
      }
    }
    __syncthreads();
    if (t0 + TR < L) fetch(t0 + TR);
    if (live) {
      const float *row = dyt + cl * (TR + 1);
      for (int rr = part; rr < nr; rr += PARTS) {
        const float v = row[rr];
        w0 += v * dxt[3 * rr];
        w1 += v * dxt[3 * rr + 1];
        w2 += v * dxt[3 * rr + 2];
        wb += v;
      }
      if (a.dtab) {
#pragma unroll
        for (int w64 = 0; w64 < TR / 64; w64++) {      // (rows in increasing order: word 0, then word 1)
          unsigned long long m = msk[2 * part + w64];
          while (m) {
            const int rr = __builtin_ctzll(m) + 64 * w64;
            m &= m - 1;
            dP[cl * NP + it[rr]] += row[rr];
          }
        }
        // centre sums: the rows of a centre are CONTIGUOUS (K per centre), so the tile holds at most 64 / K + 2 segments;
        // owner class j mod PARTS adds segment j's rows in a register (independent LDS reads) and touches dQ once.
        // (Scattering them row by row like the points above put all K rows of a centre on ONE owner: a chain of K
        // dependent LDS read-modify-writes per tile, ~14 k cycles at K = 48, with the other seven owners idle.)
        const int s_first = st[0], nseg = st[nr - 1] - s_first + 1;
        for (int j = part; j < nseg; j += PARTS) {
          const int s = s_first + j;
          const int lo = s * K - t0 > 0 ? s * K - t0 : 0, hi = (s + 1) * K - t0 < nr ? (s + 1) * K - t0 : nr;
          float acc = 0.f;
          for (int rr = lo; rr < hi; rr++) acc += row[rr];
          dQ[cl * SP + s] += acc;
        }
      }
    }
    __syncthreads();
  }
  comb[(part * CS + cl) * 4] = w0;
  comb[(part * CS + cl) * 4 + 1] = w1;
  comb[(part * CS + cl) * 4 + 2] = w2;
  comb[(part * CS + cl) * 4 + 3] = wb;
  __syncthreads();
  if (tid < CS * 4) {
    const int c = tid >> 2, j = tid & 3;
    float s = 0.f;
    for (int p = 0; p < PARTS; p++) s += comb[(p * CS + c) * 4 + j];
    if (c0 + c < c1) a.dwa[(b * c1 + c0 + c) * 4 + j] = s;
  }
  if (a.dtab) {
    float *dt = a.dtab + b * 2 * c1 * N;
    for (int e = tid; e < CS * N; e += kThreads) {
      const int c = e / N, i = e - c * N;
      if (c0 + c < c1) {
        dt[(size_t)(c0 + c) * N + i] = dP[c * NP + i];
        dt[(size_t)(c1 + c0 + c) * N + i] = i < S ? dQ[c * SP + i] : 0.f;
      }
    }
  }
}

// max over the K rows of every centre of relu(scale * y + shift), with the winning k (first maximum) and the raw y at
// that row (the backward's BatchNorm sums need it: a gather of B C S scattered floats otherwise).
// One workgroup = 32 channels of one cloud, tiles of G = 192 / K centres z, z + gridDim.z, ...: the (32, G K) tile is
// fetched by 16-byte pieces, all of a thread's loads in flight before the first LDS write, then 32 x G threads each
// scan the K rows of one (channel, centre).  grid.z spreads a cloud's tiles over enough workgroups to fill the chip
// (C = 32 is ONE channel block: 512 workgroups of 22 serial tiles before).
struct PoolArgs {
  const float *y;              // (B,C,S*K) raw layer-3 output
  const float *scale, *shift;  // BatchNorm affine of this batch
  float *pooled;               // (B,C,S)
  int *argmax;                 // (B,C,S)
  float *ymax;                 // (B,C,S) or null
  int C, S, K;
};

__global__ __launch_bounds__(kThreads) void sa_pool_fwd_kernel(PoolArgs a) {
  constexpr int CS = 32, kMaxQ = 6;               // 32 channels x <= 48 pieces = 1536 pieces = 6 per thread
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int C = a.C, S = a.S, K = a.K, L = S * K;
  const int G = 192 / K > 0 ? 192 / K : 1;     // centres per staged tile
  const int RP = G * K + 1;
  float *tile = smem;                           // [CS][RP]
  const int tid = threadIdx.x, cl = tid & 31, gi = tid >> 5;
  const size_t b = blockIdx.y;
  const int c0 = blockIdx.x * CS;
  const float *y = a.y + (b * C + c0) * L;
  const float sc = c0 + cl < C ? a.scale[c0 + cl] : 0.f, sh = c0 + cl < C ? a.shift[c0 + cl] : 0.f;
  const bool vec = (K & 3) == 0;
  bool first = true;
  for (int s0 = blockIdx.z * G; s0 < S; s0 += gridDim.z * G, first = false) {
    const int ng = S - s0 < G ? S - s0 : G, nrow = ng * K;
    if (!first) __syncthreads();
    if (vec) {
      const int npc = nrow >> 2, totq = CS * npc;
      f32x4 v[kMaxQ];
#pragma unroll
      for (int u = 0; u < kMaxQ; u++) {
        const int e = tid + u * kThreads;
        const int c = e / npc, q = e - c * npc;
        const bool ok = e < totq && c0 + c < C;
        v[u] = ok ? *reinterpret_cast<const f32x4 *>(y + (size_t)c * L + (size_t)s0 * K + 4 * q) : f32x4{0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int u = 0; u < kMaxQ; u++) {
        const int e = tid + u * kThreads;
        if (e < totq) {
          const int c = e / npc, q = e - c * npc;
          float *d = tile + c * RP + 4 * q;
          d[0] = v[u][0];
          d[1] = v[u][1];
          d[2] = v[u][2];
          d[3] = v[u][3];
        }
      }
    } else {
      // (wave w stages channels w, w + 4, ...: lanes run along the rows, no index division)
      for (int c = tid >> 6; c < CS; c += kThreads / 64) {
        const float *src = y + (size_t)c * L + (size_t)s0 * K;
        const bool okc = c0 + c < C;
        for (int rr = tid & 63; rr < nrow; rr += 64) tile[c * RP + rr] = okc ? src[rr] : 0.f;
      }
    }
    __syncthreads();
    if (c0 + cl < C)
      for (int gc = gi; gc < ng; gc += kThreads / 32) {   // (up to 192 / K centres per tile, eight lanes of centres)
        const float *row = tile + cl * RP + gc * K;
        float best = -INFINITY, raw = 0.f;
        int bk = 0;
        for (int k = 0; k < K; k++) {
          const float r = row[k];
          const float v = fmaxf(r * sc + sh, 0.f);
          if (v > best) {
            best = v;
            bk = k;
            raw = r;
          }
        }
        const size_t o = (b * C + c0 + cl) * S + s0 + gc;
        a.pooled[o] = best;
        a.argmax[o] = bk;
        if (a.ymax) a.ymax[o] = raw;
      }
  }
}

// BatchNorm-backward sums of the pooled gradient: per cloud, S1[c] = sum_s g_eff, S2[c] = sum_s g_eff * ymax[c][s]
// with g_eff = gp where pooled > 0 (ymax = the raw layer-3 output at the winning row, kept by the forward).
// Partials [B][2][ceil32(C)].  One wave per (cloud, channel): lanes over the centres, fixed-order lane sums.
__global__ __launch_bounds__(kThreads) void sa_pool_bwd_stats_kernel(const float *__restrict__ gp,
                                                                     const float *__restrict__ pooled,
                                                                     const float *__restrict__ ymax,
                                                                     float *__restrict__ part, float *__restrict__ gz, int C,
                                                                     int S) {
  const size_t b = blockIdx.y;
  const int CP = ceil32(C);
  const int lane = threadIdx.x & 63, c = blockIdx.x * (kThreads / 64) + (threadIdx.x >> 6);
  if (c >= CP) return;
  float s1 = 0.f, s2 = 0.f;
  if (c < C) {
    const size_t o = (b * C + c) * S;
    for (int s = lane; s < S; s += 64) {
      const float gv = pooled[o + s] > 0.f ? gp[o + s] : 0.f;
      if (gz) gz[o + s] = gv;          // the routed gradient (zero where the ReLU is closed): what dy_mode 3 consumes
      s1 += gv;
      s2 += gv * ymax[o + s];
    }
  }
  s1 = wave_sum_f(s1);
  s2 = wave_sum_f(s2);
  if (lane == 0) {
    part[(b * 2) * CP + c] = s1;
    part[(b * 2 + 1) * CP + c] = s2;
  }
}

}  // namespace

template <class F>
static void l1_dispatch(int cs, F f) {
  switch (cs) {
    case 32: f(std::integral_constant<int, 32>()); break;
    case 16: f(std::integral_constant<int, 16>()); break;
    case 8: f(std::integral_constant<int, 8>()); break;
    case 4: f(std::integral_constant<int, 4>()); break;
    case 2: f(std::integral_constant<int, 2>()); break;
    default: f(std::integral_constant<int, 1>()); break;
  }
}

static int l1_pow2(int cs) {
  int p = 1;
  while (p * 2 <= cs) p *= 2;
  return p;
}

PCR_EXPORT int pcr_sa_l1_fwd_f32(const float *xyz, const int *idx, const float *tab, const float *wa, const float *bias,
                                 float *y, float *stats, int B, int N, int S, int K, int c1, pcr_stream_t stream) {
  if (!xyz || !idx || !wa || !y || B < 0 || N < 1 || S < 1 || S > N || K < 1 || c1 < 1) return PCR_ERR_INVALID;
  if (B == 0) return PCR_OK;
  if (B > 65535) return PCR_ERR_INVALID;
  const int cs = l1_pow2(l1_chunk(N, c1));
  L1Args a{xyz, idx, tab, wa, bias, y, stats, N, S, K, c1, cs};
  const size_t lds = ((tab ? (size_t)cs * (N + S) : 0) + 3 * (size_t)N + 4 * cs + 8 * cs) * sizeof(float);
  if (lds > (size_t)kMaxDynLds) return PCR_ERR_INVALID;
  const dim3 grid((c1 + cs - 1) / cs, B);
  l1_dispatch(cs, [&](auto tag) {
    constexpr int CS = decltype(tag)::value;
    static bool ok = allow_big_lds(sa_l1_fwd_kernel<CS>);
    (void)ok;
    hipLaunchKernelGGL(sa_l1_fwd_kernel<CS>, grid, dim3(kThreads), lds, pcr_s(stream), a);
  });
  PCR_CHECK_LAUNCH();
  return PCR_OK;
}

PCR_EXPORT int pcr_sa_l1_bwd_f32(const float *xyz, const int *idx, const float *g, const float *y, const float *ka,
                                 const float *kb, const float *kc, float *dtab, float *dwa, int B, int N, int S, int K,
                                 int c1, pcr_stream_t stream) {
  if (!xyz || !idx || !g || !y || !ka || !kb || !kc || !dwa || B < 0 || N < 1 || S < 1 || S > N || K < 1 || c1 < 1)
    return PCR_ERR_INVALID;
  if (B == 0) return PCR_OK;
  if (B > 65535) return PCR_ERR_INVALID;
  int cs = l1_pow2(l1_chunk(N, c1));
  if (cs > 32) cs = 32;
  L1BwdArgs a{xyz, idx, g, y, ka, kb, kc, dtab, dwa, N, S, K, c1, cs};
  const int parts = kThreads / cs;
  const size_t lds = ((dtab ? (size_t)cs * ((N | 1) + (S | 1)) : 0) + 3 * (size_t)N + (size_t)cs * (kL1Tr + 1) + 3 * kL1Tr + 2 * kL1Tr +
                      (size_t)parts * cs * 4 + 4 * (size_t)parts + 2) * sizeof(float);
  if (lds > (size_t)kMaxDynLds) return PCR_ERR_INVALID;
  const dim3 grid((c1 + cs - 1) / cs, B);
  l1_dispatch(cs, [&](auto tag) {
    constexpr int CS = decltype(tag)::value;
    static bool ok = allow_big_lds(sa_l1_bwd_kernel<CS>);
    (void)ok;
    hipLaunchKernelGGL(sa_l1_bwd_kernel<CS>, grid, dim3(kThreads), lds, pcr_s(stream), a);
  });
  PCR_CHECK_LAUNCH();
  return PCR_OK;
}

PCR_EXPORT int pcr_sa_pool_fwd_f32(const float *y, const float *scale, const float *shift, float *pooled, int *argmax,
                                   float *ymax, int B, int C, int S, int K, pcr_stream_t stream) {
  if (!y || !scale || !shift || !pooled || !argmax || B < 0 || C < 1 || S < 1 || K < 1 || K > 192) return PCR_ERR_INVALID;
  if (B == 0) return PCR_OK;
  if (B > 65535) return PCR_ERR_INVALID;
  PoolArgs a{y, scale, shift, pooled, argmax, ymax, C, S, K};
  const int G = 192 / K > 0 ? 192 / K : 1;
  const size_t lds = (size_t)32 * (G * K + 1) * sizeof(float);
  const int cb = (C + 31) / 32, tiles = (S + G - 1) / G;
  int gz = (2048 + cb * B - 1) / (cb * B);      // ~2048 workgroups
  gz = gz < 1 ? 1 : (gz > tiles ? tiles : gz);
  hipLaunchKernelGGL(sa_pool_fwd_kernel, dim3(cb, B, gz), dim3(kThreads), lds, pcr_s(stream), a);
  PCR_CHECK_LAUNCH();
  return PCR_OK;
}

PCR_EXPORT int pcr_sa_pool_bwd_stats_f32(const float *gp, const float *pooled, const float *ymax, float *part, float *gz,
                                         int B, int C, int S, pcr_stream_t stream) {
  if (!gp || !pooled || !ymax || !part || B < 0 || C < 1 || S < 1) return PCR_ERR_INVALID;
  if (B == 0) return PCR_OK;
  if (B > 65535) return PCR_ERR_INVALID;
  const int CP = ceil32(C);
  hipLaunchKernelGGL(sa_pool_bwd_stats_kernel, dim3(CP / (kThreads / 64), B), dim3(kThreads), 0, pcr_s(stream), gp, pooled,
                     ymax, part, gz, C, S);
  PCR_CHECK_LAUNCH();
  return PCR_OK;
}
