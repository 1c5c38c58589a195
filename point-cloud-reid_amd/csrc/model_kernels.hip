// Fused model kernels of the siamese ReID hot path for gfx950 (MI355X).
//
// Data layout everywhere: channel-major feature tensors (B, C, L) exactly as the reference's
// model path carries them ([B,C,N] tensors, models/backbone_net.py:96-124), so a tile of 32*TB
// consecutive tokens of one cloud is C contiguous runs in HBM and lands in LDS as [C][RP]
// (RP = tokens + 1, odd => every access pattern used below is bank-conflict free).
//
// All matmuls run on the f32-input matrix core (v_mfma_f32_32x32x2_f32: exact fp32 fmaf chain,
// 64 FLOP/clk/SIMD) with the WEIGHTS as the A operand, read straight from a host-packed image
// (one 16-byte load per lane covers four k-steps), and the LDS-resident activations as the B
// operand (token = lane => conflict-free ds_read_b32, token-contiguous epilogue stores).
#include <math.h>
#include <stdlib.h>

#include "pcr_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

__host__ __device__ inline int ceil8(int x) { return (x + 7) & ~7; }
__host__ __device__ inline int ceil32(int x) { return (x + 31) & ~31; }

constexpr int kThreads = 256;
constexpr int kMaxDynLds = 160 * 1024;

// out[o][t] = epi(sum_k W[o][k] * in[k][t], o, t) for o < OP (multiple of 32), t < 32*TB.
//   in : LDS [CP][RP], CP multiple of 8, rows >= real cin must be ZERO
//   wp : packed image [CP/8][OP][2][4]  (pcr_pack_weight_f32)
// The (cout-block, token-block) tiles are dealt round-robin to the waves, cout-block major, so
// that the waves of a workgroup share weight lines in L1.
template <class Epi>
__device__ __forceinline__ void tile_dense(const float *__restrict__ in, int CP, int RP, int TB,
                                           const float *__restrict__ wp, int OP, Epi epi) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
  const int l31 = lane & 31, h = lane >> 5;
  const int nItems = (OP >> 5) * TB;
  const int KB = CP >> 3;
  const size_t wstride = (size_t)OP * 2;  // f32x4 units per k-block
  for (int item = wave; item < nItems; item += nwaves) {
    const int cb = item / TB, tb = item - cb * TB;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; r++) acc[r] = 0.f;
    const f32x4 *wv = reinterpret_cast<const f32x4 *>(wp) + (size_t)(cb * 32 + l31) * 2 + h;
    const float *bp = in + h * RP + tb * 32 + l31;
#pragma unroll 2
    for (int kb = 0; kb < KB; kb++) {
      const f32x4 a = wv[(size_t)kb * wstride];
      const float *b0 = bp + (kb * 8) * RP;
      const float x0 = b0[0], x1 = b0[2 * RP], x2 = b0[4 * RP], x3 = b0[6 * RP];
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[0], x0, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[1], x1, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[2], x2, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[3], x3, acc, 0, 0, 0);
    }
    const int t = tb * 32 + l31;
#pragma unroll
    for (int r = 0; r < 16; r++) {
      const int o = cb * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
      epi(acc[r], o, t);
    }
  }
}

// Second-generation dense tile: compile-time token tile (RP immediate offsets), each wave OWNS
// cout blocks and sweeps the token blocks with the weight fragment held in registers (one 16-byte
// weight load feeds 4*TB MFMAs), weight fragments prefetched one k-block ahead, and an optional
// barrier between the k-loop and the epilogue so that the output may overwrite the input buffer.
//   nCB = OP/32 >= 3 : wave w owns cout blocks w, w+4 (NR rounds), all TB token blocks
//   nCB == 2         : wave w owns cout block w&1 and token blocks (w>>1), (w>>1)+2, ...
//   nCB == 1         : wave w owns token blocks w, w+4, ...
template <int TB, int NR, int WAYS, class Epi>
__device__ __forceinline__ void tile_dense_impl(const float *__restrict__ in, int CP,
                                                const float *__restrict__ wp, int OP, bool sync_epi,
                                                Epi epi) {
  constexpr int RP = 32 * TB + 1;
  constexpr int TBW = (TB + WAYS - 1) / WAYS;  // token blocks per wave
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int l31 = lane & 31, h = lane >> 5;
  const int nCB = OP >> 5, KB = CP >> 3;
  const int cb0 = WAYS == 1 ? wave : (WAYS == 2 ? (wave & 1) : 0);
  const int tb0 = WAYS == 1 ? 0 : (WAYS == 2 ? (wave >> 1) : wave);
  // The k-loop is branch-free: a tile the wave does not own (cb >= nCB or tb >= TB, which only
  // happens for shapes that do not divide evenly) is computed on clamped addresses and dropped in
  // the epilogue, so the accumulators stay pinned in AGPRs.
  f32x16 acc[NR][TBW];
#pragma unroll
  for (int nr = 0; nr < NR; nr++)
#pragma unroll
    for (int j = 0; j < TBW; j++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[nr][j][r] = 0.f;
  const size_t wstride = (size_t)OP * 2;
  const f32x4 *wrow[NR];
#pragma unroll
  for (int nr = 0; nr < NR; nr++) {
    int cb = cb0 + 4 * nr;
    cb = cb < nCB ? cb : nCB - 1;
    wrow[nr] = reinterpret_cast<const f32x4 *>(wp) + (size_t)cb * 64 + (size_t)l31 * 2 + h;
  }
  const float *brow[TBW];
#pragma unroll
  for (int j = 0; j < TBW; j++) {
    int tb = tb0 + j * WAYS;
    tb = tb < TB ? tb : TB - 1;
    brow[j] = in + h * RP + tb * 32 + l31;
  }
  // two weight-fragment register sets in ping-pong: the 16-byte load for k-block kb+2 is issued
  // right after the last use of set (kb & 1) and has a full block of MFMAs to land
  auto step = [&](const f32x4 (&aw)[NR], int kb) {
#pragma unroll
    for (int j = 0; j < TBW; j++) {
      const float *bt = brow[j] + kb * 8 * RP;
      const float x0 = bt[0], x1 = bt[2 * RP], x2 = bt[4 * RP], x3 = bt[6 * RP];
#pragma unroll
      for (int nr = 0; nr < NR; nr++) {
        acc[nr][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(aw[nr][0], x0, acc[nr][j], 0, 0, 0);
        acc[nr][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(aw[nr][1], x1, acc[nr][j], 0, 0, 0);
        acc[nr][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(aw[nr][2], x2, acc[nr][j], 0, 0, 0);
        acc[nr][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(aw[nr][3], x3, acc[nr][j], 0, 0, 0);
      }
    }
  };
  f32x4 a0[NR], a1[NR];
  const int k1 = KB > 1 ? 1 : 0;
#pragma unroll
  for (int nr = 0; nr < NR; nr++) {
    a0[nr] = wrow[nr][0];
    a1[nr] = wrow[nr][(size_t)k1 * wstride];
  }
  for (int kb = 0; kb < KB; kb += 2) {
    step(a0, kb);
    const int kn0 = kb + 2 < KB ? kb + 2 : KB - 1;
#pragma unroll
    for (int nr = 0; nr < NR; nr++) a0[nr] = wrow[nr][(size_t)kn0 * wstride];
    if (kb + 1 < KB) {
      step(a1, kb + 1);
      const int kn1 = kb + 3 < KB ? kb + 3 : KB - 1;
#pragma unroll
      for (int nr = 0; nr < NR; nr++) a1[nr] = wrow[nr][(size_t)kn1 * wstride];
    }
  }
  if (sync_epi) __syncthreads();
#pragma unroll
  for (int nr = 0; nr < NR; nr++) {
    const int cb = cb0 + 4 * nr;
    if (cb < nCB) {
#pragma unroll
      for (int j = 0; j < TBW; j++) {
        const int tb = tb0 + j * WAYS;
        if (tb < TB) {
          const int t = tb * 32 + l31;
#pragma unroll
          for (int r = 0; r < 16; r++) epi(acc[nr][j][r], cb * 32 + (r & 3) + 8 * (r >> 2) + 4 * h, t);
        }
      }
    }
  }
}

template <int TB, int NR, class Epi>
__device__ __forceinline__ void tile_dense2(const float *__restrict__ in, int CP,
                                            const float *__restrict__ wp, int OP, bool sync_epi, Epi epi) {
  const int nCB = OP >> 5;
  if (nCB >= 3) tile_dense_impl<TB, NR, 1>(in, CP, wp, OP, sync_epi, epi);
  else if (nCB == 2) tile_dense_impl<TB, 1, 2>(in, CP, wp, OP, sync_epi, epi);
  else tile_dense_impl<TB, 1, 4>(in, CP, wp, OP, sync_epi, epi);
}

__device__ __forceinline__ float elu1(float x) { return x > 0.f ? x + 1.0f : (expf(x) - 1.0f) + 1.0f; }

// LayerNorm over the channel rows [0,C) of buf ([C][RP]) for each of the T token columns, in
// place; part = tid / T handles channels part, part+np, ...; partial sums meet in `red`
// ([2][np][T] floats).  Two passes (mean, then centred variance), eps inside the sqrt, affine.
__device__ __forceinline__ void tile_layernorm(float *buf, int C, int RP, int T, const float *g,
                                               const float *bta, float *red) {
  const int tid = threadIdx.x;
  const int np = blockDim.x / T;  // T is 32 or 64 => np = 8 or 4
  const int t = tid % T, part = tid / T;
  float s = 0.f;
  if (part < np)
    for (int c = part; c < C; c += np) s += buf[c * RP + t];
  if (part < np) red[part * T + t] = s;
  __syncthreads();
  float mean = 0.f;
  for (int p = 0; p < np; p++) mean += red[p * T + t];
  mean /= (float)C;
  float v = 0.f;
  if (part < np)
    for (int c = part; c < C; c += np) {
      float d = buf[c * RP + t] - mean;
      v += d * d;
    }
  if (part < np) red[(np + part) * T + t] = v;
  __syncthreads();
  float var = 0.f;
  for (int p = 0; p < np; p++) var += red[(np + p) * T + t];
  var /= (float)C;
  const float inv = 1.0f / sqrtf(var + 1e-5f);
  if (part < np)
    for (int c = part; c < C; c += np) buf[c * RP + t] = (buf[c * RP + t] - mean) * inv * g[c] + bta[c];
  __syncthreads();
}

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
        if (f < D) v = feat[(size_t)f * N + ci];
        else v = feat[(size_t)(f - D) * N + i] - feat[(size_t)(f - D) * N + ci];
      } else {
        v = feat[(size_t)(ch - 3) * N + i];
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
    p.out[(b * c3 + o) * p.S + c0 + c] = m;
  }
}

// ------------------------------------------------- grouped SA MLP, second generation ----
// Layer 1 is linear in its input rows [dxyz, f_c, f_i - f_c] (edge) or [dxyz, f_i] (query-and-
// group), so  W1 row = Wa dxyz + P[i] + Q[c]  with the per-POINT tables P = Wf f, Q = (Wc - Wf) f
// computed once per cloud by dense_pm_kernel (K times fewer FLOPs than per (centre,neighbour)
// row).  The kernel gathers P rows (16-byte loads) straight into the layer-1 activation tile,
// then runs layers 2 and 3 on the matrix core IN PLACE in one LDS buffer and reduces max over K.
struct Sa2Args {
  int B, N, S, K, c1, c2, c3, CPW;
  const float *xyz;
  const int *idx, *centre_idx;
  const float *wa;          // (c1,3) row-major
  const float *pq;          // (B,N,pqw) point-major or null (no features)
  int pqw, qoff;            // row width; offset of Q inside a row, -1 = no Q term
  int dbg;                  // PCR_SA_DBG ablation mask (diagnostics only; 0 in production)
  int skew_div;
  int skew;                 // start-up stagger of the first generation of workgroups, in s_sleep(127) units
  const float *wp2, *wp3;
  const float *sc1, *sh1, *sc2, *sh2, *sc3, *sh3;
  float *out;
};

template <int TB, int NR>
__global__ __launch_bounds__(kThreads) void sa_fused_kernel(Sa2Args a) {
  constexpr int ROWS = 32 * TB, RP = ROWS + 1;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int c1 = a.c1, c2 = a.c2, c3 = a.c3, K = a.K;
  int rowsC = c1 > c3 ? c1 : c3;
  if (ceil8(c2) > rowsC) rowsC = ceil8(c2);
  float *buf = smem;                                        // [rowsC][RP]
  float *sdx = buf + rowsC * RP;                            // [3][ROWS]
  int *sidx = reinterpret_cast<int *>(sdx + 3 * ROWS);      // [ROWS] neighbour, [ROWS] centre point
  int *scen = sidx + ROWS;
  float *sq = reinterpret_cast<float *>(scen + ROWS);     // [CPW][c1] per-centre Q rows + shift
  const int tid = threadIdx.x;
  const size_t b = blockIdx.y;
  const int c0 = blockIdx.x * a.CPW;
  const int nc = (a.S - c0 < a.CPW) ? a.S - c0 : a.CPW;
  const int rows = nc * K;
  const float *xyz = a.xyz + b * a.N * 3;

  if (a.skew) {
    // Identical workgroups started together run their phases in lockstep (all gathering, then all
    // on the matrix core).  Delaying the co-resident workgroups of the FIRST generation by a
    // fraction of a workgroup's lifetime de-phases every later generation too, because each CU slot
    // runs its workgroups back to back.  Pure scheduling heuristic: results do not depend on it.
    const unsigned lin = blockIdx.y * gridDim.x + blockIdx.x;
    if (lin < 256u * 3u) {
      const int n = (int)((lin / (unsigned)a.skew_div) % 3u) * a.skew;
      for (int i = 0; i < n; i++) __builtin_amdgcn_s_sleep(127);
    }
  }
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
          for (int j = 0; j < 4; j++)
            v[j] = fmaxf(w[3 * j] * dx + w[3 * j + 1] * dy + w[3 * j + 2] * dz + p4[u][j] + qr[j], 0.f);
        }
#pragma unroll
        for (int j = 0; j < 4; j++) buf[(o + j) * RP + rw] = v[j];
      }
    }
  }
  }
  __syncthreads();
  if (!(a.dbg & 2)) {
  {
    const float *sc = a.sc2, *sh = a.sh2;
    const int lim = ceil8(c2);
    tile_dense2<TB, NR>(buf, c1, a.wp2, ceil32(c2), true, [&](float v, int o, int t) {
      if (o < lim) buf[o * RP + t] = o < c2 ? fmaxf(v * sc[o] + sh[o], 0.f) : 0.f;
    });
  }
  __syncthreads();
  {
    const float *sc = a.sc3, *sh = a.sh3;
    tile_dense2<TB, NR>(buf, ceil8(c2), a.wp3, ceil32(c3), true, [&](float v, int o, int t) {
      if (o < c3) buf[o * RP + t] = fmaxf(v * sc[o] + sh[o], 0.f);
    });
  }
  }
  __syncthreads();
  if (a.dbg & 4) return;
  for (int e = tid; e < c3 * nc; e += kThreads) {
    const int c = e / c3, o = e - c * c3;
    const float *row = buf + o * RP + c * K;
    float m = row[0];
    for (int k = 1; k < K; k++) m = fmaxf(m, row[k]);
    a.out[(b * c3 + o) * a.S + c0 + c] = m;
  }
}

// y (B,L,cout) POINT-major = W x for x (B,cin,L) channel-major; cout <= 256, no activation.
struct DensePmArgs {
  const float *x, *wp;
  float *y;
  int cin, cout, L;
};

__global__ __launch_bounds__(kThreads) void dense_pm_kernel(DensePmArgs a) {
  constexpr int TB = 2, T = 64, RP = 65;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int cinP = ceil8(a.cin), cout = a.cout;
  float *X = smem;
  float *Y = smem + cinP * RP;
  const size_t b = blockIdx.y;
  const int t0 = blockIdx.x * T;
  for (int e = threadIdx.x; e < cinP * T; e += kThreads) {
    const int c = e / T, t = e - c * T;
    X[c * RP + t] = (c < a.cin && t0 + t < a.L) ? a.x[(b * a.cin + c) * a.L + t0 + t] : 0.f;
  }
  __syncthreads();
  tile_dense2<TB, 2>(X, cinP, a.wp, ceil32(cout), false, [&](float v, int o, int t) {
    if (o < cout) Y[o * RP + t] = v;
  });
  __syncthreads();
  float *out = a.y + (b * a.L + t0) * (size_t)cout;
  for (int e = threadIdx.x; e < cout * T; e += kThreads) {
    const int t = e / cout, c = e - t * cout;
    if (t0 + t < a.L) out[(size_t)t * cout + c] = Y[c * RP + t];
  }
}

// -------------------------------------------------------------- linear attention ----
struct AttnArgs {
  pcr_attn_params p;
};

// loads a [C][T] tile of a (B,C,L) tensor into LDS rows [0,CP), zero beyond C or beyond L
__device__ __forceinline__ void load_tile(float *dst, int RP, const float *src, int C, int CP, int L,
                                          int t0, int T) {
  for (int e = threadIdx.x; e < CP * T; e += blockDim.x) {
    const int c = e / T, t = e - c * T;
    dst[c * RP + t] = (c < C && t0 + t < L) ? src[(size_t)c * L + t0 + t] : 0.f;
  }
}

// xyz (L,3) rows t0.. -> LDS [8][RP] (rows 3..7 zero)
__device__ __forceinline__ void load_xyz_tile(float *dst, int RP, const float *xyz, int L, int t0, int T) {
  for (int e = threadIdx.x; e < 8 * T; e += blockDim.x) {
    const int c = e / T, t = e - c * T;
    dst[c * RP + t] = (c < 3 && t0 + t < L) ? xyz[(size_t)(t0 + t) * 3 + c] : 0.f;
  }
}

// Algebra used by both kernels (the host folds it into the weights, see AttnPlan in
// pcr_amd/engine.py): with h = relu(W0 xyz + b0) the position encoding is W2 h + b2, so
//   Wq (x + W2 h + b2) = [Wq | Wq W2] [x ; h] + Wq b2        (one dense instead of three)
//   [K ; V] pre-activation = [[Wk | kpos Wk W2] ; [Wv | Wv W2]] [x ; h] + [kpos Wk b2 ; Wv b2]
// and the merge projection is folded into the per-cloud KV matrix by the kv kernel:
//   merge(msg)[o] = sum_dd M[o][dd] Q'[dd],  M[o][dd] = sum_{v in head(dd)} Wm[o][v] KV[dd][v],
//   Q'[dd] = Q[dd] * Sk / (Q_head . ksum_head + 1e-6).

// hidden = relu(W0 xyz + b0) for the T tokens of a tile -> dst rows [0,d) ([d][RP]); zero xyz beyond L
__device__ __forceinline__ void pos_hidden(float *dst, int RP, const float *P, const float *w0,
                                           const float *b0, int d, int T) {
  for (int e = threadIdx.x; e < d * T; e += blockDim.x) {
    const int o = e / T, t = e - o * T;
    const float v = w0[o * 3] * P[t] + w0[o * 3 + 1] * P[RP + t] + w0[o * 3 + 2] * P[2 * RP + t] + b0[o];
    dst[o * RP + t] = fmaxf(v, 0.f);
  }
}

__device__ __forceinline__ void load_xyz3(float *P, int RP, const float *xyz, int L, int t0, int T) {
  for (int e = threadIdx.x; e < 3 * T; e += blockDim.x) {
    const int c = e / T, t = e - c * T;
    P[c * RP + t] = t0 + t < L ? xyz[(size_t)(t0 + t) * 3 + c] : 0.f;
  }
}

// One workgroup per key-side cloud, token tiles of T = 32*TB (TB = 2 for d = 32, else 1: the fused
// K/V projection has 2d/32 >= 4 cout blocks, one per wave).
// kv image per cloud: packed (d x d) matrix M (merge folded in, see above) followed by ksum[d].
// LDS: XH [c2 + d] key features ; hidden, KB [d], VB [d], P [3]; after the loop KVl [d][d+1].
template <int TB>
__global__ __launch_bounds__(kThreads) void attn_kv_kernel(AttnArgs a) {
  constexpr int T = 32 * TB, RP = T + 1;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const pcr_attn_params &p = a.p;
  const int d = p.d, c2 = p.c2;
  float *XH = smem;
  float *KB = XH + (c2 + d) * RP;
  float *VB = KB + d * RP;
  float *P = VB + d * RP;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, h = lane >> 5;
  const size_t b = blockIdx.x;
  const float *feat = p.feat_k + b * c2 * p.Sk;
  const float *xyz = p.xyz_k + b * p.Sk * 3;
  const int nb = d >> 5, nT = nb * nb;
  const int dh = d / p.nhead;
  const float sk = (float)p.Sk;
  const float *bkv = p.bkv;

  f32x16 acc[4];
#pragma unroll
  for (int i = 0; i < 4; i++)
#pragma unroll
    for (int r = 0; r < 16; r++) acc[i][r] = 0.f;
  float ksum = 0.f;

  for (int t0 = 0; t0 < p.Sk; t0 += T) {
    const int valid = p.Sk - t0;
    load_tile(XH, RP, feat, c2, c2, p.Sk, t0, T);
    load_xyz3(P, RP, xyz, p.Sk, t0, T);
    __syncthreads();
    pos_hidden(XH + c2 * RP, RP, P, p.pos0_w, p.pos0_b, d, T);
    __syncthreads();
    tile_dense2<TB, 2>(XH, c2 + d, p.wkv, 2 * d, false, [&](float v, int o, int t) {
      if (o < d) KB[o * RP + t] = t < valid ? elu1(v + bkv[o]) : 0.f;
      else VB[(o - d) * RP + t] = t < valid ? (v + bkv[o]) / sk : 0.f;
    });
    __syncthreads();
    if (tid < d) {
      const float *row = KB + tid * RP;
      float s = 0.f;
      for (int t = 0; t < T; t++) s += row[t];
      ksum += s;
    }
#pragma unroll
    for (int it = 0; it < 4; it++) {
      const int item = wave + 4 * it;
      if (item < nT) {
        const int ib = item / nb, jb = item - ib * nb;
        const float *ap = KB + (ib * 32 + l31) * RP + h;
        const float *bp = VB + (jb * 32 + l31) * RP + h;
#pragma unroll 4
        for (int ks = 0; ks < T / 2; ks++)
          acc[it] = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[2 * ks], bp[2 * ks], acc[it], 0, 0, 0);
      }
    }
    // no barrier here: the next tile only rewrites XH/P before its first barrier, KB/VB after it
  }
  __syncthreads();
  // KV (head-masked) -> LDS [dd][d+1], then fold the merge projection and write the packed image
  float *KVl = smem;
  const int ld = d + 1;
#pragma unroll
  for (int it = 0; it < 4; it++) {
    const int item = wave + 4 * it;
    if (item < nT) {
      const int ib = item / nb, jb = item - ib * nb;
      const int v = jb * 32 + l31;
#pragma unroll
      for (int r = 0; r < 16; r++) {
        const int dd = ib * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        KVl[dd * ld + v] = (dd / dh == v / dh) ? acc[it][r] : 0.f;
      }
    }
  }
  __syncthreads();
  float *kv = p.kv + b * ((size_t)d * d + d);
  for (int e = tid; e < d * d; e += kThreads) {
    const int o = e / d, dd = e - o * d;
    const int v0 = (dd / dh) * dh;
    const float *wm = p.wmerge + (size_t)o * d + v0;
    const float *kr = KVl + dd * ld + v0;
    float m = 0.f;
    for (int v = 0; v < dh; v++) m += wm[v] * kr[v];
    const int kb = dd >> 3, rem = dd & 7;
    kv[(((size_t)kb * d + o) * 2 + (rem & 1)) * 4 + (rem >> 1)] = m;
  }
  if (tid < d) kv[(size_t)d * d + tid] = ksum;
}

// One workgroup per (query cloud, tile of T query tokens), T = 128 / 64 / 32 for d = 32 / 64 / 128.
// LDS: CAT [c1 + d (pad 8)]: rows [0,c1) query features, rows [c1,c1+d) position hidden -> later
// the merged message; W [max(2d,cout,cfinal)] working buffer; P [3]; zs [nhead]; red.
template <int TB>
__global__ __launch_bounds__(kThreads) void attn_apply_kernel(AttnArgs a) {
  constexpr int T = 32 * TB, RP = T + 1;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const pcr_attn_params &p = a.p;
  const int d = p.d, c1 = p.c1, cout = p.cout;
  const int catC = c1 + d, catP = ceil8(catC);
  int rowsW = 2 * d;
  if (cout > rowsW) rowsW = cout;
  if (p.cfinal > rowsW) rowsW = p.cfinal;
  float *CAT = smem;
  float *W = CAT + catP * RP;
  float *P = W + rowsW * RP;
  float *zs = P + 3 * RP;
  float *red = zs + p.nhead * RP;  // [2 * (256/T)][T]
  const int tid = threadIdx.x;
  const size_t b = blockIdx.y;
  const int t0 = blockIdx.x * T;
  const float *feat = p.feat_q + b * c1 * p.Lq;
  const size_t kb_ = p.kv_index ? (size_t)p.kv_index[b] : b;
  const float *kv = p.kv + kb_ * ((size_t)d * d + d);
  const float *ksum = kv + (size_t)d * d;
  const int dh = d / p.nhead;

  load_tile(CAT, RP, feat, c1, c1, p.Lq, t0, T);
  if (p.q_pos) {
    load_xyz3(P, RP, p.xyz_q + b * p.Lq * 3, p.Lq, t0, T);
    __syncthreads();
    pos_hidden(CAT + c1 * RP, RP, P, p.pos0_w, p.pos0_b, d, T);
    for (int e = tid; e < (catP - catC) * T; e += kThreads) CAT[(catC + e / T) * RP + e % T] = 0.f;
  } else {
    for (int e = tid; e < (catP - c1) * T; e += kThreads) CAT[(c1 + e / T) * RP + e % T] = 0.f;
  }
  __syncthreads();
  {  // Q = elu(Wq' [x ; h] + bq) + 1
    const float *bq = p.bq;
    tile_dense2<TB, 2>(CAT, p.q_pos ? catP : ceil8(c1), p.wq, d, false,
                       [&](float v, int o, int t) { W[o * RP + t] = elu1(v + bq[o]); });
  }
  __syncthreads();
  for (int e = tid; e < p.nhead * T; e += kThreads) {
    const int hd = e / T, t = e - hd * T;
    float z = 0.f;
    for (int c = 0; c < dh; c++) z += W[(hd * dh + c) * RP + t] * ksum[hd * dh + c];
    zs[hd * RP + t] = (1.0f / (z + 1e-6f)) * (float)p.Sk;
  }
  __syncthreads();
  for (int e = tid; e < d * T; e += kThreads) {
    const int o = e / T, t = e - o * T;
    W[o * RP + t] *= zs[(o / dh) * RP + t];
  }
  __syncthreads();
  tile_dense2<TB, 2>(W, d, kv, d, false, [&](float v, int o, int t) { CAT[(c1 + o) * RP + t] = v; });
  __syncthreads();
  tile_layernorm(CAT + c1 * RP, d, RP, T, p.ln1_g, p.ln1_b, red);
  tile_dense2<TB, 2>(CAT, catP, p.wmlp0, 2 * d, false, [&](float v, int o, int t) { W[o * RP + t] = fmaxf(v, 0.f); });
  __syncthreads();
  tile_dense2<TB, 2>(W, 2 * d, p.wmlp2, ceil32(cout), true, [&](float v, int o, int t) {
    if (o < cout) W[o * RP + t] = v;
  });
  __syncthreads();
  tile_layernorm(W, cout, RP, T, p.ln2_g, p.ln2_b, red);
  if (p.residual) {
    for (int e = tid; e < cout * T; e += kThreads) {
      const int c = e / T, t = e - c * T;
      W[c * RP + t] = CAT[c * RP + t] + W[c * RP + t];
    }
    __syncthreads();
  }
  int cres = cout;
  if (p.cfinal) {  // trailing 1x1 conv with bias (cov_final); needs cout % 8 == 0
    const float *bf = p.bfinal;
    const int cf = p.cfinal;
    tile_dense2<TB, 2>(W, cout, p.wfinal, ceil32(cf), true, [&](float v, int o, int t) {
      if (o < cf) W[o * RP + t] = v + bf[o];
    });
    __syncthreads();
    cres = cf;
  }
  float *out = p.out + b * cres * p.Lq;
  for (int e = tid; e < cres * T; e += kThreads) {
    const int c = e / T, t = e - c * T;
    if (t0 + t < p.Lq) out[(size_t)c * p.Lq + t0 + t] = W[c * RP + t];
  }
}

// ------------------------------------------------------------------ pool + head ----
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v = fmaxf(v, __shfl_xor(v, m, 64));
  return v;
}

// GroupNorm of a length-n vector in LDS (one sample), groups of n/g channels; in place.
__device__ __forceinline__ void vec_groupnorm(float *v, int n, int g, const float *gamma, const float *beta) {
  const int gs = n / g;
  const int tid = threadIdx.x;
  float out = 0.f;
  if (tid < n) {
    const int g0 = (tid / gs) * gs;
    float m = 0.f;
    for (int i = 0; i < gs; i++) m += v[g0 + i];
    m /= (float)gs;
    float var = 0.f;
    for (int i = 0; i < gs; i++) { float dd = v[g0 + i] - m; var += dd * dd; }
    var /= (float)gs;
    out = (v[tid] - m) * (1.0f / sqrtf(var + 1e-5f)) * gamma[tid] + beta[tid];
  }
  __syncthreads();
  if (tid < n) v[tid] = out;
  __syncthreads();
}

__global__ __launch_bounds__(kThreads) void pool_head_kernel(pcr_head_params p) {
  __shared__ float x[256], y[256], z[256];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const size_t pr = blockIdx.x;
  const int C = p.C, L = p.L, n = 2 * C;
  const float *o1 = p.o + pr * C * L;
  const float *o2 = p.o + (pr + p.P) * C * L;
  for (int c = wave; c < C; c += kThreads / 64) {
    float mx = -INFINITY, sm = 0.f;
    for (int i = lane; i < L; i += 64) {
      const float a = o1[(size_t)c * L + i], bq = o2[(size_t)c * L + i];
      mx = fmaxf(mx, fmaxf(a, bq));
      sm += a + bq;
    }
    mx = wave_max(mx);
    sm = wave_sum(sm);
    if (lane == 0) { x[c] = mx; x[C + c] = sm / (float)(2 * L); }
  }
  __syncthreads();
  if (p.pooled && tid < n) p.pooled[pr * n + tid] = x[tid];
  if (tid < n) {
    const float *w = p.w1 + (size_t)tid * n;
    float s = 0.f;
    for (int i = 0; i < n; i++) s += w[i] * x[i];
    y[tid] = s;
  }
  __syncthreads();
  vec_groupnorm(y, n, p.groups, p.gn1_g, p.gn1_b);
  if (tid < n) y[tid] = fmaxf(y[tid], 0.f);
  __syncthreads();
  if (tid < n) {
    const float *w = p.w2 + (size_t)tid * n;
    float s = 0.f;
    for (int i = 0; i < n; i++) s += w[i] * y[i];
    z[tid] = s;
  }
  __syncthreads();
  vec_groupnorm(z, n, p.groups, p.gn2_g, p.gn2_b);
  float part = 0.f;
  if (tid < n) part = fmaxf(z[tid] + x[tid], 0.f) * p.w_out[tid];
  part = wave_sum(part);
  if (lane == 0) y[wave] = part;  // y is free after the second matvec
  __syncthreads();
  if (tid == 0) {
    float s = 0.f;
    for (int w = 0; w < kThreads / 64; w++) s += y[w];
    p.logits[pr] = s + p.b_out[0];
  }
}

// pool 'both' of a (B,C,L) tensor: out (B,2C) = [max over L, mean over L]
__global__ __launch_bounds__(kThreads) void pool_both_kernel(const float *__restrict__ x,
                                                             float *__restrict__ out, int C, int L) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const size_t b = blockIdx.x;
  const float *xb = x + b * C * L;
  for (int c = wave; c < C; c += kThreads / 64) {
    float mx = -INFINITY, sm = 0.f;
    for (int i = lane; i < L; i += 64) {
      const float a = xb[(size_t)c * L + i];
      mx = fmaxf(mx, a);
      sm += a;
    }
    mx = wave_max(mx);
    sm = wave_sum(sm);
    if (lane == 0) { out[b * 2 * C + c] = mx; out[b * 2 * C + C + c] = sm / (float)L; }
  }
}

// ------------------------------------------------------------------- generic dense ----
struct DenseArgs {
  const float *x, *wp, *scale, *shift;
  float *y;
  int cin, cout, L, act, TB, RP;
};

__global__ __launch_bounds__(kThreads) void dense_kernel(DenseArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int RP = a.RP, T = 32 * a.TB;
  const int cinP = ceil8(a.cin), coutP = ceil32(a.cout);
  float *X = smem;
  float *Y = smem + cinP * RP;
  const size_t b = blockIdx.y;
  const int t0 = blockIdx.x * T;
  load_tile(X, RP, a.x + b * a.cin * a.L, a.cin, cinP, a.L, t0, T);
  __syncthreads();
  const float *sc = a.scale, *sh = a.shift;
  const int cout = a.cout, act = a.act;
  tile_dense(X, cinP, RP, a.TB, a.wp, coutP, [&](float v, int o, int t) {
    if (o < cout) {
      float r = v * (sc ? sc[o] : 1.0f) + (sh ? sh[o] : 0.0f);
      Y[o * RP + t] = act ? fmaxf(r, 0.f) : r;
    }
  });
  __syncthreads();
  float *out = a.y + b * a.cout * a.L;
  for (int e = threadIdx.x; e < a.cout * T; e += kThreads) {
    const int c = e / T, t = e - c * T;
    if (t0 + t < a.L) out[(size_t)c * a.L + t0 + t] = Y[c * RP + t];
  }
}

template <class Kern>
bool allow_big_lds(Kern k) {
  return hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize,
                             kMaxDynLds) == hipSuccess;
}

}  // namespace

// ------------------------------------------------------------------------------ C ABI ----
PCR_EXPORT long pcr_packed_weight_floats(int cout, int cin) {
  if (cout < 1 || cin < 1) return 0;
  return (long)ceil8(cin) * ceil32(cout);
}

PCR_EXPORT int pcr_pack_weight_f32(const float *w, int cout, int cin, float *packed) {
  if (!w || !packed || cout < 1 || cin < 1) return PCR_ERR_INVALID;
  const int CP = ceil8(cin), OP = ceil32(cout);
  for (int kb = 0; kb < CP / 8; kb++)
    for (int o = 0; o < OP; o++)
      for (int h = 0; h < 2; h++)
        for (int j = 0; j < 4; j++) {
          const int k = kb * 8 + j * 2 + h;
          packed[(((size_t)kb * OP + o) * 2 + h) * 4 + j] = (o < cout && k < cin) ? w[(size_t)o * cin + k] : 0.f;
        }
  return PCR_OK;
}

PCR_EXPORT long pcr_attn_kv_floats(int d) { return (long)d * d + d; }

template <int TB>
static int sa2_launch_tb(const Sa2Args &a, int nr, size_t lds, hipStream_t st, dim3 grid) {
  if (nr == 1) {
    static bool ok = allow_big_lds(sa_fused_kernel<TB, 1>);
    (void)ok;
    hipLaunchKernelGGL((sa_fused_kernel<TB, 1>), grid, dim3(kThreads), lds, st, a);
  } else {
    static bool ok = allow_big_lds(sa_fused_kernel<TB, 2>);
    (void)ok;
    hipLaunchKernelGGL((sa_fused_kernel<TB, 2>), grid, dim3(kThreads), lds, st, a);
  }
  return 0;
}

// fast path; returns -1 when the configuration is not covered (caller falls back to sa_mlp_kernel)
static int sa2_try(const pcr_sa_params &p, hipStream_t st) {
  if (!p.wa || (p.D && (!p.wpq || !p.pq_ws))) return -1;
  if ((p.c1 & 7) || p.c1 > 256 || p.c2 > 256 || p.c3 > 256) return -1;
  const int pqw = p.mode == 0 ? 2 * p.c1 : p.c1;
  if (p.D && pqw > 256) return -1;
  int rowsC = p.c1 > p.c3 ? p.c1 : p.c3;
  if (ceil8(p.c2) > rowsC) rowsC = ceil8(p.c2);
  const int n2 = ceil32(p.c2) >> 5, n3 = ceil32(p.c3) >> 5;
  const int nmin = n2 < n3 ? n2 : n3;
  const int ways = nmin >= 3 ? 1 : (nmin == 2 ? 2 : 4);
  const int nr = (n2 > 4 || n3 > 4) ? 2 : 1;
  int best_cpw = 0, best_tb = 0;
  for (int pass = 0; pass < 2 && !best_cpw; pass++) {
    // pass 0: token-block count divisible among the waves and >= 2 workgroups per CU; pass 1: anything that fits
    for (int cpw = 192 / p.K > 0 ? 192 / p.K : 1; cpw >= 1; cpw--) {
      const int tb = (cpw * p.K + 31) / 32;
      if (tb > 6) continue;
      const size_t lds = ((size_t)rowsC * (32 * tb + 1) + 5 * 32 * tb + (size_t)cpw * p.c1) * sizeof(float);
      if (pass == 0 && (tb % ways || lds > 80 * 1024)) continue;
      if (lds > 150 * 1024) continue;
      best_cpw = cpw;
      best_tb = tb;
      break;
    }
  }
  if (!best_cpw) return -1;
  if (p.D) {
    DensePmArgs d{p.feat, p.wpq, p.pq_ws, p.D, pqw, p.N};
    size_t lds = ((size_t)(ceil8(p.D) + pqw) * 65) * sizeof(float);
    if (lds > (size_t)kMaxDynLds) return -1;
    static bool ok = allow_big_lds(dense_pm_kernel);
    (void)ok;
    hipLaunchKernelGGL(dense_pm_kernel, dim3((p.N + 63) / 64, p.B), dim3(kThreads), lds, st, d);
    if (hipGetLastError() != hipSuccess) return PCR_ERR_LAUNCH;
  }
  Sa2Args a;
  a.B = p.B; a.N = p.N; a.S = p.S; a.K = p.K; a.c1 = p.c1; a.c2 = p.c2; a.c3 = p.c3; a.CPW = best_cpw;
  a.xyz = p.xyz; a.idx = p.idx; a.centre_idx = p.centre_idx; a.wa = p.wa;
  a.pq = p.D ? p.pq_ws : nullptr;
  a.pqw = pqw;
  a.qoff = p.mode == 0 ? p.c1 : -1;
  static const int dbg = getenv("PCR_SA_DBG") ? atoi(getenv("PCR_SA_DBG")) : 0;
  a.dbg = dbg;
  static const int skew = getenv("PCR_SA_SKEW") ? atoi(getenv("PCR_SA_SKEW")) : 0;
  a.skew = skew;
  static const int skew_div = getenv("PCR_SA_SKEW_DIV") ? atoi(getenv("PCR_SA_SKEW_DIV")) : 256;
  a.skew_div = skew_div > 0 ? skew_div : 256;
  a.wp2 = p.wp[1]; a.wp3 = p.wp[2];
  a.sc1 = p.scale[0]; a.sh1 = p.shift[0]; a.sc2 = p.scale[1]; a.sh2 = p.shift[1];
  a.sc3 = p.scale[2]; a.sh3 = p.shift[2];
  a.out = p.out;
  const size_t lds = ((size_t)rowsC * (32 * best_tb + 1) + 5 * 32 * best_tb + (size_t)best_cpw * p.c1) * sizeof(float);
  dim3 grid((p.S + best_cpw - 1) / best_cpw, p.B);
  switch (best_tb) {
    case 1: sa2_launch_tb<1>(a, nr, lds, st, grid); break;
    case 2: sa2_launch_tb<2>(a, nr, lds, st, grid); break;
    case 3: sa2_launch_tb<3>(a, nr, lds, st, grid); break;
    case 4: sa2_launch_tb<4>(a, nr, lds, st, grid); break;
    case 5: sa2_launch_tb<5>(a, nr, lds, st, grid); break;
    default: sa2_launch_tb<6>(a, nr, lds, st, grid); break;
  }
  if (hipGetLastError() != hipSuccess) return PCR_ERR_LAUNCH;
  return PCR_OK;
}

PCR_EXPORT int pcr_sa_mlp_f32(const pcr_sa_params *pp, pcr_stream_t stream) {
  if (!pp) return PCR_ERR_INVALID;
  const pcr_sa_params &p = *pp;
  if (p.B < 0 || p.N < 1 || p.S < 0 || p.K < 1 || p.D < 0 || p.c1 < 1 || p.c2 < 1 || p.c3 < 1 ||
      !p.xyz || !p.idx || !p.out || (p.D && !p.feat) || (p.mode != 0 && p.mode != 1))
    return PCR_ERR_INVALID;
  for (int l = 0; l < 3; l++)
    if (!p.wp[l] || !p.scale[l] || !p.shift[l]) return PCR_ERR_INVALID;
  if (p.B == 0 || p.S == 0) return PCR_OK;
  if (p.B > 65535) return PCR_ERR_INVALID;
  const int fast = sa2_try(p, pcr_s(stream));
  if (fast >= 0) return fast;
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

static int attn_check(const pcr_attn_params &p) {
  if (p.B < 0 || p.Lq < 1 || p.Sk < 1 || p.c1 < 1 || p.c2 < 1 || p.cout < 1 || p.nhead < 1) return 1;
  if (p.d < 32 || p.d > 128 || (p.d & 31) || p.d % p.nhead) return 1;  // d_model in {32,64,96,128}
  if ((p.c2 & 7) || p.cout > 256 || p.cfinal > 256) return 1;
  if (!p.feat_q || !p.feat_k || !p.xyz_k || !p.kv || !p.pos0_w || !p.pos0_b || !p.wq || !p.bq || !p.wkv ||
      !p.bkv || !p.wmerge || !p.wmlp0 || !p.wmlp2 || !p.ln1_g || !p.ln1_b || !p.ln2_g || !p.ln2_b)
    return 1;
  if (p.q_pos && (!p.xyz_q || p.c1 != p.c2 || p.c1 != p.d)) return 1;
  if (p.residual && p.cout != p.c1) return 1;
  if (p.cfinal && (!p.wfinal || !p.bfinal || (p.cout & 7))) return 1;
  return 0;
}

PCR_EXPORT int pcr_attn_kv_f32(const pcr_attn_params *pp, pcr_stream_t stream) {
  if (!pp || attn_check(*pp)) return PCR_ERR_INVALID;
  if (pp->B == 0) return PCR_OK;
  AttnArgs a;
  a.p = *pp;
  const int d = pp->d;
  const int tb = d <= 32 ? 2 : 1, RP = 32 * tb + 1;
  size_t lds = ((size_t)(pp->c2 + 3 * d + 3) * RP) * sizeof(float);
  const size_t lds2 = (size_t)d * (d + 1) * sizeof(float);
  if (lds2 > lds) lds = lds2;
  if (lds > (size_t)kMaxDynLds) return PCR_ERR_INVALID;
  static bool ok = allow_big_lds(attn_kv_kernel<1>) && allow_big_lds(attn_kv_kernel<2>);
  (void)ok;
  dim3 g(pp->B), blk(kThreads);
  hipStream_t st = pcr_s(stream);
  if (tb == 2) hipLaunchKernelGGL(attn_kv_kernel<2>, g, blk, lds, st, a);
  else hipLaunchKernelGGL(attn_kv_kernel<1>, g, blk, lds, st, a);
  PCR_CHECK_LAUNCH();
  return PCR_OK;
}

PCR_EXPORT int pcr_attn_apply_f32(const pcr_attn_params *pp, pcr_stream_t stream) {
  if (!pp || attn_check(*pp) || !pp->out) return PCR_ERR_INVALID;
  if (pp->B == 0) return PCR_OK;
  if (pp->B > 65535) return PCR_ERR_INVALID;
  const pcr_attn_params &p = *pp;
  AttnArgs a;
  a.p = p;
  const int tb = p.d <= 32 ? 4 : (p.d <= 64 ? 2 : 1), T = 32 * tb, RP = T + 1;
  const int catP = ceil8(p.c1 + p.d);
  int rowsW = 2 * p.d;
  if (p.cout > rowsW) rowsW = p.cout;
  if (p.cfinal > rowsW) rowsW = p.cfinal;
  size_t lds = ((size_t)(catP + rowsW + 3 + p.nhead) * RP + 2 * (kThreads / T) * T) * sizeof(float);
  if (lds > (size_t)kMaxDynLds) return PCR_ERR_INVALID;
  static bool ok = allow_big_lds(attn_apply_kernel<1>) && allow_big_lds(attn_apply_kernel<2>) &&
                   allow_big_lds(attn_apply_kernel<4>);
  (void)ok;
  dim3 g((p.Lq + T - 1) / T, p.B), blk(kThreads);
  hipStream_t st = pcr_s(stream);
  if (tb == 4) hipLaunchKernelGGL(attn_apply_kernel<4>, g, blk, lds, st, a);
  else if (tb == 2) hipLaunchKernelGGL(attn_apply_kernel<2>, g, blk, lds, st, a);
  else hipLaunchKernelGGL(attn_apply_kernel<1>, g, blk, lds, st, a);
  PCR_CHECK_LAUNCH();
  return PCR_OK;
}

PCR_EXPORT int pcr_pool_head_f32(const pcr_head_params *pp, pcr_stream_t stream) {
  if (!pp) return PCR_ERR_INVALID;
  const pcr_head_params &p = *pp;
  if (p.P < 0 || p.C < 1 || 2 * p.C > 256 || p.L < 1 || p.groups < 1 || (2 * p.C) % p.groups || !p.o ||
      !p.w1 || !p.w2 || !p.gn1_g || !p.gn1_b || !p.gn2_g || !p.gn2_b || !p.w_out || !p.b_out || !p.logits)
    return PCR_ERR_INVALID;
  if (p.P == 0) return PCR_OK;
  hipLaunchKernelGGL(pool_head_kernel, dim3(p.P), dim3(kThreads), 0, pcr_s(stream), p);
  PCR_CHECK_LAUNCH();
  return PCR_OK;
}

PCR_EXPORT int pcr_pool_both_f32(const float *x, float *out, int B, int C, int L, pcr_stream_t stream) {
  if (!x || !out || B < 0 || C < 1 || L < 1) return PCR_ERR_INVALID;
  if (B == 0) return PCR_OK;
  hipLaunchKernelGGL(pool_both_kernel, dim3(B), dim3(kThreads), 0, pcr_s(stream), x, out, C, L);
  PCR_CHECK_LAUNCH();
  return PCR_OK;
}

PCR_EXPORT int pcr_dense_f32(const float *x, const float *wp, const float *scale, const float *shift,
                             float *y, int B, int cin, int cout, int L, int act, pcr_stream_t stream) {
  if (!x || !wp || !y || B < 0 || cin < 1 || cout < 1 || L < 1) return PCR_ERR_INVALID;
  if (B == 0) return PCR_OK;
  if (B > 65535) return PCR_ERR_INVALID;
  DenseArgs a{x, wp, scale, shift, y, cin, cout, L, act, 1, 33};
  size_t lds = ((size_t)(ceil8(cin) + cout) * a.RP) * sizeof(float);
  if (lds > (size_t)kMaxDynLds) return PCR_ERR_INVALID;
  static bool ok = allow_big_lds(dense_kernel);
  (void)ok;
  hipLaunchKernelGGL(dense_kernel, dim3((L + 31) / 32, B), dim3(kThreads), lds, pcr_s(stream), a);
  PCR_CHECK_LAUNCH();
  return PCR_OK;
}
