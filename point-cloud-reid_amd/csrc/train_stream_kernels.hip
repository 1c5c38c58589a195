// Wave-autonomous ("streaming") forms of the train-dense launches for the narrow grouped-MLP layers (32 / 64 channels
// in and out, L a multiple of 32: layers 2 / 3 of the first two set-abstraction modules in training mode, reference
// models/pointnet2_utils.py:333-357 with BatchNorm2d batch statistics).
//
// Why a second form: the workgroup-cooperative kernels of train_kernels.hip walk a 64-token tile through
// load -> LDS commit -> barrier -> matrix -> barrier -> statistics / store, and on a 32-channel layer those phases of the
// workgroups sharing a CU simply add up (1.4 - 2.3 TB/s of algorithmic traffic on a part that copies at 5.5 TB/s).
// Here a WAVE owns a 32-token block end to end and there is no workgroup barrier inside the loop:
//   * the block's operands are loaded straight into the MFMA B-operand layout (lane = token, register = channel pair
//     2s + h: one 128-byte line per lane half and instruction), the NEXT block's loads are in flight while this one is
//     computed, the weights (<= 64 x 64) live in registers for the whole launch;
//   * BatchNorm affine + ReLU / the BatchNorm backward are applied in registers, the output leaves from the accumulator
//     layout (again one 128-byte line per lane half and instruction);
//   * the backward's dW contracts over the TOKENS, i.e. needs dy and f(x) with lane = channel: the two 32 x 32 tiles
//     are transposed through a wave-private LDS strip (no barrier: a wave's LDS operations complete in order);
//   * per-channel sums stay per-lane partials across all of a wave's blocks and are folded once at the end.
// Waves therefore drift apart and the memory, matrix and store phases of the waves on a SIMD overlap by themselves.
// Results: same MFMA products in the same k order as the tile kernels (y / dx bit-identical), partial sums in a
// different but fixed order (per wave, then waves 0..3 of a workgroup, then pcr_reduce_parts_f32).
#include <type_traits>

#include "tile_dense.h"
#include "train_stream.h"

namespace {

typedef __amdgpu_buffer_rsrc_t rsrc_t;

__device__ __forceinline__ rsrc_t ts_rsrc(const void *p, size_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, (int)(bytes > 0xFFFFFFFFull ? 0xFFFFFFFFull : bytes),
                                           0x00020000);
}
__device__ __forceinline__ float ts_ld(rsrc_t r, int voff, int soff) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
}
__device__ __forceinline__ int ts_ldi(rsrc_t r, int voff, int soff) {
  return (int)__builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0);
}
__device__ __forceinline__ void ts_st(rsrc_t r, float v, int voff, int soff) {
  __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, voff, soff, 0);
}

template <int CTRL>
__device__ __forceinline__ float ts_dpp(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
// sum over the 32 lanes of each wave half; the result is valid in lanes 0 and 32 (all lanes of rows 0 / 2 in fact)
__device__ __forceinline__ float ts_half_sum(float v) {
  v += ts_dpp<0xB1>(v);
  v += ts_dpp<0x4E>(v);
  v += ts_dpp<0x141>(v);
  v += ts_dpp<0x140>(v);                               // every lane: the sum of its 16-lane row
  v += __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(v), 0x401F));   // xor 16: rows 0+1, 2+3
  return v;
}

__device__ __forceinline__ void ts_wave_sync() {       // orders a wave's own LDS writes before its later LDS reads
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

constexpr int kTsWaves = 4;

// Fused max over the K rows of every centre (the grouped MLP's pooling, reference pointnet2_utils.py:357: the max of
// relu(BatchNorm(y)) over K) inside the LAST layer's forward.  relu(scale y + shift) is monotone in y -- increasing for
// gamma >= 0, decreasing otherwise, and gamma is known before the batch statistics are -- so the winner of a centre is
// the max (min) of the RAW output, which the wave has in its strip: no second pass over the (B, C, S K) tensor.
// A wave's blocks are consecutive and its range starts on a centre boundary (host: per is a multiple of
// lcm(32, K) / 32 blocks), so a centre's running winner is carried in registers from block to block.
// t: the lane's 16 tokens (2 q + h) of channel ch; T0: first token of the block inside its cloud; K >= 32.
// (the carried winner as three scalars: an array of structs handed down by reference ended up in scratch memory)
// ry / ra: buffer descriptors of pool_ymax / pool_arg, voff: byte offset of the lane's (cloud, channel) row
__device__ __forceinline__ void ts_pool_block(const float (&t)[16], float sgn, int h, int T0, int K, float &cy_best,
                                              float &cy_raw, int &cy_k, rsrc_t ry, rsrc_t ra, int voff) {
  const int cA = T0 / K, k0 = T0 - cA * K, Tb = K - k0;      // tokens [0, Tb) of the block belong to centre cA
  float bA = -INFINITY, rA = 0.f, bB = -INFINITY, rB = 0.f;
  int kA = 0, kB = 0;
  // K is even (host), T0 a multiple of 32: Tb is even, so BOTH tokens 2 q, 2 q + 1 of a step belong to the same centre and
  // the A / B choice is a wave-uniform branch per step, not two predicated updates
  const int qs = __builtin_amdgcn_readfirstlane(Tb >> 1);
#pragma unroll
  for (int q = 0; q < 16; q++) {
    const int tau = 2 * q + h;
    const float v = sgn * t[q];
    if (q < qs) {
      if (v > bA) { bA = v; rA = t[q]; kA = k0 + tau; }
    } else {
      if (v > bB) { bB = v; rB = t[q]; kB = tau - Tb; }
    }
  }
  auto comb = [&](float &b, float &r, int &k) __attribute__((always_inline)) {      // the other token parity (lane half): larger value, then smaller k
    const float ob = __shfl_xor(b, 32, 64), orr = __shfl_xor(r, 32, 64);
    const int ok = __shfl_xor(k, 32, 64);
    const bool take = ob > b || (ob == b && ok < k);
    b = take ? ob : b;
    r = take ? orr : r;
    k = take ? ok : k;
  };
  comb(bA, rA, kA);
  comb(bB, rB, kB);
  if (bA > cy_best) {       // (the carried part holds the centre's earlier rows: it wins ties)
    cy_best = bA;
    cy_raw = rA;
    cy_k = kA;
  }
  if (Tb <= 32) {           // centre cA ends inside this block
    const int vo = h == 0 ? voff : 0x7FFFFF00;      // (one lane half writes; the other's store is dropped by the range check)
    ts_st(ry, cy_raw, vo, cA * 4);
    __builtin_amdgcn_raw_buffer_store_b32((unsigned)cy_k, ra, vo, cA * 4, 0);
    cy_best = bB;           // centre cA + 1 (nothing yet: -inf)
    cy_raw = rB;
    cy_k = kB;
  }
}

// ---------------------------------------------------------------------------------- forward ----
struct TSFwd {
  const float *x;            // (B, 32 NBI, L)
  const float *isc, *ish;    // input affine (previous BatchNorm) or null
  int in_relu;
  const float *wp;           // packed image of W (pcr_pack_weight_f32 layout: [cin / 8][coutP][2][4])
  const float *bias;         // zero-padded to coutP, or null
  float *y;                  // (B, 32 NBO, L)
  float *stats;              // partials [gridDim.x][2][32 NBO] or null
  int B, L, nblk, per;       // 32-token blocks in all, blocks per wave
  int poolK;                 // > 0: fused max over the K rows of every centre (L = S K)
  const float *pool_gamma;   // (cout): its sign picks max or min of the raw output
  float *pool_ymax;          // (B, cout, S) raw output at the winning row
  int *pool_arg;             // (B, cout, S) winning row
};

// WLDS: the weights as the A operand come from an LDS image [k = input channel][output channel] (one ds_read_b32 per
// MFMA: 12 % of the LDS pipe at full matrix rate) instead of registers -- layers wider than 32 x 32, up to 128 x 128.
template <int NBI, int NBO, bool WLDS>
__global__ __launch_bounds__(64 * kTsWaves) void tstream_fwd_kernel(TSFwd a) {
  constexpr int CIN = 32 * NBI, COUT = 32 * NBO, KS = CIN / 2, TP = 33;
  extern __shared__ __attribute__((aligned(16))) float ts_smem[];
  f32x2 *s_aff = reinterpret_cast<f32x2 *>(ts_smem);            // [CIN] (scale, shift) of the input affine
  float *s_bias = ts_smem + 2 * CIN;                            // [COUT]
  float *s_w = s_bias + COUT;                                   // [CIN][COUT] (WLDS)
  float *s_t = s_w + (WLDS ? CIN * COUT : 0);                   // [waves][COUT * TP]: transposition strip (statistics)
  const int tid = threadIdx.x, lane = tid & 63, j = lane & 31, h = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int L = a.L, nbpc = L >> 5;
  const bool aff = a.isc != nullptr;
  for (int e = tid; e < CIN; e += 64 * kTsWaves) s_aff[e] = f32x2{aff ? a.isc[e] : 1.f, aff ? a.ish[e] : 0.f};
  for (int e = tid; e < COUT; e += 64 * kTsWaves) s_bias[e] = a.bias ? a.bias[e] : 0.f;
  const float lo = (aff && a.in_relu) ? 0.f : -INFINITY;     // relu as max(v, lo): no branch in the k-loop
  // weights: lane (i = j, h) needs W[32 nbo + i][2 s + h] for every k-step s.
  // packed image: element ((kb * COUT + o) * 2 + hh) * 4 + q = W[o][8 kb + 2 q + hh]
  float aw[WLDS ? 1 : NBO][WLDS ? 1 : KS];
  if constexpr (WLDS) {
    for (int e = tid; e < CIN * COUT; e += 64 * kTsWaves) {
      const int q = e & 3, hh = (e >> 2) & 1, o = (e >> 3) % COUT, kb = (e >> 3) / COUT;
      s_w[(8 * kb + 2 * q + hh) * COUT + o] = a.wp[e];
    }
  } else {
#pragma unroll
    for (int nbo = 0; nbo < NBO; nbo++) {
      const f32x4 *wv = reinterpret_cast<const f32x4 *>(a.wp) + (size_t)(nbo * 32 + j) * 2 + h;
#pragma unroll
      for (int kb = 0; kb < CIN / 8; kb++) {
        const f32x4 w4 = wv[(size_t)kb * COUT * 2];
#pragma unroll
        for (int q = 0; q < 4; q++) aw[nbo][4 * kb + q] = w4[q];
      }
    }
  }
  // per-lane partial sums with lane (i, h) = channel 32 nbo + i, its tokens of parity h
  float ssum[NBO], ssq[NBO];
#pragma unroll
  for (int nbo = 0; nbo < NBO; nbo++) ssum[nbo] = ssq[nbo] = 0.f;
  __syncthreads();
  // (every load of the prologue has landed before the loop: its waits would otherwise sit INSIDE the loop, as vmcnt(0),
  // and drain the block prefetch with them)
  __builtin_amdgcn_s_waitcnt(0);

  const rsrc_t rx = ts_rsrc(a.x, (size_t)a.B * CIN * L * 4), ry = ts_rsrc(a.y, (size_t)a.B * COUT * L * 4);
  const int pS = a.poolK > 0 ? L / a.poolK : 1;
  const rsrc_t rpy = ts_rsrc(a.poolK > 0 ? (const void *)a.pool_ymax : (const void *)a.y, a.poolK > 0 ? (size_t)a.B * COUT * pS * 4 : 0);
  const rsrc_t rpa = ts_rsrc(a.poolK > 0 ? (const void *)a.pool_arg : (const void *)a.y, a.poolK > 0 ? (size_t)a.B * COUT * pS * 4 : 0);
  const int gw = blockIdx.x * kTsWaves + wave;
  const int n0 = gw * a.per;
  const int n1 = n0 + a.per < a.nblk ? n0 + a.per : a.nblk;
  const int vx = (h * L + j) * 4;          // lane part of the B-operand address: channel 2 s + h, token j
  const int vy = (4 * h * L + j) * 4;      // lane part of the accumulator address: channel 8 g + 4 h + q, token j
  float *strip = s_t + wave * (COUT * TP);
  const bool want_stats = a.stats != nullptr;
  const bool pool = a.poolK > 0;
  float psgn[NBO], pbest[NBO], praw[NBO];
  int pk[NBO];
#pragma unroll
  for (int nbo = 0; nbo < NBO; nbo++) {
    // (gamma == 0: every row gives the same activation; sign 0 makes the FIRST row the winner, as the pooling kernel does)
    const float gmm = pool ? a.pool_gamma[nbo * 32 + j] : 1.f;
    psgn[nbo] = gmm < 0.f ? -1.f : (gmm > 0.f ? 1.f : 0.f);
    pbest[nbo] = -INFINITY;
    praw[nbo] = 0.f;
    pk[nbo] = 0;
  }
  // The loop body has NO conditional memory operation (the wait counts stay exact): every wave runs a.per (even) rounds,
  // a round beyond the wave's range re-reads its last block and its stores are dropped by the buffer range check.
  auto load = [&](float (&xr)[KS], int bb, int tt) __attribute__((always_inline)) {
    const int so = (bb * CIN * L + tt * 32) * 4;
#pragma unroll
    for (int s = 0; s < KS; s++) xr[s] = ts_ld(rx, vx, so + s * 2 * L * 4);
  };
  auto round = [&](float (&xr)[KS], int bb, int tt, bool valid) __attribute__((always_inline)) {
    f32x16 acc[NBO];
#pragma unroll
    for (int nbo = 0; nbo < NBO; nbo++)
#pragma unroll
      for (int g4 = 0; g4 < 4; g4++) {
        const f32x4 b4 = *reinterpret_cast<const f32x4 *>(s_bias + nbo * 32 + 8 * g4 + 4 * h);
#pragma unroll
        for (int q = 0; q < 4; q++) acc[nbo][4 * g4 + q] = b4[q];
      }
    // LDS operands (weight fragments, affine pairs) are requested PD steps ahead of the MFMAs that use them, in a ring
    // of register sets; the scheduling barriers pin that order (left alone, hipcc issues each ds_read right before its
    // use and waits for it: with one wave per SIMD that latency is the matrix pipe's idle time)
    constexpr int PD = 2;
    float wr[PD + 1][NBO];
    f32x2 ar[PD + 1];
    auto fetch_k = [&](int slot, int s) __attribute__((always_inline)) {
      ar[slot] = s_aff[2 * s + h];
      if constexpr (WLDS) {
#pragma unroll
        for (int nbo = 0; nbo < NBO; nbo++) wr[slot][nbo] = s_w[(2 * s + h) * COUT + nbo * 32 + j];
      }
    };
#pragma unroll
    for (int s = 0; s < PD; s++) fetch_k(s, s);
#pragma unroll
    for (int s = 0; s < KS; s++) {
      if (s + PD < KS) fetch_k((s + PD) % (PD + 1), s + PD);
      __builtin_amdgcn_sched_barrier(0);
      const f32x2 sc = ar[s % (PD + 1)];
      const float v = fmaxf(fmaf(xr[s], sc[0], sc[1]), lo);
#pragma unroll
      for (int nbo = 0; nbo < NBO; nbo++) {
        const float wv = WLDS ? wr[s % (PD + 1)][nbo] : aw[WLDS ? 0 : nbo][WLDS ? 0 : s];
        acc[nbo] = __builtin_amdgcn_mfma_f32_32x32x2f32(wv, v, acc[nbo], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    const int so = (bb * COUT * L + tt * 32) * 4;
    const int vo = valid ? vy : 0x7FFFFF00;
#pragma unroll
    for (int nbo = 0; nbo < NBO; nbo++)
#pragma unroll
      for (int r = 0; r < 16; r++) {
        ts_st(ry, acc[nbo][r], vo, so + (nbo * 32 + (r & 3) + 8 * (r >> 2)) * L * 4);
        strip[(nbo * 32 + (r & 3) + 8 * (r >> 2) + 4 * h) * TP + j] = valid ? acc[nbo][r] : 0.f;
      }
    ts_wave_sync();
    if (want_stats || pool) {
#pragma unroll
      for (int nbo = 0; nbo < NBO; nbo++) {
        float t[16];
#pragma unroll
        for (int s = 0; s < 16; s++) t[s] = strip[(nbo * 32 + j) * TP + 2 * s + h];
#pragma unroll
        for (int s = 0; s < 16; s++) {
          ssum[nbo] += t[s];
          ssq[nbo] += t[s] * t[s];
        }
        if (pool && valid) {
          ts_pool_block(t, psgn[nbo], h, tt * 32, a.poolK, pbest[nbo], praw[nbo], pk[nbo], rpy, rpa,
                        ((bb * COUT + nbo * 32 + j) * pS) * 4);
        }
      }
    }
    ts_wave_sync();
  };
  auto next = [&](int &bb, int &tt, bool go) __attribute__((always_inline)) {      // scalar only
    const int t2 = tt + 1;
    const bool wrap = t2 == nbpc;
    const int nb = wrap ? bb + 1 : bb, nt = wrap ? 0 : t2;
    bb = go ? nb : bb;
    tt = go ? nt : tt;
  };
  const int nc = n0 < a.nblk ? n0 : a.nblk - 1;
  int bA = nc / nbpc, tA = nc - bA * nbpc;
  float xa[KS], xb[KS];
  load(xa, bA, tA);
  // (the other register set is requested BEFORE this set's k-loop: its round trip hides behind the matrix phase)
  for (int it = 0, idx = n0; it < a.per; it += 2, idx += 2) {
    int bB = bA, tB = tA;
    next(bB, tB, idx + 1 < n1);
    load(xb, bB, tB);
    round(xa, bA, tA, idx < n1);
    int bC = bB, tC = tB;
    next(bC, tC, idx + 2 < n1);
    load(xa, bC, tC);
    round(xb, bB, tB, idx + 1 < n1);
    bA = bC;
    tA = tC;
  }
  if (!want_stats) return;
  // fold: the two token parities of a channel, then the workgroup's waves in order
  __syncthreads();
  float *red = s_t;                                 // [waves][2][COUT]
#pragma unroll
  for (int nbo = 0; nbo < NBO; nbo++) {
    const float d1 = ssum[nbo] + __shfl_xor(ssum[nbo], 32, 64), d2 = ssq[nbo] + __shfl_xor(ssq[nbo], 32, 64);
    if (h == 0) {
      red[(wave * 2) * COUT + nbo * 32 + j] = d1;
      red[(wave * 2 + 1) * COUT + nbo * 32 + j] = d2;
    }
  }
  __syncthreads();
  for (int e = tid; e < 2 * COUT; e += 64 * kTsWaves) {
    const int st = e / COUT, c = e - st * COUT;
    float v = red[st * COUT + c];
#pragma unroll
    for (int w = 1; w < kTsWaves; w++) v += red[(w * 2 + st) * COUT + c];
    a.stats[((size_t)blockIdx.x * 2 + st) * COUT + c] = v;
  }
}

// The 128 x 128 forward (layers 2 / 3 of the third set-abstraction module) is bound by the f32 matrix pipe, not by HBM,
// and its register sets allow ONE wave per SIMD: nothing overlaps that wave's epilogue with its own k-loop unless the
// instruction stream does.  This form software-pipelines the blocks INSIDE the wave: k-step s of block i carries, in the
// shadow of its four MFMAs, item s of block i - 1's epilogue (one output row: store + a copy into the statistics strip)
// and item s of block i + 1's operand fetch (one B-operand register).  64 k-steps, 64 output rows, 64 operand
// registers: the mapping is one to one and the memory operations are spread evenly over the round.
// Everything the k-loop touches besides the accumulators stays in the ARCHITECTURAL registers: an accumulator-file
// register read (v_accvgpr_read) while MFMAs are in flight waits for the matrix pipe to drain -- with the previous
// block's rows or spilled operands living there, every k-step took twice its MFMA time (measured: 0.337 ms against
// 0.15 ms for the bare k-loops).  So the finished block is copied out of the accumulators in one burst at the end of its
// round, and the per-channel sums are taken lane = channel from the strip (8 registers) instead of per accumulator row
// (128).  Same MFMA products in the same order as the plain form: y is bit-identical.
// EIGHT waves per workgroup (two per SIMD, one workgroup per CU, all sharing the LDS weight image): a wave can have 63
// memory operations in flight (vmcnt is six bits), 256 bytes each in these layouts, and with four waves per CU that
// window -- 16 MB over the chip against ~6 us of loaded memory latency -- capped the launch at 2.5 TB/s, half of what
// the matrix pipe asks for.  The statistics strip holds ONE cout block (32 channels x 32 tokens, 4.2 KB per wave): the
// 16 items of a cout block are consecutive k-steps, the lane = channel sums follow them.
constexpr int kTpWaves = 8;

// POOL: the fused max over K (ts_pool_block) -- compiled out of the eight-wave instantiation, whose 256 registers per
// lane are spoken for (measured with four waves and 512: the pooling's ~180 VALU instructions per cout block cost the
// matrix-bound launch more than the separate pooling pass over y it saves; the 32- / 64-channel forms keep it)
template <int NB, bool POOL>
__global__ __launch_bounds__(64 * kTpWaves, 1) void tstream_fwd_pipe_kernel(TSFwd a) {
  constexpr int C = 32 * NB, KS = C / 2, TP = 33;
  static_assert(KS == 16 * NB, "one epilogue item per k-step");
  extern __shared__ __attribute__((aligned(16))) float ts_smem[];
  f32x2 *s_aff = reinterpret_cast<f32x2 *>(ts_smem);            // [C] (scale, shift) of the input affine
  float *s_bias = ts_smem + 2 * C;                              // [C]
  float *s_w = s_bias + C;                                      // [k = input channel][i = 32 lanes][cout block]: ONE 16-byte read
                                                                // per lane and k-step, its address an immediate offset
  float *s_t = s_w + C * C;                                     // [waves][32 * TP]: statistics strip; at the end the fold
  const int tid = threadIdx.x, lane = tid & 63, j = lane & 31, h = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int L = a.L, nbpc = L >> 5;
  const bool aff = a.isc != nullptr;
  for (int e = tid; e < C; e += 64 * kTpWaves) {
    s_aff[e] = f32x2{aff ? a.isc[e] : 1.f, aff ? a.ish[e] : 0.f};
    s_bias[e] = a.bias ? a.bias[e] : 0.f;
  }
  const float lo = (aff && a.in_relu) ? 0.f : -INFINITY;
  for (int e = tid; e < C * C; e += 64 * kTpWaves) {
    const int q = e & 3, hh = (e >> 2) & 1, o = (e >> 3) % C, kb = (e >> 3) / C;
    s_w[((8 * kb + 2 * q + hh) * 32 + (o & 31)) * NB + (o >> 5)] = a.wp[e];
  }
  float ssum[NB], ssq[NB];               // per-lane partials, lane (i, h) = channel 32 nbo + i, its tokens of parity h
#pragma unroll
  for (int nbo = 0; nbo < NB; nbo++) ssum[nbo] = ssq[nbo] = 0.f;
  __syncthreads();
  __builtin_amdgcn_s_waitcnt(0);

  const rsrc_t rx = ts_rsrc(a.x, (size_t)a.B * C * L * 4), ry = ts_rsrc(a.y, (size_t)a.B * C * L * 4);
  const int pS = a.poolK > 0 ? L / a.poolK : 1;
  const rsrc_t rpy = ts_rsrc(a.poolK > 0 ? (const void *)a.pool_ymax : (const void *)a.y, a.poolK > 0 ? (size_t)a.B * C * pS * 4 : 0);
  const rsrc_t rpa = ts_rsrc(a.poolK > 0 ? (const void *)a.pool_arg : (const void *)a.y, a.poolK > 0 ? (size_t)a.B * C * pS * 4 : 0);
  const int gw = blockIdx.x * kTpWaves + wave;
  const int n0 = gw * a.per;
  const int n1 = n0 + a.per < a.nblk ? n0 + a.per : a.nblk;
  const int vx = (h * L + j) * 4, vy = (4 * h * L + j) * 4;
  float *strip = s_t + wave * (32 * TP);
  const bool pool = POOL && a.poolK > 0;
  float psgn[NB], pbest[NB], praw[NB];
  int pk[NB];
#pragma unroll
  for (int nbo = 0; nbo < NB; nbo++) {
    // (gamma == 0: every row gives the same activation; sign 0 makes the FIRST row the winner, as the pooling kernel does)
    const float gmm = pool ? a.pool_gamma[nbo * 32 + j] : 1.f;
    psgn[nbo] = gmm < 0.f ? -1.f : (gmm > 0.f ? 1.f : 0.f);
    pbest[nbo] = -INFINITY;
    praw[nbo] = 0.f;
    pk[nbo] = 0;
  }
  float pv[KS];                          // the previous block's output rows (row s = cout block s >> 4, accumulator row s & 15)
#pragma unroll
  for (int s = 0; s < KS; s++) pv[s] = 0.f;
  // xc: this block's operands; xp: the previous block's operand set, refilled with the NEXT block's
  auto round = [&](float (&xc)[KS], float (&xp)[KS], int pb, int pt, bool validp, int nb, int nt) __attribute__((always_inline)) {
    f32x16 ac[NB];
#pragma unroll
    for (int nbo = 0; nbo < NB; nbo++)
#pragma unroll
      for (int g4 = 0; g4 < 4; g4++) {
        const f32x4 b4 = *reinterpret_cast<const f32x4 *>(s_bias + nbo * 32 + 8 * g4 + 4 * h);
#pragma unroll
        for (int q = 0; q < 4; q++) ac[nbo][4 * g4 + q] = b4[q];
      }
    constexpr int PD = 1;      // (two waves per SIMD cover the LDS latency; the registers are at their limit)
    static_assert(NB == 4, "one 16-byte weight fragment per lane and k-step");
    f32x4 wr[PD + 1];
    f32x2 ar[PD + 1];
    const f32x4 *wl = reinterpret_cast<const f32x4 *>(s_w) + h * 32 + j;
    auto fetch_k = [&](int slot, int s) __attribute__((always_inline)) {
      ar[slot] = s_aff[2 * s + h];
      wr[slot] = wl[s * 64];
    };
#pragma unroll
    for (int s = 0; s < PD; s++) fetch_k(s, s);
    const int so_p = (pb * C * L + pt * 32) * 4, vo_p = validp ? vy : 0x7FFFFF00;
    const int so_n = (nb * C * L + nt * 32) * 4;
    // A k-step's side work sits BETWEEN its four MFMAs, one piece per gap (pinned by scheduling barriers): an MFMA
    // issues when the matrix pipe is free, so whatever precedes the next MFMA in program order must have issued by
    // then -- bunched in front of the four MFMAs, the store / fetch / LDS traffic of a step took longer to issue than
    // the last MFMA takes to execute and the pipe idled every step (82 of 157 TFLOP/s).
    float v = fmaxf(fmaf(xc[0], ar[0][0], ar[0][1]), lo);
    // (two nested loops, 4 cout blocks x 16 steps: as ONE 64-step loop carrying the per-cout-block statistics / pooling
    // code the body exceeded the compiler's full-unroll budget, the loop stayed a loop and every register array it
    // indexes went to scratch memory)
#pragma unroll
    for (int cb = 0; cb < NB; cb++) {
#pragma unroll
    for (int r16 = 0; r16 < 16; r16++) {
      const int s = cb * 16 + r16;
      const f32x4 w4 = wr[s % (PD + 1)];
      const int r32 = (s & 3) + 8 * ((s & 15) >> 2), row = (s >> 4) * 32 + r32;
      ac[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(w4[0], v, ac[0], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      ts_st(ry, pv[s], vo_p, so_p + row * L * 4);            // item s of the previous block: output row (s >> 4, s & 15)
      __builtin_amdgcn_sched_barrier(0);
      ac[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(w4[1], v, ac[1], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      xp[s] = ts_ld(rx, vx, so_n + s * 2 * L * 4);           // item s of the next block's operand fetch
      if (s + PD < KS) fetch_k((s + PD) % (PD + 1), s + PD);
      __builtin_amdgcn_sched_barrier(0);
      ac[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(w4[2], v, ac[2], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      strip[(r32 + 4 * h) * TP + j] = validp ? pv[s] : 0.f;
      __builtin_amdgcn_sched_barrier(0);
      ac[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(w4[3], v, ac[3], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (s + 1 < KS) {
        const f32x2 sc = ar[(s + 1) % (PD + 1)];
        v = fmaxf(fmaf(xc[s + 1], sc[0], sc[1]), lo);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
      {    // cout block cb of the previous block is complete in the strip: its lane = channel sums (and pooling)
        const int s = cb * 16 + 15;
        ts_wave_sync();
        float t[16];
#pragma unroll
        for (int q = 0; q < 16; q++) t[q] = strip[j * TP + 2 * q + h];
#pragma unroll
        for (int q = 0; q < 16; q++) {
          ssum[s >> 4] += t[q];
          ssq[s >> 4] = fmaf(t[q], t[q], ssq[s >> 4]);
        }
        if constexpr (POOL) if (pool && validp) {
          ts_pool_block(t, psgn[s >> 4], h, pt * 32, a.poolK, pbest[s >> 4], praw[s >> 4], pk[s >> 4], rpy, rpa,
                        ((pb * C + (s >> 4) * 32 + j) * pS) * 4);
        }
        ts_wave_sync();
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    // this block's rows out of the accumulators
    // (an explicit accumulator -> architectural register move: written as a plain copy the compiler keeps the rows in
    // the accumulator file and stores from there)
#pragma unroll
    for (int s = 0; s < KS; s++) asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(pv[s]) : "a"(ac[s >> 4][s & 15]));
    __builtin_amdgcn_sched_barrier(0);
  };
  auto next = [&](int &bb, int &tt, bool go) __attribute__((always_inline)) {      // scalar only
    const int t2 = tt + 1;
    const bool wrap = t2 == nbpc;
    const int nb = wrap ? bb + 1 : bb, nt = wrap ? 0 : t2;
    bb = go ? nb : bb;
    tt = go ? nt : tt;
  };
  const int nc = n0 < a.nblk ? n0 : a.nblk - 1;
  int bA = nc / nbpc, tA = nc - bA * nbpc;
  float xa[KS], xb[KS];
  {
    const int so = (bA * C * L + tA * 32) * 4;
#pragma unroll
    for (int s = 0; s < KS; s++) xa[s] = ts_ld(rx, vx, so + s * 2 * L * 4);
  }
  int bP = bA, tP = tA;             // previous block (of the round about to run)
  bool vP = false;
  for (int it = 0, idx = n0; it < a.per; it += 2, idx += 2) {
    int bB = bA, tB = tA;
    next(bB, tB, idx + 1 < n1);
    round(xa, xb, bP, tP, vP, bB, tB);                           // block idx; block idx - 1 leaves, idx + 1 is fetched
    int bC = bB, tC = tB;
    next(bC, tC, idx + 2 < n1);
    round(xb, xa, bA, tA, idx < n1, bC, tC);                     // block idx + 1; block idx leaves, idx + 2 is fetched
    bP = bB;
    tP = tB;
    vP = idx + 1 < n1;
    bA = bC;
    tA = tC;
  }
  {   // the last block's rows
    const int so = (bP * C * L + tP * 32) * 4, vo = vP ? vy : 0x7FFFFF00;
#pragma unroll
    for (int cb = 0; cb < NB; cb++) {
#pragma unroll
      for (int r16 = 0; r16 < 16; r16++) {
        const int s = cb * 16 + r16;
        const int r32 = (s & 3) + 8 * ((s & 15) >> 2), row = (s >> 4) * 32 + r32;
        ts_st(ry, pv[s], vo, so + row * L * 4);
        strip[(r32 + 4 * h) * TP + j] = vP ? pv[s] : 0.f;
      }
      {
        const int s = cb * 16 + 15;
        ts_wave_sync();
        float t[16];
#pragma unroll
        for (int q = 0; q < 16; q++) t[q] = strip[j * TP + 2 * q + h];
#pragma unroll
        for (int q = 0; q < 16; q++) {
          ssum[s >> 4] += t[q];
          ssq[s >> 4] = fmaf(t[q], t[q], ssq[s >> 4]);
        }
        if constexpr (POOL) if (pool && vP) {
          ts_pool_block(t, psgn[s >> 4], h, tP * 32, a.poolK, pbest[s >> 4], praw[s >> 4], pk[s >> 4], rpy, rpa,
                        ((bP * C + (s >> 4) * 32 + j) * pS) * 4);
        }
        ts_wave_sync();
      }
    }
  }
  if (!a.stats) return;
  __syncthreads();
  float *red = s_t;                                 // [waves][2][C]
#pragma unroll
  for (int nbo = 0; nbo < NB; nbo++) {
    const float d1 = ssum[nbo] + __shfl_xor(ssum[nbo], 32, 64), d2 = ssq[nbo] + __shfl_xor(ssq[nbo], 32, 64);
    if (h == 0) {
      red[(wave * 2) * C + nbo * 32 + j] = d1;
      red[(wave * 2 + 1) * C + nbo * 32 + j] = d2;
    }
  }
  __syncthreads();
  for (int e = tid; e < 2 * C; e += 64 * kTpWaves) {
    const int st = e / C, c = e - st * C;
    float v = red[st * C + c];
#pragma unroll
    for (int w = 1; w < kTpWaves; w++) v += red[(w * 2 + st) * C + c];
    a.stats[((size_t)blockIdx.x * 2 + st) * C + c] = v;
  }
}

// --------------------------------------------------------------------------------- backward ----
// dy = ka g + kb y + kc (dy_mode 1) or the same with g routed from the max-pooled gradient (dy_mode 3);
// dx = (W^T dy) [f(x) > 0], f(x) = relu(isc x + ish); dstats = sum dx, sum dx * x; dW += dy f(x)^T; db += sum dy
struct TSBwd {
  const float *g, *y;
  int mode;
  const float *ka, *kb, *kc;
  const int *argmax;
  int K, S;
  const float *x;
  const float *isc, *ish, *iinv;
  const float *wpT;            // packed image of W^T
  float *dx, *dstats, *dwp, *dbp;
  long stride;                 // floats between the workgroups' dW (and db) partials
  int B, L, nblk, per;
};

template <int NB, int MODE>
__global__ __launch_bounds__(64 * kTsWaves, NB == 1 ? 3 : 1) void tstream_bwd_kernel(TSBwd a) {
  constexpr int C = 32 * NB, KS = C / 2, TP = 33;
  constexpr bool WLDS = NB > 1;                  // W^T from LDS (64 x 64 would take 64 registers per lane)
  __shared__ __attribute__((aligned(16))) float s_k[C][4];     // ka, kb, kc, -
  __shared__ __attribute__((aligned(16))) float s_sc[C], s_sh[C];
  __shared__ float s_w[WLDS ? C * C : 1];        // [k = output channel][input channel]
  __shared__ float s_t[kTsWaves][2][C * TP];     // the wave's two transposition strips; afterwards the combine buffer
  const int tid = threadIdx.x, lane = tid & 63, j = lane & 31, h = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int L = a.L, nbpc = L >> 5, K = a.K, S = a.S;
  for (int e = tid; e < C; e += 64 * kTsWaves) {
    s_k[e][0] = a.ka[e];
    s_k[e][1] = a.kb[e];
    s_k[e][2] = a.kc[e];
    s_k[e][3] = 0.f;
    s_sc[e] = a.isc[e];
    s_sh[e] = a.ish[e];
  }
  // W^T: lane (i = j, h) needs W[2 s + h][32 nbi + i] for every k-step s (k = output channel of the layer).
  // packed image of W^T: element ((kb * C + o) * 2 + hh) * 4 + q = W^T[o][8 kb + 2 q + hh] = W[8 kb + 2 q + hh][o]
  float wt[WLDS ? 1 : NB][WLDS ? 1 : KS];
  if constexpr (WLDS) {
    for (int e = tid; e < C * C; e += 64 * kTsWaves) {
      const int q = e & 3, hh = (e >> 2) & 1, o = (e >> 3) % C, kb = (e >> 3) / C;
      s_w[(8 * kb + 2 * q + hh) * C + o] = a.wpT[e];
    }
  } else {
#pragma unroll
    for (int nbi = 0; nbi < NB; nbi++) {
      const f32x4 *wv = reinterpret_cast<const f32x4 *>(a.wpT) + (size_t)(nbi * 32 + j) * 2 + h;
#pragma unroll
      for (int kb = 0; kb < C / 8; kb++) {
        const f32x4 w4 = wv[(size_t)kb * C * 2];
#pragma unroll
        for (int q = 0; q < 4; q++) wt[nbi][4 * kb + q] = w4[q];
      }
    }
  }
  f32x16 dw[NB][NB];      // [cout block][cin block]
  // per-lane partial sums with lane (i, h) = channel 32 nb + i, its tokens of parity h: sum dy | sum dx | sum dx * x
  float db[NB], s1[NB], s2[NB], shl[NB], invl[NB];
#pragma unroll
  for (int p = 0; p < NB; p++) {
    db[p] = s1[p] = s2[p] = 0.f;
    shl[p] = a.ish[p * 32 + j];
    invl[p] = a.iinv[p * 32 + j];
#pragma unroll
    for (int q = 0; q < NB; q++)
#pragma unroll
      for (int r = 0; r < 16; r++) dw[p][q][r] = 0.f;
  }
  __syncthreads();
  __builtin_amdgcn_s_waitcnt(0);           // (the prologue's loads: see the forward kernel)

  const size_t tbytes = (size_t)a.B * C * L * 4, pbytes = (size_t)a.B * C * S * 4;
  const rsrc_t rg = MODE == 3 ? ts_rsrc(a.g, pbytes) : ts_rsrc(a.g, tbytes);
  const rsrc_t ram = ts_rsrc(MODE == 3 ? (const void *)a.argmax : (const void *)a.g, MODE == 3 ? pbytes : 0);
  const rsrc_t ry = ts_rsrc(a.y, tbytes), rx = ts_rsrc(a.x, tbytes), rdx = ts_rsrc(a.dx, tbytes);
  const int gw = blockIdx.x * kTsWaves + wave;
  const int n0 = gw * a.per;
  const int n1 = n0 + a.per < a.nblk ? n0 + a.per : a.nblk;
  const int vt = (h * L + j) * 4;          // B-operand layout: channel 2 s + h, token j
  const int vd = (4 * h * L + j) * 4;      // accumulator layout: channel 8 g + 4 h + q, token j
  float *sa = s_t[wave][0], *sb = s_t[wave][1];

  // ONE register set per operand, refilled for the NEXT block right after this block has consumed it (y and the gradient
  // part after dy is formed, the forward input after the mask): the refill then has the rest of the block -- the
  // transposes and the dW products, most of its time -- to land.  Every load and store below is unconditional.
  struct BlkG {
    float g[KS];
    int am[MODE == 3 ? KS : 1];
    int k;
  };
  float ry_[KS], rx_[NB][16];
  auto load_y = [&](int bb, int tt) __attribute__((always_inline)) {
    const int so = (bb * C * L + tt * 32) * 4;
#pragma unroll
    for (int s = 0; s < KS; s++) ry_[s] = ts_ld(ry, vt, so + s * 2 * L * 4);
  };
  auto load_x = [&](int bb, int tt) __attribute__((always_inline)) {
    const int so = (bb * C * L + tt * 32) * 4;
#pragma unroll
    for (int nbi = 0; nbi < NB; nbi++)
#pragma unroll
      for (int q = 0; q < 16; q++) rx_[nbi][q] = ts_ld(rx, vd, so + (nbi * 32 + (q & 3) + 8 * (q >> 2)) * L * 4);
  };
  auto load_g = [&](BlkG &r, int bb, int tt) __attribute__((always_inline)) {
    if constexpr (MODE == 3) {
      const int t = tt * 32 + j, sc = t / K;
      r.k = t - sc * K;
      const int vp = (h * S + sc) * 4, sp = bb * C * S * 4;
#pragma unroll
      for (int s = 0; s < KS; s++) {
        r.am[s] = ts_ldi(ram, vp, sp + s * 2 * S * 4);
        r.g[s] = ts_ld(rg, vp, sp + s * 2 * S * 4);
      }
    } else {
      const int so = (bb * C * L + tt * 32) * 4;
#pragma unroll
      for (int s = 0; s < KS; s++) r.g[s] = ts_ld(rg, vt, so + s * 2 * L * 4);
    }
  };
  BlkG gq;
  auto compute = [&](int bb, int tt, bool valid, int gb, int gt) __attribute__((always_inline)) {
    // 1. dy in the B-operand layout (k = output channel), a copy into strip A; dx = W^T dy
    // (LDS operands are requested PD steps ahead of their use, in rings of register sets, the order pinned by
    // scheduling barriers: see the forward kernel)
    constexpr int PD = 2;
    float dy[KS];
    {
      f32x4 kr[PD + 1];
#pragma unroll
      for (int s = 0; s < PD; s++) kr[s] = *reinterpret_cast<const f32x4 *>(s_k[2 * s + h]);
#pragma unroll
      for (int s = 0; s < KS; s++) {
        if (s + PD < KS) kr[(s + PD) % (PD + 1)] = *reinterpret_cast<const f32x4 *>(s_k[2 * (s + PD) + h]);
        __builtin_amdgcn_sched_barrier(0);
        const f32x4 kk = kr[s % (PD + 1)];
        float gv = gq.g[s];
        if constexpr (MODE == 3) gv = gq.am[s] == gq.k ? gv : 0.f;
        const float v = fmaf(kk[0], gv, fmaf(kk[1], ry_[s], kk[2]));
        dy[s] = valid ? v : 0.f;             // (a round beyond the wave's range adds nothing to dW / db / the sums)
        sa[(2 * s + h) * TP + j] = dy[s];
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    load_g(gq, gb, gt);                    // the next block's gradient part and raw output
    load_y(gb, gt);
    f32x16 acc[NB];
#pragma unroll
    for (int nbi = 0; nbi < NB; nbi++)
#pragma unroll
      for (int q = 0; q < 16; q++) acc[nbi][q] = 0.f;
    {
      float wr[PD + 1][NB];
      auto fetch_w = [&](int slot, int s) __attribute__((always_inline)) {
        if constexpr (WLDS) {
#pragma unroll
          for (int nbi = 0; nbi < NB; nbi++) wr[slot][nbi] = s_w[(2 * s + h) * C + nbi * 32 + j];
        }
      };
#pragma unroll
      for (int s = 0; s < PD; s++) fetch_w(s, s);
#pragma unroll
      for (int s = 0; s < KS; s++) {
        if (s + PD < KS) fetch_w((s + PD) % (PD + 1), s + PD);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int nbi = 0; nbi < NB; nbi++) {
          const float wv = WLDS ? wr[s % (PD + 1)][nbi] : wt[WLDS ? 0 : nbi][WLDS ? 0 : s];
          acc[nbi] = __builtin_amdgcn_mfma_f32_32x32x2f32(wv, dy[s], acc[nbi], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    ts_wave_sync();
    // 2. dy with lane = channel: token 2 s' + h of channel 32 nbo + j (all reads first, then the sums)
    float dyc[NB][16];
#pragma unroll
    for (int nbo = 0; nbo < NB; nbo++)
#pragma unroll
      for (int s = 0; s < 16; s++) dyc[nbo][s] = sa[(nbo * 32 + j) * TP + 2 * s + h];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int nbo = 0; nbo < NB; nbo++)
#pragma unroll
      for (int s = 0; s < 16; s++) db[nbo] += dyc[nbo][s];
    ts_wave_sync();
    // 3. masked dx: to memory and into strip A; f(x) into strip B
    const int so = (bb * C * L + tt * 32) * 4;
    const int vo = valid ? vd : 0x7FFFFF00;
#pragma unroll
    for (int nbi = 0; nbi < NB; nbi++)
#pragma unroll
      for (int g4 = 0; g4 < 4; g4++) {
        const int c0 = nbi * 32 + 8 * g4;
        const f32x4 sc4 = *reinterpret_cast<const f32x4 *>(s_sc + c0 + 4 * h), sh4 = *reinterpret_cast<const f32x4 *>(s_sh + c0 + 4 * h);
#pragma unroll
        for (int q = 0; q < 4; q++) {
          const int rr = 4 * g4 + q;
          const float av = fmaf(rx_[nbi][rr], sc4[q], sh4[q]);
          const float dxm = av > 0.f ? acc[nbi][rr] : 0.f;     // (zero for an invalid round: dy was zeroed)
          ts_st(rdx, dxm, vo, so + (c0 + q) * L * 4);
          sa[(c0 + 4 * h + q) * TP + j] = dxm;
          sb[(c0 + 4 * h + q) * TP + j] = fmaxf(av, 0.f);
        }
      }
    load_x(gb, gt);                        // the next block's forward input
    ts_wave_sync();
    // 4. lane = channel again: the two sums the next BatchNorm backward needs, and dW += dy f(x)^T
#pragma unroll
    for (int nbi = 0; nbi < NB; nbi++) {
      float ac[16], dxc[16];
#pragma unroll
      for (int s = 0; s < 16; s++) {
        ac[s] = sb[(nbi * 32 + j) * TP + 2 * s + h];
        dxc[s] = sa[(nbi * 32 + j) * TP + 2 * s + h];
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int nbo = 0; nbo < NB; nbo++)
#pragma unroll
        for (int s = 0; s < 16; s++)
          dw[nbo][nbi] = __builtin_amdgcn_mfma_f32_32x32x2f32(dyc[nbo][s], ac[s], dw[nbo][nbi], 0, 0, 0);
#pragma unroll
      for (int s = 0; s < 16; s++) {
        s1[nbi] += dxc[s];
        s2[nbi] += dxc[s] * ((ac[s] - shl[nbi]) * invl[nbi]);      // (raw input where the mask is open; dxc = 0 elsewhere)
      }
    }
    ts_wave_sync();
  };
  auto next = [&](int &bb, int &tt, bool go) __attribute__((always_inline)) {      // scalar only
    const int t2 = tt + 1;
    const bool wrap = t2 == nbpc;
    const int nb = wrap ? bb + 1 : bb, nt = wrap ? 0 : t2;
    bb = go ? nb : bb;
    tt = go ? nt : tt;
  };
  const int nc = n0 < a.nblk ? n0 : a.nblk - 1;
  int bA = nc / nbpc, tA = nc - bA * nbpc;
  load_g(gq, bA, tA);
  load_y(bA, tA);
  load_x(bA, tA);
  for (int it = 0, idx = n0; it < a.per; it++, idx++) {
    int bB = bA, tB = tA;
    next(bB, tB, idx + 1 < n1);
    compute(bA, tA, idx < n1, bB, tB);
    bA = bB;
    tA = tB;
  }
  // ---- fold: the two lane halves, then the workgroup's waves in order 0..3 through one LDS image ----
  __syncthreads();
  float *img = &s_t[0][0][0];                    // [C*C dW | C db | 2 C dstats], C*C + 3 C <= 8 * C * TP floats
  float *img_db = img + C * C, *img_ds = img_db + C;
  for (int w = 0; w < kTsWaves; w++) {
    if (wave == w) {
#pragma unroll
      for (int nbo = 0; nbo < NB; nbo++)
#pragma unroll
        for (int nbi = 0; nbi < NB; nbi++)
#pragma unroll
          for (int r = 0; r < 16; r++) {
            float *d = img + (nbo * 32 + (r & 3) + 8 * (r >> 2) + 4 * h) * C + nbi * 32 + j;
            *d = w == 0 ? dw[nbo][nbi][r] : *d + dw[nbo][nbi][r];
          }
#pragma unroll
      for (int nb = 0; nb < NB; nb++) {
        // channel 32 nb + j: the two token parities (lane halves) added in a fixed order
        const float d0 = db[nb] + __shfl_xor(db[nb], 32, 64), d1 = s1[nb] + __shfl_xor(s1[nb], 32, 64),
                    d2 = s2[nb] + __shfl_xor(s2[nb], 32, 64);
        if (h == 0) {
          const int c = nb * 32 + j;
          img_db[c] = (w == 0 ? 0.f : img_db[c]) + d0;
          img_ds[c] = (w == 0 ? 0.f : img_ds[c]) + d1;
          img_ds[C + c] = (w == 0 ? 0.f : img_ds[C + c]) + d2;
        }
      }
    }
    __syncthreads();
  }
  float *dwo = a.dwp + (size_t)blockIdx.x * a.stride, *dbo = a.dbp + (size_t)blockIdx.x * a.stride;
  for (int e = tid; e < C * C; e += 64 * kTsWaves) dwo[e] = img[e];
  for (int e = tid; e < C; e += 64 * kTsWaves) dbo[e] = img_db[e];
  for (int e = tid; e < 2 * C; e += 64 * kTsWaves) a.dstats[(size_t)blockIdx.x * 2 * C + e] = img_ds[e];
}

int ts_cus() {
  static int cus = [] {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 256;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n < 1) return 256;
    return n;
  }();
  return cus;
}

}  // namespace

// ---- eligibility + grid (shared by the launch paths in train_kernels.hip and the *_groups exports) ----
// launches with fewer 32-token blocks than this stay on the tile kernels (policy knob, see pcr_set_stream_min_blocks)
static int g_ts_min_blocks = 8192;

PCR_EXPORT int pcr_set_stream_min_blocks(int n) {
  const int old = g_ts_min_blocks;
  if (n >= 0) g_ts_min_blocks = n;
  return old;
}

// forward: 32 / 64 / 128 channels in and out (32 x 32 with the weights in registers, the rest from an LDS image), one
// input tensor, no residual / output ReLU, L a multiple of 32, tensors < 2 GB
static bool ts_width(int c) { return c == 32 || c == 64 || c == 128; }

bool pcr_ts_fwd_ok(const pcr_tdense_fwd *p) {
  const int cin = p->cin1, cout = p->cout;
  if (p->cin2 || p->res || p->out_relu) return false;
  if (!(ts_width(cin) && ts_width(cout)) || ((cin == 128) != (cout == 128))) return false;
  if ((p->L & 31) || p->B < 1) return false;
  if ((size_t)p->B * (cin > cout ? cin : cout) * p->L * 4 >= (1ull << 31)) return false;
  // (worth it only when a wave gets a stream of blocks: the grouped-MLP launches, not the per-point layers)
  return (long)p->B * (p->L >> 5) >= g_ts_min_blocks;
}

static size_t ts_fwd_lds(int cin, int cout) {
  if (cin == 128) return ((size_t)3 * 128 + 128 * 128 + (size_t)kTpWaves * 32 * 33) * sizeof(float);   // (pipe kernel: the strips; the fold reuses them)
  const bool wlds = !(cin == 32 && cout == 32);
  return ((size_t)2 * cin + cout + (wlds ? (size_t)cin * cout : 0) + (size_t)kTsWaves * cout * 33) * sizeof(float);
}

static int ts_gcd(int a, int b) { return b ? ts_gcd(b, a % b) : a; }

// the launch produces the fused max over K (pool_ymax / pool_arg): a streaming launch whose clouds are whole multiples
// of lcm(32, K) tokens, K >= 32 (a 32-token block then touches at most two centres)
bool pcr_ts_fwd_pools(const pcr_tdense_fwd *p) {
  if (!pcr_ts_fwd_ok(p) || p->pool_K < 32 || (p->pool_K & 1) || !p->pool_gamma || !p->pool_ymax || !p->pool_arg) return false;
  if (p->cout == 128) return false;        // (matrix-bound launch: the separate pooling pass is cheaper, see the pipe kernel)
  if (p->L % p->pool_K) return false;
  const int lcm = p->pool_K / ts_gcd(32, p->pool_K) * 32;
  return p->L % lcm == 0;
}

int pcr_ts_fwd_grid(const pcr_tdense_fwd *p, int *per) {
  const int nblk = p->B * (p->L >> 5);
  // workgroups per CU: what the LDS image + strips admit, at most four
  int wgs_per_cu = (int)((size_t)(160 * 1024) / ts_fwd_lds(p->cin1, p->cout));
  wgs_per_cu = wgs_per_cu < 1 ? 1 : (wgs_per_cu > 4 ? 4 : wgs_per_cu);
  if (p->cout == 64 || p->cin1 == 64) wgs_per_cu = wgs_per_cu > 3 ? 3 : wgs_per_cu;
  const int waves = p->cin1 == 128 ? kTpWaves : kTsWaves;
  if (p->cin1 == 128) wgs_per_cu = 1;
  int g = ts_cus() * wgs_per_cu;
  int pw = (nblk + g * waves - 1) / (g * waves);
  // rounds come in pairs (two register sets); with the fused pooling a wave's range is whole centre groups
  // (lcm(32, K) / 32 blocks), so that a centre never straddles two waves
  int unit = 2;
  if (pcr_ts_fwd_pools(p)) {
    const int grp = p->pool_K / ts_gcd(32, p->pool_K);
    unit = (grp & 1) ? 2 * grp : grp;
  }
  pw = (pw + unit - 1) / unit * unit;
  if (pw < unit) pw = unit;
  g = (nblk + pw * waves - 1) / (pw * waves);
  if (per) *per = pw;
  return g;
}

template <int NBI, int NBO, bool WLDS>
static void ts_fwd_go(const TSFwd &a, int g, size_t lds, hipStream_t st) {
  static bool ok = allow_big_lds(tstream_fwd_kernel<NBI, NBO, WLDS>);
  (void)ok;
  hipLaunchKernelGGL((tstream_fwd_kernel<NBI, NBO, WLDS>), dim3(g), dim3(64 * kTsWaves), lds, st, a);
}

int pcr_ts_fwd_launch(const pcr_tdense_fwd *p, hipStream_t st) {
  TSFwd a;
  a.x = p->x; a.isc = p->isc; a.ish = p->ish; a.in_relu = p->in_relu; a.wp = p->wp; a.bias = p->bias;
  a.y = p->y; a.stats = p->stats; a.B = p->B; a.L = p->L; a.nblk = p->B * (p->L >> 5);
  const bool pools = pcr_ts_fwd_pools(p);
  a.poolK = pools ? p->pool_K : 0;
  a.pool_gamma = p->pool_gamma; a.pool_ymax = p->pool_ymax; a.pool_arg = p->pool_arg;
  const int g = pcr_ts_fwd_grid(p, &a.per);
  const size_t lds = ts_fwd_lds(p->cin1, p->cout);
  const int ci = p->cin1, co = p->cout;
  if (ci == 32 && co == 32) ts_fwd_go<1, 1, false>(a, g, lds, st);
  else if (ci == 64 && co == 64) ts_fwd_go<2, 2, true>(a, g, lds, st);
  else if (ci == 32 && co == 64) ts_fwd_go<1, 2, true>(a, g, lds, st);
  else if (ci == 64 && co == 32) ts_fwd_go<2, 1, true>(a, g, lds, st);
  else {
    static bool ok = allow_big_lds(tstream_fwd_pipe_kernel<4, false>);
    (void)ok;
    hipLaunchKernelGGL((tstream_fwd_pipe_kernel<4, false>), dim3(g), dim3(64 * kTpWaves), lds, st, a);
  }
  return hipGetLastError() == hipSuccess ? PCR_OK : PCR_ERR_LAUNCH;
}

// backward: square 32 / 64-channel layers behind a BatchNorm (dy_mode 1 / 3) with every output wanted
bool pcr_ts_bwd_ok(const pcr_tdense_bwd *p) {
  if (p->cin2 || p->cin1 != p->cout || !(p->cout == 32 || p->cout == 64)) return false;
  if (!(p->dy_mode == 1 || (p->dy_mode == 3 && !p->pooled && p->argmax))) return false;   // (mode 3: routed gradient)
  if (!p->isc || !p->ish || !p->iinv || !p->in_relu || !p->wpT || !p->dx || !p->dstats || !p->dwp || !p->dbp) return false;
  if ((p->L & 31) || p->B < 1) return false;
  if (p->part_stride < (long)p->cout * p->cin1 + p->cout) return false;   // (one strided block of dW | db partials)
  if ((size_t)p->B * p->cout * p->L * 4 >= (1ull << 31)) return false;
  return (long)p->B * (p->L >> 5) >= g_ts_min_blocks;
}

int pcr_ts_bwd_grid(const pcr_tdense_bwd *p, int *per) {
  const int nblk = p->B * (p->L >> 5);
  const int wgs_per_cu = p->cout == 64 ? 1 : 3;      // what registers / LDS admit (one / three waves per SIMD)
  int g = ts_cus() * wgs_per_cu;
  int pw = (nblk + g * kTsWaves - 1) / (g * kTsWaves);
  pw = (pw + 1) & ~1;                    // rounds come in pairs (two register sets)
  if (pw < 2) pw = 2;
  g = (nblk + pw * kTsWaves - 1) / (pw * kTsWaves);
  if (per) *per = pw;
  return g;
}

int pcr_ts_bwd_launch(const pcr_tdense_bwd *p, hipStream_t st) {
  TSBwd a;
  a.g = p->g; a.y = p->y; a.mode = p->dy_mode; a.ka = p->ka; a.kb = p->kb; a.kc = p->kc;
  a.argmax = p->argmax; a.K = p->K; a.S = p->S;
  a.x = p->x; a.isc = p->isc; a.ish = p->ish; a.iinv = p->iinv; a.wpT = p->wpT;
  a.dx = p->dx; a.dstats = p->dstats; a.dwp = p->dwp; a.dbp = p->dbp;
  a.stride = p->part_stride ? p->part_stride : 0;
  a.B = p->B; a.L = p->L; a.nblk = p->B * (p->L >> 5);
  const int g = pcr_ts_bwd_grid(p, &a.per);
  const dim3 grid(g), blk(64 * kTsWaves);
  if (p->cout == 32) {
    if (p->dy_mode == 3) hipLaunchKernelGGL((tstream_bwd_kernel<1, 3>), grid, blk, 0, st, a);
    else hipLaunchKernelGGL((tstream_bwd_kernel<1, 1>), grid, blk, 0, st, a);
  } else {
    if (p->dy_mode == 3) hipLaunchKernelGGL((tstream_bwd_kernel<2, 3>), grid, blk, 0, st, a);
    else hipLaunchKernelGGL((tstream_bwd_kernel<2, 1>), grid, blk, 0, st, a);
  }
  return hipGetLastError() == hipSuccess ? PCR_OK : PCR_ERR_LAUNCH;
}
