// Linear-attention kernels (pcr_attn_kv_f32 / pcr_attn_apply_f32).  Body of two translation units:
//   attn_kernels.hip      PCR_ATTN_PREC 0  f32-input MFMA, every shape, and the C-ABI entry points
//   attn_kernels_bf3.hip  PCR_ATTN_PREC 1  the dense phases of the apply kernel (Q, message, feed-forward, cov_final) as
//                                          split bf16 on v_mfma_f32_32x32x16_bf16 for d_model <= 128; the kv kernel
//                                          writes the per-cloud matrix as a bf16 image; exports pcr_attn_{kv,apply}_bf3
#pragma once
#ifndef PCR_ATTN_PREC
#define PCR_ATTN_PREC 0
#endif
#include "tile_dense.h"

#include <stdio.h>

#include <vector>

namespace {
constexpr int kAPrec = PCR_ATTN_PREC;
// -------------------------------------------------------------- linear attention ----
struct AttnArgs {
  pcr_attn_params p;
  int dbg;   // diagnostics (trace builds): 256 = stamp the shader clock
};

// shader-clock stamps of the wave-autonomous kv kernel (trace builds only: -DPCR_SA_TRACE_BUILD, PCR_ATTN_TRACE=<file>);
// one record of kATraceMarks stamps per cloud round of waves 0 and 5 of the first workgroups
constexpr int kATraceWgs = 512, kATraceRecs = 24, kATraceMarks = 8;
__device__ unsigned long long g_attn_trace[2 * kATraceWgs * kATraceRecs * kATraceMarks];

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

// merge fold of one cloud from its head-masked KV in LDS (KVl [dd][d + 1]) and total key sums (s_kt [d]):
// M[o][dd] = sum_{v in head(dd)} Wm[o][v] KV[dd][v], written as the packed A-operand image (f32 or bf16 hi / lo), then ksum
__device__ __forceinline__ void attn_kv_fold_write(const pcr_attn_params &p, const float *KVl, const float *s_kt, size_t b) {
  const int d = p.d, dh = d / p.nhead, ld = d + 1, tid = threadIdx.x;
  float *kv = p.kv + b * ((size_t)d * d + d);
  // e = (o = e / d, dd = e % d) advances by kThreads: one division per thread, then increments; the head of dd is
  // re-derived only when dd changes (never for d = 32 / 64 / 128, where kThreads % d == 0)
  const int d_o = kThreads / d, d_dd = kThreads - d_o * d;
  int o = tid / d, dd = tid - o * d;
  int v0 = (dd / dh) * dh;
  for (int e = tid; e < d * d; e += kThreads, o += d_o, dd += d_dd) {
    if (dd >= d) {
      dd -= d;
      o++;
    }
    if (d_dd) v0 = (dd / dh) * dh;
    const float *wm = p.wmerge + (size_t)o * d + v0;
    const float *kr = KVl + dd * ld + v0;
    float m = 0.f;
#pragma unroll 8
    for (int v = 0; v < dh; v++) m += wm[v] * kr[v];   // (unrolled: batches of independent loads)
    if constexpr (kAPrec == 0) {
      const int kb = dd >> 3, rem = dd & 7;
      kv[(((size_t)kb * d + o) * 2 + (rem & 1)) * 4 + (rem >> 1)] = m;
    } else {
      // the bf16 A-operand image of M (cout o, cin dd; tile_dense.h: accumulator-order K inside a 16-channel step)
      const int s16 = dd >> 4, kk = dd & 15;
      const int hh = (kk >> 2) & 1, jj = (kk & 3) + ((kk >> 3) << 2);
      const size_t unit = (((size_t)s16 * (d >> 5) + (o >> 5)) * 2) * 64 + hh * 32 + (o & 31);
      const __bf16 hi = (__bf16)m;
      const __bf16 lo = (__bf16)(m - (float)hi);
      __bf16 *img = reinterpret_cast<__bf16 *>(kv);
      img[unit * 8 + jj] = hi;
      img[(unit + 64) * 8 + jj] = lo;
    }
  }
  __syncthreads();
  if (tid < d) kv[(size_t)d * d + tid] = s_kt[tid];
}

// The same fold on the matrix core (round 5): per head, M[:, head] = Wm[:, head] KV_head^T is a dense call over k = v with
// the head's k-blocks of the packed merge weights (p.wmerge_packed) as the A operand and the TRANSPOSED KV tile in LDS as
// the B operand (KVt [v][ld], token = dd).  The scalar loop above costs every thread d d / 256 dot products of dh terms
// whose weights it fetches from global memory one float at a time -- 40-100 us per cloud at d = 128, the larger part of
// the tile kernel's launch (pt128: one 32-token tile per cloud and 0.27 ms; pt1024: a third of 0.72 ms) -- against ~4 us of
// MFMAs.  Heads of whole 32-channel blocks only (the caller falls back otherwise); the k-steps run in the loop's order.
__device__ __forceinline__ void attn_kv_fold_mfma(const pcr_attn_params &p, const float *KVt, const float *s_kt, size_t b) {
  const int d = p.d, dh = d / p.nhead, ld = d + 1, tid = threadIdx.x;
  float *kv = p.kv + b * ((size_t)d * d + d);
  for (int hd = 0; hd < p.nhead; hd++) {
    const float *wm = p.wmerge_packed + (size_t)(hd * dh / 8) * d * 8;
    tile_dense(KVt + (size_t)(hd * dh) * ld + hd * dh, dh, ld, dh >> 5, wm, d, [&](float m, int o, int t) {
      const int dd = hd * dh + t;
      if constexpr (kAPrec == 0) {
        const int kb = dd >> 3, rem = dd & 7;
        kv[(((size_t)kb * d + o) * 2 + (rem & 1)) * 4 + (rem >> 1)] = m;
      } else {
        const int s16 = dd >> 4, kk = dd & 15;
        const int hh = (kk >> 2) & 1, jj = (kk & 3) + ((kk >> 3) << 2);
        const size_t unit = (((size_t)s16 * (d >> 5) + (o >> 5)) * 2) * 64 + hh * 32 + (o & 31);
        const __bf16 hi = (__bf16)m;
        const __bf16 lo = (__bf16)(m - (float)hi);
        __bf16 *img = reinterpret_cast<__bf16 *>(kv);
        img[unit * 8 + jj] = hi;
        img[(unit + 64) * 8 + jj] = lo;
      }
    });
  }
  __syncthreads();
  if (tid < d) kv[(size_t)d * d + tid] = s_kt[tid];
}

// One workgroup per key-side cloud, token tiles of T = 32*TB (TB = 2 for d <= 64, 1 for d = 128).
// kv image per cloud: packed (d x d) matrix M (merge folded in, see above) followed by ksum[d].
// LDS: XH [c2 + d] key features ; hidden -- the fused K/V projection (2d <= c2 + d rows) is written IN PLACE over
// it (barrier between k-loop and epilogue) -- and P [3]; after the loop KVl [d][d+1].  34 KB at d = c2 = 64, so four
// workgroups share a CU; the next tile's features are fetched into registers while this tile is on the matrix core.
//   WSEL: dense shape of the 2d-row projection (2 / 1 / 1 for d = 32 / 64 / 128), NTW: KV tiles per wave (1 / 1 / 4)
// BFP (bf unit only): the K / V projection as split bf16 on the wkv_bf image (the caller checks it is there)
template <int TB, int NR, int WSEL, int NTW, bool BFP = false>
__device__ __forceinline__ void attn_kv_body(const AttnArgs &a) {
  constexpr int T = 32 * TB, RP = T + 1;
  constexpr int NPF = 4;   // 16-byte feature pieces per thread and tile held in registers (c2 * T / 4 / 256 <= NPF)
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const pcr_attn_params &p = a.p;
  const int d = p.d, c2 = p.c2;
  float *XH = smem;
  float *KB = XH;             // rows [0,d) after the projection
  float *VB = XH + d * RP;    // rows [d,2d)
  float *P = XH + (c2 + d) * RP;
  float *s_w0 = P + 3 * RP, *s_b0 = s_w0 + 3 * d;   // staged pos-MLP first layer
  // [256] partial key sums, beyond both the loop's buffers and the KVl / total-sum overlay used after it
  const int tail_a = (c2 + d + 3) * RP + 4 * d, tail_b = d * (d + 1) + d;
  float *s_ks = smem + (tail_a > tail_b ? tail_a : tail_b);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, h = lane >> 5;
  const size_t b = blockIdx.x;
  for (int e = tid; e < 3 * d; e += kThreads) s_w0[e] = p.pos0_w[e];
  for (int e = tid; e < d; e += kThreads) s_b0[e] = p.pos0_b[e];
  const float *feat = p.feat_k + b * c2 * p.Sk;
  const float *xyz = p.xyz_k + b * p.Sk * 3;
  const int nb = d >> 5, nT = nb * nb;
  const int dh = d / p.nhead;
  const float sk = (float)p.Sk;
  const float *bkv = p.bkv;

  f32x16 acc[NTW];
#pragma unroll
  for (int i = 0; i < NTW; i++)
#pragma unroll
    for (int r = 0; r < 16; r++) acc[i][r] = 0.f;
  int tile_ib[NTW];   // KV tile (ib, jb) of this wave's it-th item: the division is done once, not per token tile
#pragma unroll
  for (int it = 0; it < NTW; it++) tile_ib[it] = (wave + 4 * it) / nb;
  // key sums: thread (row = tid % d, part = tid / d) adds up its share of the tokens of every tile
  const int krow = tid % d, kpart = tid / d, kparts = kThreads / d;
  float ksum = 0.f;

  // feature tile prefetch (whole, aligned tiles: one 16-byte piece per (row, 4 tokens); others go the plain way)
  const int Q = T >> 2, npieces = c2 * Q;
#ifndef PCR_KV_PREFETCH
#define PCR_KV_PREFETCH 1
#endif
  const bool vec_ok = PCR_KV_PREFETCH && ((p.Sk & 3) == 0) && ((reinterpret_cast<size_t>(feat) & 15) == 0) && npieces <= NPF * kThreads;
  f32x4 pf[NPF];
  auto fetch = [&](int t0) {
#pragma unroll
    for (int u = 0; u < NPF; u++) {
      const int e = tid + u * kThreads;
      const int c = e / Q, q = e - c * Q;
      const bool ok = e < npieces;
      pf[u] = *reinterpret_cast<const f32x4 *>(feat + (size_t)(ok ? c : 0) * p.Sk + t0 + 4 * (ok ? q : 0));
    }
  };
  auto stash = [&]() {
#pragma unroll
    for (int u = 0; u < NPF; u++) {
      const int e = tid + u * kThreads;
      if (e < npieces) {
        const int c = e / Q, q = e - c * Q;
        float *dst = XH + c * RP + 4 * q;
        dst[0] = pf[u][0];
        dst[1] = pf[u][1];
        dst[2] = pf[u][2];
        dst[3] = pf[u][3];
      }
    }
  };
  // token split (kv_splits > 1): workgroup (b, y) takes the tiles [y * tps, (y + 1) * tps) of its cloud and leaves its
  // raw partial KV / key sums in kv_part; attn_kv_fold_kernel adds the partials in order and folds the merge projection
  const int nsplit = p.kv_splits > 1 ? p.kv_splits : 1;
  const int ntile = (p.Sk + T - 1) / T, tps = (ntile + nsplit - 1) / nsplit;
  const int t_lo = (int)blockIdx.y * tps * T;
  const int t_hi = (t_lo + tps * T) < p.Sk ? (t_lo + tps * T) : p.Sk;
  bool have = false;   // the registers hold the tile about to be processed
  if (vec_ok && t_lo + T <= p.Sk && t_lo < t_hi) {
    fetch(t_lo);
    have = true;
  }
  for (int t0 = t_lo; t0 < t_hi; t0 += T) {
    const int valid = p.Sk - t0;
    if (have) stash();
    else load_tile(XH, RP, feat, c2, c2, p.Sk, t0, T);
    load_xyz3(P, RP, xyz, p.Sk, t0, T);
    __syncthreads();
    // the next tile is whole: it is requested AFTER the projection's k-loop (the dense call's hook) and stored at the top of
    // the next round.  (Requested here, before the projection, it was older than the first refill of the projection's
    // weight ring -- a wave's vector loads retire in order, so the ring's first wait sat out the whole HBM round trip
    // and the prefetch overlapped nothing; behind the k-loop it has the epilogue and the KV MFMAs to land.  Explicit wave /
    // tile splits only: the generic split runs the hook before its k-loop, i.e. as before.)
    have = vec_ok && t0 + 2 * T <= p.Sk && t0 + T < t_hi;
    pos_hidden(XH + c2 * RP, RP, P, s_w0, s_b0, d, T);
    __syncthreads();
    // (whole-tile epilogue: a 32-cout block is all K or all V, so the branch is wave-uniform)
    // (f32-input MFMA in both units: this kernel is one workgroup per cloud walking its token tiles serially -- latency,
    // not the matrix pipe, bounds it, and the bf16 form's operand conversion measured 0-15 % SLOWER here)
    auto kv_epi = [&](const f32x16 &acc, int cb, int tb, int l31, int h) {
      const int t = tb * 32 + l31;
      float *dst = XH + (cb * 32 + 4 * h) * RP + t;
      const bool live = t < valid;
      if (cb * 32 < d) {
#pragma unroll
        for (int r = 0; r < 16; r++) dst[((r & 3) + 8 * (r >> 2)) * RP] = live ? elu1(acc[r]) : 0.f;
      } else {
#pragma unroll
        for (int r = 0; r < 16; r++) dst[((r & 3) + 8 * (r >> 2)) * RP] = live ? acc[r] / sk : 0.f;
      }
    };
    auto kv_hook = [&]() { if (have) fetch(t0 + T); };
    // (biases seed the accumulators.  Round 5: with the fold on the matrix core and the prefetch behind the k-loop the
    // projection's 256 f32 MFMAs of 64 cycles per wave and tile ARE the tile's time for long key sets, so the bf unit runs
    // them as split bf16 like the streaming kernels do -- explicit wave / tile splits, wkv_bf given)
    if constexpr (BFP && kAPrec != 0 && WSEL != 0)
      tile_dense2p<kAPrec, TB, NR, WSEL, true>(XH, c2 + d, p.wkv_bf, 2 * d, true, kv_epi, bkv, nullptr, kv_hook);
    else
      tile_dense2<TB, NR, WSEL, true>(XH, c2 + d, p.wkv, 2 * d, true, kv_epi, bkv, nullptr, kv_hook);
    __syncthreads();
    {
      const float *row = KB + krow * RP;
      for (int t = kpart; t < T; t += kparts) ksum += row[t];
    }
#pragma unroll
    for (int it = 0; it < NTW; it++) {
      const int item = wave + 4 * it;
      if (item < nT) {
        const int ib = tile_ib[it], jb = item - ib * nb;
        const float *ap = KB + (ib * 32 + l31) * RP + h;
        const float *bp = VB + (jb * 32 + l31) * RP + h;
#pragma unroll 4
        for (int ks = 0; ks < T / 2; ks++)
          acc[it] = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[2 * ks], bp[2 * ks], acc[it], 0, 0, 0);
      }
    }
    __syncthreads();   // K / V live in XH: the next tile may only be written once every wave is done with them
  }
  if (nsplit > 1) {   // raw partials: [d][d] KV (row dd, column v), then d key sums
    float *part = p.kv_part + ((size_t)b * nsplit + blockIdx.y) * ((size_t)d * d + d);
    s_ks[tid] = ksum;
#pragma unroll
    for (int it = 0; it < NTW; it++) {
      const int item = wave + 4 * it;
      if (item < nT) {
        const int ib = tile_ib[it], jb = item - ib * nb;
#pragma unroll
        for (int r = 0; r < 16; r++)
          part[(size_t)(ib * 32 + (r & 3) + 8 * (r >> 2) + 4 * h) * d + jb * 32 + l31] = acc[it][r];
      }
    }
    __syncthreads();
    if (tid < d) {
      float s = 0.f;
      for (int pp = 0; pp < kparts; pp++) s += s_ks[pp * d + tid];
      part[(size_t)d * d + tid] = s;
    }
    return;
  }
  // KV (head-masked) -> LDS [dd][d+1], then fold the merge projection and write the packed image
  float *KVl = smem;
  const int ld = d + 1;
  const bool mfma_fold = p.wmerge_packed != nullptr && (dh & 31) == 0 && (d & 31) == 0;
  float *s_kt = smem + d * ld;   // [d] total key sums
  s_ks[tid] = ksum;
#pragma unroll
  for (int it = 0; it < NTW; it++) {
    const int item = wave + 4 * it;
    if (item < nT) {
      const int ib = tile_ib[it], jb = item - ib * nb;
      const int v = jb * 32 + l31;
      const int hv = v / dh;
      const bool aligned = (dh & 31) == 0;           // heads made of whole 32-blocks: one test per tile
      const bool same_blk = (ib * 32) / dh == hv;
#pragma unroll
      for (int r = 0; r < 16; r++) {
        const int dd = ib * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        const bool same = aligned ? same_blk : (dd / dh == hv);
        if (mfma_fold) KVl[v * ld + dd] = same ? acc[it][r] : 0.f;     // (transposed: the fold's B operand)
        else KVl[dd * ld + v] = same ? acc[it][r] : 0.f;
      }
    }
  }
  __syncthreads();
  if (tid < d) {
    float s = 0.f;
    for (int pp = 0; pp < kparts; pp++) s += s_ks[pp * d + tid];
    s_kt[tid] = s;
  }
  if (mfma_fold) attn_kv_fold_mfma(p, KVl, s_kt, b);
  else attn_kv_fold_write(p, KVl, s_kt, b);
}

template <int TB, int NR, int WSEL, int NTW>
__global__ __launch_bounds__(kThreads) void attn_kv_kernel(AttnArgs a) {
  attn_kv_body<TB, NR, WSEL, NTW>(a);
}
// the same body held to a third of the register file (three workgroups per CU): worth 5-15 % for d = 64 / 128 even
// where it costs a few spilled registers, not for d = 32 (already three per CU on its own)
template <int TB, int NR, int WSEL, int NTW>
__global__ __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(3, 3))) void attn_kv_kernel_o3(AttnArgs a) {
  attn_kv_body<TB, NR, WSEL, NTW>(a);
}
// ... and to half of it: the d = 128 shape, whose LDS allows two workgroups per CU and no more
template <int TB, int NR, int WSEL, int NTW, bool BFP = false>
__global__ __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(2, 2))) void attn_kv_kernel_o2(AttnArgs a) {
  attn_kv_body<TB, NR, WSEL, NTW, BFP>(a);
}

// ---- wave-autonomous form for d = c2 = 64 (the shape of six of pt1024's eight kv launches, of every gallery / SSG kv
// launch): a WAVE owns a 32-token block end to end, no workgroup barrier inside a cloud's token loop.
//   * lane (t = lane % 32, h = lane / 32) loads / computes the channels k = 2 s + h of token t: feature channels
//     (s < 32, coalesced 128-byte rows of the channel-major input) and the position-MLP hidden channels (s >= 32);
//     register s IS the A operand of k-step s of the TRANSPOSED projection  Y^T[token][cout] = [x ; h]^T W^T;
//   * the B operand W^T comes from an LDS copy of the packed weight image, one ds_read_b128 per (cout block, four k-steps);
//   * the accumulators of the transposed projection hold, in lane (cout c, h), the tokens 8 g + 4 h + q of output channel c
//     -- which is exactly an MFMA operand of KV[dd][v] += sum_t K[dd][t] V[v][t] with the contraction index (step r,
//     half h) <-> token 8 (r / 4) + 4 h + r % 4, the same assignment for the K and the V operand: after elu + 1 / the
//     1 / Sk scale, accumulator register r of a K block and of a V block go straight into the 64 KV MFMAs.  No LDS
//     round trip, no transposition, no barrier between the 256 projection MFMAs and the 64 KV MFMAs of a block;
//   * eight waves (two per SIMD: one's loads / hidden layer / epilogue under the other's MFMAs) share one 64 KB weight
//     image; a workgroup is persistent over clouds and takes CPG = 8 / min(4, Sk / 32) clouds at a time (round 6; before:
//     min(8, Sk / 32) waves per cloud), so key sets of any length occupy every wave; the waves of a cloud add their KV in
//     wave order (fixed).
// Same per-cloud result whatever the batch (shape-only dispatch); rounding differs from the tile kernel's in the
// last bits (different summation order, 1 / Sk as a multiplication).
constexpr int kKvsWaves = 8;
// DIAG: the heads are at most 32 channels wide (nhead >= 2), so KV[dd][v] is only needed where dd and v lie in the same
// 32-block: the two off-diagonal 32 x 32 tiles are neither accumulated (32 of a block's 320 MFMAs, 32 registers) nor
// reduced.  The merge weights sit in LDS (the fold reads d * dh of them per output row; from global that was a chain
// of L2 latencies per cloud).
// BF: the projection as split bf16 (x = hi + lo, W = hi + lo from the pcr_pack_weight_bf16x2_f32 image wkv_bf; the three
// products W_hi x_hi + W_hi x_lo + W_lo x_hi on v_mfma_f32_32x32x16_bf16, f32 accumulate: 96 MFMAs of 32 cycles instead
// of 256 of 64 per block).  Lane (t, h) then holds the channels 16 s + bf_kpos(h, .) of a 16-channel step -- the K order
// of the weight image -- so eight loaded / computed values convert into one A operand (bf_split8); K / V, the KV
// accumulation and the merge fold stay f32.
// XS: 16-channel steps of the key features in the BF form (c2 = 64 or 128: the FP_SA blocks)
// ONEW (round 4; short key sets, Sk <= 128: every gallery / SSG matching launch): ONE wave per cloud.  The trace of the
// 36864-cloud gallery launch (tools/trace_attn.py) put 58 % of a workgroup round into what follows the blocks: four
// wave-order reduction rounds through LDS with a barrier each, the fold, the closing barrier.  A wave that takes all of a
// cloud's blocks needs none of it: the KV tile is accumulated TRANSPOSED (V as the A operand: lane = dd, register r =
// v 8 g + 4 h + q), which makes register r of the tile the A operand of fold step r as it stands (contraction over the v
// pairs (v_r, v_r + 4) instead of (2 s, 2 s + 1): other rounding in the last bits, still one fixed order per cloud), the
// key sums never leave the wave, and eight clouds are in flight per workgroup with no barrier after the staging.
template <bool DIAG, bool BF, int XS = 4, bool ONEW = false>
__global__ __launch_bounds__(64 * kKvsWaves) __attribute__((amdgpu_waves_per_eu(2, 2)))
void attn_kv_stream64_kernel(AttnArgs a) {
  static_assert(BF || XS == 4, "the f32 form is c2 = 64 only");
  constexpr int D = 64, LD = D + 1, KVS = D * LD + D;   // per-cloud reduction area: KVl [64][65] + key sums [64]
  constexpr int NKV = DIAG ? 2 : 4;
  constexpr int C2 = 16 * XS, WU = BF ? (XS + 4) * 512 : 4096, MINW = XS == 8 ? 4 : 2;   // weight image: 16-byte units
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const pcr_attn_params &p = a.p;
  f32x4 *s_w = reinterpret_cast<f32x4 *>(smem);                   // [16 kb][128 couts][2 halves]: 64 KB
  f32x4 *s_p0 = reinterpret_cast<f32x4 *>(smem + 4 * WU);         // [64] {w0x, w0y, w0z, b0}
  float *s_bkv = smem + 4 * WU + 256;                             // [128]
  float *s_wm = s_bkv + 128;                                      // [64][65] merge weights
  float *s_red = s_wm + D * LD;                                   // [CPG][KVS]
  const int tid = threadIdx.x, lane = tid & 63, j = lane & 31, h = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  {
    const f32x4 *src = reinterpret_cast<const f32x4 *>(BF ? p.wkv_bf : p.wkv);   // (both images are 4096 16-byte units)
    for (int e = tid; e < WU; e += 64 * kKvsWaves) s_w[e] = src[e];
    for (int e = tid; e < D * D; e += 64 * kKvsWaves) s_wm[(e >> 6) * LD + (e & 63)] = p.wmerge[e];
    if (tid < D) s_p0[tid] = f32x4{p.pos0_w[3 * tid], p.pos0_w[3 * tid + 1], p.pos0_w[3 * tid + 2], p.pos0_b[tid]};
    if (tid < 2 * D) s_bkv[tid] = p.bkv[tid];
  }
  __syncthreads();
  const int nblk = p.Sk >> 5;
  // waves per cloud: min(4, Sk / 32) rounded down to a power of two, at least MINW -- i.e. two or four clouds per workgroup
  // round.  (Until round 6 a cloud took up to EIGHT waves: at Sk = 1024 the eight partial KV tiles then met in eight
  // barrier-separated rounds, and that reduction + the fold were 30 % of a round; with four waves a cloud's blocks amortise
  // them twice as well.  In-process A/B, profiles/r06_kv_wpc_ab.txt: kv[c2=64,Sk=1024] 0.182 -> 0.162 ms, [Sk=512] 0.119 ->
  // 0.095, [c2=128,Sk=512] 0.174 -> 0.149, [c2=128,Sk=256] 0.127 -> 0.095.  TWO waves per cloud measured another 5 % at
  // 1024 clouds per launch but leave half the chip idle from 512 clouds down -- pt4096's batch: 0.30 -> 0.51 ms -- and the
  // choice may not look at the batch: a pair's bits must not depend on the batch it travels in.)  Shape-only, like the rest.
  constexpr int kMaxWpc = 4;
  const int wpc = nblk < kMaxWpc ? nblk : kMaxWpc;
  int wpc2 = 1;
  while (wpc2 * 2 <= wpc) wpc2 *= 2;
  if (wpc2 < MINW) wpc2 = MINW;                                   // (c2 = 128: LDS holds two clouds' reduction areas)
  if constexpr (ONEW) wpc2 = 1;
  const int cpg = kKvsWaves / wpc2;                               // clouds per workgroup round
  const int cslot = wave / wpc2, wsub = wave - cslot * wpc2;      // this wave's cloud slot and rank inside it
  const float inv_sk = 1.0f / (float)p.Sk;
  float bias[4];
#pragma unroll
  for (int cb = 0; cb < 4; cb++) bias[cb] = s_bkv[cb * 32 + j];
  const int dh = D / p.nhead;
#ifdef PCR_SA_TRACE_BUILD
  const bool tracing = (a.dbg & 256) && lane == 0 && (wave == 0 || wave == 5) && blockIdx.x < kATraceWgs;
  unsigned long long *trace = g_attn_trace + (size_t)(2 * blockIdx.x + (wave ? 1 : 0)) * (kATraceRecs * kATraceMarks);
  int trace_it = 0;
#define PCR_AMARK(m)                                                                                       \
  do {                                                                                                     \
    if (tracing && trace_it < kATraceRecs) trace[trace_it * kATraceMarks + (m)] = __builtin_readcyclecounter(); \
  } while (0)
#define PCR_ANEXT() trace_it++
#else
#define PCR_AMARK(m) do { } while (0)
#define PCR_ANEXT() do { } while (0)
#endif
  for (long c0 = (long)blockIdx.x * cpg; c0 < p.B; c0 += (long)gridDim.x * cpg) {
    const long b = c0 + cslot;
    const bool live = b < p.B;
    PCR_AMARK(4);
    f32x16 kv[NKV];
#pragma unroll
    for (int i = 0; i < NKV; i++)
#pragma unroll
      for (int r = 0; r < 16; r++) kv[i][r] = 0.f;
    float ks0 = 0.f, ks1 = 0.f;
    if (live) {
      const float *feat = p.feat_k + (size_t)b * C2 * p.Sk;
      const float *xyz = p.xyz_k + (size_t)b * p.Sk * 3;
      const __amdgpu_buffer_rsrc_t rfeat =
          __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(feat), 0, C2 * p.Sk * 4, 0x00020000);
      for (int blk = wsub; blk < nblk; blk += wpc2) {
        asm volatile("" ::: "memory");   // (the weight reads below stay inside the block loop: hoisted, they are 256 registers)
        PCR_AMARK(0);
        const int t = blk * 32 + j;
        const float px = xyz[3 * t], py = xyz[3 * t + 1], pz = xyz[3 * t + 2];
        f32x16 acc[4];
        if constexpr (!BF) {
          float xr[64];
          {
            // buffer loads: ONE address register (lane part) + a scalar offset per channel pair
            const int vo = (h * p.Sk + t) * 4;
#pragma unroll
            for (int s2 = 0; s2 < 32; s2++)
              xr[s2] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rfeat, vo, s2 * 2 * p.Sk * 4, 0));
          }
#pragma unroll
          for (int s2 = 0; s2 < 32; s2++) {
            const f32x4 w = s_p0[2 * s2 + h];
            const float v = w[0] * px + w[1] * py + w[2] * pz + w[3];
            xr[32 + s2] = fmaxf(v, 0.f);
          }
          // (all 32 feature loads are in flight above and land HERE: left alone, the register allocator sinks each load
          // to just before its MFMA and every k-step waits out a full memory latency)
#pragma unroll
          for (int s2 = 0; s2 < 32; s2++) asm volatile("" : "+v"(xr[s2]));
#pragma unroll
          for (int cb = 0; cb < 4; cb++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[cb][r] = 0.f;   // (the bias joins in the epilogue: seeded, 64 registers of copies were live through the load phase)
#pragma unroll
          for (int kb = 0; kb < 16; kb++) {
            f32x4 w[4];
#pragma unroll
            for (int cb = 0; cb < 4; cb++) w[cb] = s_w[((kb * 128 + cb * 32 + j) << 1) + h];
#pragma unroll
            for (int i = 0; i < 4; i++)
#pragma unroll
              for (int cb = 0; cb < 4; cb++)
                acc[cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(xr[4 * kb + i], w[cb][i], acc[cb], 0, 0, 0);
          }
        } else {
          float xf[8 * XS];
          {
            const int vo = (4 * h * p.Sk + t) * 4;
#pragma unroll
            for (int e = 0; e < 8 * XS; e++) {
              const int ch = 16 * (e >> 3) + bf_kpos(0, e & 7);   // + 4 h: in the lane offset
              xf[e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rfeat, vo, ch * p.Sk * 4, 0));
            }
          }
          bf16x8 ah[XS + 4], al[XS + 4];
#pragma unroll
          for (int s2 = 0; s2 < 4; s2++) {   // hidden channels 16 s2 + bf_kpos(h, .): the steps after the features'' 
            float hv[8];
#pragma unroll
            for (int e = 0; e < 8; e++) {
              const f32x4 w = s_p0[16 * s2 + bf_kpos(0, e) + 4 * h];
              const float v = w[0] * px + w[1] * py + w[2] * pz + w[3];
              hv[e] = relu_i(v);
            }
            bf_split8(hv, ah[XS + s2], al[XS + s2], true);
          }
#pragma unroll
          for (int e = 0; e < 8 * XS; e++) asm volatile("" : "+v"(xf[e]));   // (the loads land here, see the f32 form)
#pragma unroll
          for (int s2 = 0; s2 < XS; s2++) {
            float xv[8];
#pragma unroll
            for (int e = 0; e < 8; e++) xv[e] = xf[8 * s2 + e];
            bf_split8(xv, ah[s2], al[s2], true);
          }
#pragma unroll
          for (int cb = 0; cb < 4; cb++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[cb][r] = 0.f;
          PCR_AMARK(1);
          const bf16x8 *wb = reinterpret_cast<const bf16x8 *>(s_w) + lane;
#pragma unroll
          for (int s2 = 0; s2 < XS + 4; s2++) {
            bf16x8 wh[4], wl[4];
#pragma unroll
            for (int cb = 0; cb < 4; cb++) {
              wh[cb] = wb[((s2 * 4 + cb) * 2) * 64];
              wl[cb] = wb[((s2 * 4 + cb) * 2 + 1) * 64];
            }
#pragma unroll
            for (int cb = 0; cb < 4; cb++) acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[s2], wh[cb], acc[cb], 0, 0, 0);
#pragma unroll
            for (int cb = 0; cb < 4; cb++) acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[s2], wh[cb], acc[cb], 0, 0, 0);
#pragma unroll
            for (int cb = 0; cb < 4; cb++) acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[s2], wl[cb], acc[cb], 0, 0, 0);
          }
        }
        PCR_AMARK(2);
#pragma unroll
        for (int r = 0; r < 16; r++) {
          acc[0][r] = elu1(acc[0][r] + bias[0]);
          acc[1][r] = elu1(acc[1][r] + bias[1]);
          acc[2][r] = (acc[2][r] + bias[2]) * inv_sk;
          acc[3][r] = (acc[3][r] + bias[3]) * inv_sk;
          ks0 += acc[0][r];
          ks1 += acc[1][r];
        }
#pragma unroll
        for (int i = 0; i < NKV; i++) {
          const int ib = DIAG ? i : (i >> 1), jb = DIAG ? i : (i & 1);
          // (round 6, measured and dropped: this accumulation as split bf16 too -- registers [8 st, 8 st + 8) of a K block and of a
          // V block hold the same eight tokens, so one bf_split8 of each feeds a 16-token step: 6 MFMAs of 32 cycles instead of 16
          // of 64 per tile.  kv[d=64,Sk=1024] 0.177 -> 0.166 ms, gallery128 -0.12 ms -- and the headline's deviation from the f32
          // path 2.7e-5 -> 3.2e-5, one case of the margin sweep past 5e-5: the KV state feeds every query token, its rounding does
          // not average out.  Stays f32.)
#pragma unroll
          for (int r = 0; r < 16; r++) {
            if constexpr (ONEW) kv[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(acc[2 + jb][r], acc[ib][r], kv[i], 0, 0, 0);
            else kv[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(acc[ib][r], acc[2 + jb][r], kv[i], 0, 0, 0);
          }
        }
        PCR_AMARK(3);
      }
    }
    PCR_AMARK(5);
    ks0 += __shfl_xor(ks0, 32, 64);
    ks1 += __shfl_xor(ks1, 32, 64);
    if constexpr (ONEW) {
      // kv[i]: lane (dd = 32 ib + j, h), register r <-> v = 32 jb + 8 g + 4 h + q; fold tile (o block, dd block) from registers
      if (live) {
        float *kvo = p.kv + (size_t)b * ((size_t)D * D + D);
        bf16x8 *img = reinterpret_cast<bf16x8 *>(kvo);
#pragma unroll
        for (int db = 0; db < 2; db++) {
          const int hd = (db * 32 + j) / dh;
#pragma unroll
          for (int ob = 0; ob < 2; ob++) {
            f32x16 m;
#pragma unroll
            for (int r = 0; r < 16; r++) m[r] = 0.f;
#pragma unroll
            for (int i = 0; i < NKV; i++) {
              const int ib = DIAG ? i : (i >> 1), jb = DIAG ? i : (i & 1);
              if (ib != db) continue;
              const float *wr = s_wm + (ob * 32 + j) * LD + jb * 32 + 4 * h;
#pragma unroll
              for (int r = 0; r < 16; r++) {
                const int vl = (r & 3) + 8 * (r >> 2);
                const float av = (jb * 32 + vl + 4 * h) / dh == hd ? kv[i][r] : 0.f;
                m = __builtin_amdgcn_mfma_f32_32x32x2f32(av, wr[vl], m, 0, 0, 0);
              }
            }
#pragma unroll
            for (int G = 0; G < 2; G++) {
              float v[8];
#pragma unroll
              for (int e = 0; e < 8; e++) v[e] = m[8 * G + e];
              bf16x8 hi, lo;
              bf_split8(v, hi, lo, true);
              const size_t unit = (((size_t)(db * 2 + G) * (D >> 5) + ob) * 2) * 64 + h * 32 + j;
              img[unit] = hi;
              img[unit + 64] = lo;
            }
          }
        }
        if (h == 0) {
          kvo[(size_t)D * D + j] = ks0;
          kvo[(size_t)D * D + 32 + j] = ks1;
        }
      }
      PCR_AMARK(6);
      PCR_AMARK(7);
      PCR_ANEXT();
      continue;
    }
    // the waves of a cloud add their KV / key sums into the cloud's LDS area in wave order
    float *KVl = s_red + cslot * KVS, *s_kt = KVl + D * LD;
    for (int round = 0; round < wpc2; round++) {
      if (wsub == round) {
#pragma unroll
        for (int i = 0; i < NKV; i++) {
          const int ib = DIAG ? i : (i >> 1), jb = DIAG ? i : (i & 1);
#pragma unroll
          for (int r = 0; r < 16; r++) {
            const int dd = ib * 32 + (r & 3) + 8 * (r >> 2) + 4 * h, v = jb * 32 + j;
            float *dst = KVl + dd * LD + v;
            *dst = round == 0 ? kv[i][r] : *dst + kv[i][r];
          }
        }
        if (h == 0) {
          s_kt[j] = round == 0 ? ks0 : s_kt[j] + ks0;
          s_kt[32 + j] = round == 0 ? ks1 : s_kt[32 + j] + ks1;
        }
      }
      __syncthreads();
    }
    PCR_AMARK(6);
    // merge fold M[o][dd] = sum_{v in head(dd)} Wm[o][v] KV[dd][v] on the matrix core: four 32 x 32 tiles (o block, dd
    // block) over the cloud's waves, operands straight from LDS (A = Wm rows, B = KV rows with the head mask applied on
    // the read; DIAG: only the dd block's own 32 columns exist), then the packed image of M and the key sums
    if (live) {
      float *kvo = p.kv + (size_t)b * ((size_t)D * D + D);
      for (int ti = wsub; ti < 4; ti += wpc2) {
        const int ob = ti >> 1, db = ti & 1;
        const float *ap = s_wm + (ob * 32 + j) * LD + h;
        const float *bp = KVl + (db * 32 + j) * LD + h;
        const int hd = (db * 32 + j) / dh;
        f32x16 m;
#pragma unroll
        for (int r = 0; r < 16; r++) m[r] = 0.f;
        const int s_lo = DIAG ? db * 16 : 0, s_hi = DIAG ? db * 16 + 16 : 32;
        if constexpr (kAPrec == 0) {
#pragma unroll 4
          for (int s2 = s_lo; s2 < s_hi; s2++) {
            const float bv = bp[2 * s2];
            m = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[2 * s2], (2 * s2 + h) / dh == hd ? bv : 0.f, m, 0, 0, 0);
          }
          const int dd = db * 32 + j;
#pragma unroll
          for (int r = 0; r < 16; r++) {
            const int o = ob * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            const int kb = dd >> 3, rem = dd & 7;
            kvo[(((size_t)kb * D + o) * 2 + (rem & 1)) * 4 + (rem >> 1)] = m[r];
          }
        } else {
          // the bf16 image wants, per (output row o, half h), the eight cin values 16 s + bf_kpos(h, .) as ONE 16-byte unit:
          // with the fold TRANSPOSED (KV rows as the A operand, Wm rows as B: the same products summed in the same order,
          // so the same bits) lane (o, h) holds dd = 8 g + 4 h + q in register 4 g + q -- registers [8 G, 8 G + 8) are
          // exactly the unit of step 2 db + G.  (Round 3 stored the untransposed tile element by element: 32 scattered
          // 2-byte stores per lane and tile, 0.5 ms of a 36864-cloud gallery launch.)
#pragma unroll 4
          for (int s2 = s_lo; s2 < s_hi; s2++) {
            const float bv = bp[2 * s2];
            m = __builtin_amdgcn_mfma_f32_32x32x2f32((2 * s2 + h) / dh == hd ? bv : 0.f, ap[2 * s2], m, 0, 0, 0);
          }
          bf16x8 *img = reinterpret_cast<bf16x8 *>(kvo);
#pragma unroll
          for (int G = 0; G < 2; G++) {
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; e++) v[e] = m[8 * G + e];
            bf16x8 hi, lo;
            bf_split8(v, hi, lo, true);
            const size_t unit = (((size_t)(db * 2 + G) * (D >> 5) + ob) * 2) * 64 + h * 32 + j;
            img[unit] = hi;
            img[unit + 64] = lo;
          }
        }
      }
      for (int e = wsub * 64 + lane; e < D; e += 64 * wpc2) kvo[(size_t)D * D + e] = s_kt[e];
    }
    PCR_AMARK(7);
    PCR_ANEXT();
    __syncthreads();   // the next round overwrites the reduction areas
  }
#undef PCR_AMARK
#undef PCR_ANEXT
}

#if PCR_ATTN_PREC != 0
// The same kernel shape for d = c2 = 32 (the SA1 self-attention), split-bf16 projection only: one K and one V block,
// one 32 x 32 KV tile per wave, one fold tile per cloud.
__global__ __launch_bounds__(64 * kKvsWaves) __attribute__((amdgpu_waves_per_eu(2, 4)))
void attn_kv_stream32_kernel(AttnArgs a) {
  constexpr int D = 32, LD = D + 1, KVS = D * LD + D;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const pcr_attn_params &p = a.p;
  f32x4 *s_w = reinterpret_cast<f32x4 *>(smem);                   // bf image [4 steps][2 cb][hi, lo][64]: 1024 units
  f32x4 *s_p0 = reinterpret_cast<f32x4 *>(smem + 4096);           // [32] {w0x, w0y, w0z, b0}
  float *s_bkv = smem + 4096 + 128;                               // [64]
  float *s_wm = s_bkv + 64;                                       // [32][33] merge weights
  float *s_red = s_wm + D * LD;                                   // [CPG][KVS]
  const int tid = threadIdx.x, lane = tid & 63, j = lane & 31, h = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  {
    const f32x4 *src = reinterpret_cast<const f32x4 *>(p.wkv_bf);
    for (int e = tid; e < 1024; e += 64 * kKvsWaves) s_w[e] = src[e];
    for (int e = tid; e < D * D; e += 64 * kKvsWaves) s_wm[(e >> 5) * LD + (e & 31)] = p.wmerge[e];
    if (tid < D) s_p0[tid] = f32x4{p.pos0_w[3 * tid], p.pos0_w[3 * tid + 1], p.pos0_w[3 * tid + 2], p.pos0_b[tid]};
    if (tid < 2 * D) s_bkv[tid] = p.bkv[tid];
  }
  __syncthreads();
  const int nblk = p.Sk >> 5;
  const int wpc = nblk < kKvsWaves ? nblk : kKvsWaves;
  int wpc2 = 1;
  while (wpc2 * 2 <= wpc) wpc2 *= 2;
  if (wpc2 < 2) wpc2 = 2;
  const int cpg = kKvsWaves / wpc2;
  const int cslot = wave / wpc2, wsub = wave - cslot * wpc2;
  const float inv_sk = 1.0f / (float)p.Sk;
  const float bias0 = s_bkv[j], bias1 = s_bkv[32 + j];
  const int dh = D / p.nhead;
  for (long c0 = (long)blockIdx.x * cpg; c0 < p.B; c0 += (long)gridDim.x * cpg) {
    const long b = c0 + cslot;
    const bool live = b < p.B;
    f32x16 kv;
#pragma unroll
    for (int r = 0; r < 16; r++) kv[r] = 0.f;
    float ks = 0.f;
    if (live) {
      const float *feat = p.feat_k + (size_t)b * D * p.Sk;
      const float *xyz = p.xyz_k + (size_t)b * p.Sk * 3;
      const __amdgpu_buffer_rsrc_t rfeat =
          __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(feat), 0, D * p.Sk * 4, 0x00020000);
      for (int blk = wsub; blk < nblk; blk += wpc2) {
        asm volatile("" ::: "memory");
        const int t = blk * 32 + j;
        const float px = xyz[3 * t], py = xyz[3 * t + 1], pz = xyz[3 * t + 2];
        float xf[16];
        {
          const int vo = (4 * h * p.Sk + t) * 4;
#pragma unroll
          for (int e = 0; e < 16; e++) {
            const int ch = 16 * (e >> 3) + bf_kpos(0, e & 7);   // + 4 h: in the lane offset
            xf[e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rfeat, vo, ch * p.Sk * 4, 0));
          }
        }
        bf16x8 ah[4], al[4];
#pragma unroll
        for (int s2 = 0; s2 < 2; s2++) {
          float hv[8];
#pragma unroll
          for (int e = 0; e < 8; e++) {
            const f32x4 w = s_p0[16 * s2 + bf_kpos(0, e) + 4 * h];
            const float v = w[0] * px + w[1] * py + w[2] * pz + w[3];
            hv[e] = relu_i(v);
          }
          bf_split8(hv, ah[2 + s2], al[2 + s2], true);
        }
#pragma unroll
        for (int e = 0; e < 16; e++) asm volatile("" : "+v"(xf[e]));
#pragma unroll
        for (int s2 = 0; s2 < 2; s2++) {
          float xv[8];
#pragma unroll
          for (int e = 0; e < 8; e++) xv[e] = xf[8 * s2 + e];
          bf_split8(xv, ah[s2], al[s2], true);
        }
        f32x16 acc[2];
#pragma unroll
        for (int cb = 0; cb < 2; cb++)
#pragma unroll
          for (int r = 0; r < 16; r++) acc[cb][r] = 0.f;
        const bf16x8 *wb = reinterpret_cast<const bf16x8 *>(s_w) + lane;
#pragma unroll
        for (int s2 = 0; s2 < 4; s2++) {
          bf16x8 wh[2], wl[2];
#pragma unroll
          for (int cb = 0; cb < 2; cb++) {
            wh[cb] = wb[((s2 * 2 + cb) * 2) * 64];
            wl[cb] = wb[((s2 * 2 + cb) * 2 + 1) * 64];
          }
#pragma unroll
          for (int cb = 0; cb < 2; cb++) acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[s2], wh[cb], acc[cb], 0, 0, 0);
#pragma unroll
          for (int cb = 0; cb < 2; cb++) acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[s2], wh[cb], acc[cb], 0, 0, 0);
#pragma unroll
          for (int cb = 0; cb < 2; cb++) acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[s2], wl[cb], acc[cb], 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 16; r++) {
          acc[0][r] = elu1(acc[0][r] + bias0);
          acc[1][r] = (acc[1][r] + bias1) * inv_sk;
          ks += acc[0][r];
        }
#pragma unroll
        for (int r = 0; r < 16; r++) kv = __builtin_amdgcn_mfma_f32_32x32x2f32(acc[0][r], acc[1][r], kv, 0, 0, 0);
      }
    }
    ks += __shfl_xor(ks, 32, 64);
    float *KVl = s_red + cslot * KVS, *s_kt = KVl + D * LD;
    for (int round = 0; round < wpc2; round++) {
      if (wsub == round) {
#pragma unroll
        for (int r = 0; r < 16; r++) {
          float *dst = KVl + ((r & 3) + 8 * (r >> 2) + 4 * h) * LD + j;
          *dst = round == 0 ? kv[r] : *dst + kv[r];
        }
        if (h == 0) s_kt[j] = round == 0 ? ks : s_kt[j] + ks;
      }
      __syncthreads();
    }
    if (live && wsub == 0) {
      float *kvo = p.kv + (size_t)b * ((size_t)D * D + D);
      const float *ap = s_wm + j * LD + h;
      const float *bp = KVl + j * LD + h;
      const int hd = j / dh;
      f32x16 m;
#pragma unroll
      for (int r = 0; r < 16; r++) m[r] = 0.f;
      // (transposed fold, whole 16-byte units per lane: see attn_kv_stream64_kernel)
#pragma unroll 4
      for (int s2 = 0; s2 < 16; s2++) {
        const float bv = bp[2 * s2];
        m = __builtin_amdgcn_mfma_f32_32x32x2f32((2 * s2 + h) / dh == hd ? bv : 0.f, ap[2 * s2], m, 0, 0, 0);
      }
      bf16x8 *img = reinterpret_cast<bf16x8 *>(kvo);
#pragma unroll
      for (int G = 0; G < 2; G++) {
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; e++) v[e] = m[8 * G + e];
        bf16x8 hi, lo;
        bf_split8(v, hi, lo, true);
        const size_t unit = (((size_t)G * (D >> 5)) * 2) * 64 + h * 32 + j;
        img[unit] = hi;
        img[unit + 64] = lo;
      }
      if (lane < D) kvo[(size_t)D * D + lane] = s_kt[lane];
    }
    __syncthreads();
  }
}
#endif

// second launch of the token-split form: one workgroup per cloud adds the splits' partials in order (fixed: the result
// does not depend on the schedule), applies the head mask and folds the merge projection
__global__ __launch_bounds__(kThreads) void attn_kv_fold_kernel(AttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const pcr_attn_params &p = a.p;
  const int d = p.d, dh = d / p.nhead, ld = d + 1, nsplit = p.kv_splits, tid = threadIdx.x;
  const size_t b = blockIdx.x, per = (size_t)d * d + d;
  float *KVl = smem, *s_kt = smem + d * ld;
  const float *part = p.kv_part + b * nsplit * per;
  for (int e = tid; e < d * d + d; e += kThreads) {
    float v = part[e];
    for (int y = 1; y < nsplit; y++) v += part[(size_t)y * per + e];
    if (e < d * d) {
      const int dd = e / d, c = e - dd * d;
      KVl[dd * ld + c] = (dd / dh == c / dh) ? v : 0.f;
    } else {
      s_kt[e - d * d] = v;
    }
  }
  __syncthreads();
  attn_kv_fold_write(p, KVl, s_kt, b);
}

#if PCR_ATTN_PREC == 0
// d_model 256 / 512 (the 1.5M / 7M Point-Transformer configs, backbone_net.py:43-46,84-86): the per-cloud KV matrix
// no longer fits one workgroup's accumulators, so a cloud is split over d / 64 workgroups.  Workgroup (band g, cloud b)
// owns the rows dd in [64 g, 64 g + 64) of KV -- one head, since the head width is a multiple of 64 -- and therefore
// needs the K rows of its band and the V rows of its head: the host gathers exactly those rows of the fused K/V
// projection into one packed image per band (wkv_wide, 64 + dh couts), so the projection is one in-place dense call
// as in the narrow kernel.  With a whole head's V columns at hand the merge fold M[:, band] = Wm[:, head] KV_band^T
// is complete inside the workgroup (no cross-workgroup reduction) and runs on the matrix core as well.
//   NRW: cout-block rounds of the 64 + dh row projection (2 for dh = 128, 3 for dh = 256)
template <int NRW>
__global__ __launch_bounds__(kThreads) void attn_kv_wide_kernel(AttnArgs a) {
  constexpr int TB = 1, T = 32, RP = 33, BAND = 64, RPK = BAND + 1;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const pcr_attn_params &p = a.p;
  const int d = p.d, c2 = p.c2, dh = d / p.nhead;
  const int g = blockIdx.x, hd = (g * BAND) / dh;
  const int OPW = BAND + dh;                 // couts of this band's projection (multiple of 32)
  float *XH = smem;                          // [c2 + d][RP]: key features ; hidden -> rows [0,64) K band, [64,64+dh) V head
  float *KB = XH, *VB = XH + BAND * RP;
  float *P = XH + (c2 + d) * RP;
  float *s_w0 = P + 3 * RP, *s_b0 = s_w0 + 3 * d;
  float *s_ks = s_b0 + d;                    // [256] partial key sums
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, h = lane >> 5;
  const size_t b = blockIdx.y;
  for (int e = tid; e < 3 * d; e += kThreads) s_w0[e] = p.pos0_w[e];
  for (int e = tid; e < d; e += kThreads) s_b0[e] = p.pos0_b[e];
  const float *feat = p.feat_k + b * c2 * p.Sk;
  const float *xyz = p.xyz_k + b * p.Sk * 3;
  const float sk = (float)p.Sk;
  const float *wband = p.wkv_wide + (size_t)g * ((size_t)ceil8(c2 + d) * OPW);
  const float *bband = p.bkv_wide + (size_t)g * OPW;
  const int nj = dh >> 5, nT = 2 * nj;       // KV tiles (ib in {0,1}) x (jb < dh/32): 8 or 16, up to 4 per wave
  f32x16 acc[4];
#pragma unroll
  for (int i = 0; i < 4; i++)
#pragma unroll
    for (int r = 0; r < 16; r++) acc[i][r] = 0.f;
  const int krow = tid & (BAND - 1), kpart = tid >> 6;   // four partial sums per K row
  float ksum = 0.f;
  for (int t0 = 0; t0 < p.Sk; t0 += T) {
    const int valid = p.Sk - t0;
    load_tile(XH, RP, feat, c2, c2, p.Sk, t0, T);
    load_xyz3(P, RP, xyz, p.Sk, t0, T);
    __syncthreads();
    pos_hidden(XH + c2 * RP, RP, P, s_w0, s_b0, d, T);
    __syncthreads();
    tile_dense2<TB, NRW, 0, true>(XH, c2 + d, wband, OPW, true,
                                  [&](const f32x16 &acc, int cb, int tb, int l31, int h) {
      const int t = tb * 32 + l31;
      float *dst = XH + (cb * 32 + 4 * h) * RP + t;
      const bool live = t < valid;
      if (cb * 32 < BAND) {
#pragma unroll
        for (int r = 0; r < 16; r++) dst[((r & 3) + 8 * (r >> 2)) * RP] = live ? elu1(acc[r]) : 0.f;
      } else {
#pragma unroll
        for (int r = 0; r < 16; r++) dst[((r & 3) + 8 * (r >> 2)) * RP] = live ? acc[r] / sk : 0.f;
      }
    }, bband);
    __syncthreads();
    {
      const float *row = KB + krow * RP;
      for (int t = kpart; t < T; t += 4) ksum += row[t];
    }
#pragma unroll
    for (int it = 0; it < 4; it++) {
      const int item = wave + 4 * it;
      if (item < nT) {
        const int ib = item / nj, jb = item - ib * nj;
        const float *ap = KB + (ib * 32 + l31) * RP + h;
        const float *bp = VB + (jb * 32 + l31) * RP + h;
#pragma unroll 4
        for (int ks = 0; ks < T / 2; ks++)
          acc[it] = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[2 * ks], bp[2 * ks], acc[it], 0, 0, 0);
      }
    }
    __syncthreads();
  }
  // KV band, transposed: KVt [v (dh)][dd (64) + 1] = the B operand (k = v, token = dd) of the merge fold
  // (KVt spans dh * 65 floats from the start of the buffer and can reach past s_ks -- d_model 256 with ONE head and
  // c2 < 214 -- so the four key-sum partials are combined into a register before the first KVt write)
  float *KVt = smem;
  s_ks[tid] = ksum;
  __syncthreads();
  const float ks_total = tid < BAND ? s_ks[tid] + s_ks[64 + tid] + s_ks[128 + tid] + s_ks[192 + tid] : 0.f;
  __syncthreads();
#pragma unroll
  for (int it = 0; it < 4; it++) {
    const int item = wave + 4 * it;
    if (item < nT) {
      const int ib = item / nj, jb = item - ib * nj;
      const int v = jb * 32 + l31;
#pragma unroll
      for (int r = 0; r < 16; r++) KVt[v * RPK + ib * 32 + (r & 3) + 8 * (r >> 2) + 4 * h] = acc[it][r];
    }
  }
  __syncthreads();
  float *kv = p.kv + b * ((size_t)d * d + d);
  if (tid < BAND) kv[(size_t)d * d + g * BAND + tid] = ks_total;
  // M[o][dd] = sum_{v < dh} Wm[o][hd dh + v] KV[dd][v]: dense over k = v with the k-blocks [hd dh / 8, +dh / 8) of the
  // packed merge weights; stored in the packed (d,d) layout the apply kernel reads as an A operand
  const float *wm = p.wmerge_packed + (size_t)(hd * dh / 8) * d * 8;
  tile_dense(KVt, dh, RPK, 2, wm, d, [&](float m, int o, int t) {
    const int dd = g * BAND + t;
    const int kb = dd >> 3, rem = dd & 7;
    kv[(((size_t)kb * d + o) * 2 + (rem & 1)) * 4 + (rem >> 1)] = m;
  });
}

#endif   // PCR_ATTN_PREC == 0 (wide kv kernel)

// One workgroup per (query cloud, tile of T query tokens), T = 128 / 64 / 32 for d = 32 / 64 / 128 (32 beyond).
// ONE LDS buffer U of max(c1 + d, 2d, cout, cfinal) rows, every dense phase in place (barrier between its
// k-loop and its epilogue), so a d = 64 tile is 33 KB and four workgroups share a CU:
//   rows [0,c1) query features x, rows [c1,c1+d) position hidden h  --Q-->  rows [c1,c1+d) = elu(.)+1
//   --scale by Sk/(Q.ksum)--> --M (kv image)--> message --LayerNorm--> [x ; msg] --FFN0--> 2d rows --FFN1-->
//   cout rows --LayerNorm--> (+ x, re-read from global: it was overwritten by FFN0) --cov_final--> store.
template <int TB, int NR>
__global__ __launch_bounds__(kThreads) void attn_apply_kernel(AttnArgs a) {
  constexpr int T = 32 * TB, RP = T + 1;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const pcr_attn_params &p = a.p;
  const int d = p.d, c1 = p.c1, cout = p.cout;
  const int catC = c1 + d, catP = ceil8(catC);
  int rowsU = catP > 2 * d ? catP : 2 * d;
  if (ceil32(cout) > rowsU) rowsU = ceil32(cout);
  if (ceil32(p.cfinal) > rowsU) rowsU = ceil32(p.cfinal);
  float *U = smem;
  float *P = U + rowsU * RP;
  float *zs = P + 3 * RP;
  float *red = zs + p.nhead * RP;  // [2 * (256/T)][T]
  // small constant vectors, staged once: reading them from global inside the per-element loops costs a
  // vector-memory instruction per use
  float *cst = red + 2 * (kThreads / T) * T;
  float *s_ksum = cst, *s_ln1g = cst + d, *s_ln1b = cst + 2 * d, *s_ln2g = cst + 3 * d, *s_ln2b = s_ln2g + cout;
  float *s_w0 = s_ln2b + cout, *s_b0 = s_w0 + 3 * d;
  const int tid = threadIdx.x;
  const size_t b = blockIdx.y;
  const int t0 = blockIdx.x * T;
  const size_t bq_ = p.q_index ? (size_t)p.q_index[b] : b;    // which cloud supplies the query tokens
  const float *feat = p.feat_q + bq_ * c1 * p.Lq;
  const size_t kb_ = p.kv_index ? (size_t)p.kv_index[b] : b;
  const float *kv = p.kv + kb_ * ((size_t)d * d + d);
  const int dh = d / p.nhead;
  for (int e = tid; e < d; e += kThreads) {
    s_ksum[e] = kv[(size_t)d * d + e];
    s_ln1g[e] = p.ln1_g[e];
    s_ln1b[e] = p.ln1_b[e];
    if (p.q_pos) {
      s_w0[3 * e] = p.pos0_w[3 * e];
      s_w0[3 * e + 1] = p.pos0_w[3 * e + 1];
      s_w0[3 * e + 2] = p.pos0_w[3 * e + 2];
      s_b0[e] = p.pos0_b[e];
    }
  }
  for (int e = tid; e < cout; e += kThreads) {
    s_ln2g[e] = p.ln2_g[e];
    s_ln2b[e] = p.ln2_b[e];
  }
  const float *ksum = s_ksum;
  float *MSG = U + c1 * RP;   // rows [c1, c1+d): hidden -> Q -> message

  load_tile(U, RP, feat, c1, c1, p.Lq, t0, T);
  if (p.q_pos) {
    load_xyz3(P, RP, p.xyz_q + bq_ * p.Lq * 3, p.Lq, t0, T);
    __syncthreads();
    pos_hidden(MSG, RP, P, s_w0, s_b0, d, T);
    for (int e = tid; e < (catP - catC) * T; e += kThreads) U[(catC + e / T) * RP + e % T] = 0.f;
  } else {
    for (int e = tid; e < (catP - c1) * T; e += kThreads) U[(c1 + e / T) * RP + e % T] = 0.f;
  }
  __syncthreads();
  // Q = elu(Wq' [x ; h] + bq) + 1, written over h
  tile_dense2p<kAPrec, TB, NR>(U, p.q_pos ? catP : ceil8(c1), p.wq, d, true,
                               [&](float v, int o, int t) { MSG[o * RP + t] = elu1(v); }, p.bq);
  __syncthreads();
  for (int e = tid; e < p.nhead * T; e += kThreads) {
    const int hd = e / T, t = e - hd * T;
    float z = 0.f;
    for (int c = 0; c < dh; c++) z += MSG[(hd * dh + c) * RP + t] * ksum[hd * dh + c];
    zs[hd * RP + t] = (1.0f / (z + 1e-6f)) * (float)p.Sk;
  }
  __syncthreads();
  for (int hd = 0; hd < p.nhead; hd++)   // (head-major: no runtime division per element)
    for (int e = tid; e < dh * T; e += kThreads) {
      const int o = hd * dh + e / T, t = e % T;
      MSG[o * RP + t] *= zs[hd * RP + t];
    }
  __syncthreads();
  tile_dense2p<kAPrec, TB, NR>(MSG, d, kv, d, true, [&](float v, int o, int t) { MSG[o * RP + t] = v; });
  __syncthreads();
  tile_layernorm(MSG, d, RP, T, s_ln1g, s_ln1b, red);
  tile_dense2p<kAPrec, TB, NR>(U, catP, p.wmlp0, 2 * d, true, [&](float v, int o, int t) { U[o * RP + t] = fmaxf(v, 0.f); });
  __syncthreads();
  tile_dense2p<kAPrec, TB, NR>(U, 2 * d, p.wmlp2, ceil32(cout), true, [&](float v, int o, int t) { U[o * RP + t] = v; });
  __syncthreads();
  tile_layernorm(U, cout, RP, T, s_ln2g, s_ln2b, red);
  if (p.residual) {   // cout == c1: add the query features back (re-read: FFN0 has overwritten them)
    for (int e = tid; e < cout * T; e += kThreads) {
      const int c = e / T, t = e - c * T;
      if (t0 + t < p.Lq) U[c * RP + t] += feat[(size_t)c * p.Lq + t0 + t];
    }
    __syncthreads();
  }
  int cres = cout;
  if (p.cfinal) {  // trailing 1x1 conv with bias (cov_final); needs cout % 8 == 0
    const int cf = p.cfinal;
    tile_dense2p<kAPrec, TB, NR>(U, cout, p.wfinal, ceil32(cf), true, [&](float v, int o, int t) { U[o * RP + t] = v; },
                                 p.bfinal);   // bfinal is zero-padded to a multiple of 32 by the host
    __syncthreads();
    cres = cf;
  }
  float *out = p.out + b * cres * p.Lq;
  for (int e = tid; e < cres * T; e += kThreads) {
    const int c = e / T, t = e - c * T;
    if (t0 + t < p.Lq) out[(size_t)c * p.Lq + t0 + t] = U[c * RP + t];
  }
}

#if PCR_ATTN_PREC != 0
// ---- wave-autonomous apply kernel, d = c1 = cout = 64, split bf16, no trailing conv, whole 32-token blocks ----------
// (gallery / SSG matching: every apply launch; pt1024: the three self / cross attention launches of SA2 and the matching
// stages.)  A wave owns a 32-token block from the feature load to the output store -- NO workgroup barrier after the
// weights are staged.  Every dense phase is in the normal orientation D[cout][token] with the activations as the B
// operand held in registers, lane (t, h) = token t, k-group h:
//   * the bf16 weight images order the 16 channels of a step as 16 s + bf_kpos(h, j) -- the order in which a 32 x 32
//     accumulator tile hands its rows to a lane -- so registers [8 G, 8 G + 8) of cout block cb of one phase's
//     accumulators convert (bf_split8) into the B operand of step s = 2 cb + G of the NEXT phase: Q -> message -> FFN0 ->
//     FFN1 chain through registers, no LDS tile, no transposition;
//   * the input is loaded in the same order, so element i of the lane's 32 feature values is also element i of the
//     final accumulators (the residual add) and of the output store;
//   * LayerNorm over the 64 channels of a token = the lane's 32 registers + its partner lane (lane ^ 32): two
//     cross-lane adds per moment instead of two barrier-separated passes over an LDS tile; the per-head normaliser of
//     linear attention likewise;
//   * the per-cloud matrix M (bf16 image written by the kv kernel) is the A operand of the message phase, read straight
//     from global memory / L2 by each wave; the three weight images (128 KB) live in LDS, staged once by a persistent
//     workgroup of eight waves (two per SIMD).
constexpr int kApsWaves = 8;
// C1S: 16-channel steps of the query features (c1 = 64 or 32; 32: the FP_SA blocks, no residual); CF: trailing 64 -> 128
// conv (cov_final), its weight image read from global memory / L2 like M
// NOB: 32-channel blocks of the block's output (cout = 64, or 128: no residual, no trailing conv)
// (CF as an int: 32-channel blocks of the trailing conv's output, 0 = none -- 64 -> 128 and the first FP_SA block's 32 -> 64)
// ND: d_model / 32 (2: d = 64; 1: d = 32, the SA1 self-attention)
// POOL (round 6; the gallery's stage-2 launch): the block's output is not stored -- the wave keeps ALL blocks of a virtual
// cloud, reduces every block's 32 tokens per channel with a TRANSPOSING butterfly (five exchange steps; a lane keeps half
// of its registers per step, so the steps cost 16 + 8 + 4 + 2 + 1 registers, not 5 x 32) and writes the cloud's
// per-channel maximum and sum: p.pool_out (B, 2, 64) -- 512 bytes per cloud instead of 32 KB that pool_head would read back.
template <bool QPOS, int C1S, int CF, int NOB = 2, int ND = 2, bool POOL = false>
__global__ __launch_bounds__(64 * kApsWaves) __attribute__((amdgpu_waves_per_eu(2, 2)))
void attn_apply_stream64_kernel(AttnArgs a) {
  static_assert(!QPOS || C1S == 2 * ND, "q_pos needs c1 == d");
  static_assert(NOB <= 2 || (NOB == 4 && !CF), "cout = 128 has no trailing conv");
  static_assert(!POOL || (!CF && NOB == 2), "pooled output: cout = 64, no trailing conv");
  constexpr int D = 32 * ND, SD = 2 * ND;   // d_model, its 16-channel steps
  constexpr int SQ = C1S + (QPOS ? SD : 0), S0 = C1S + SD, NX = 8 * C1S;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const pcr_attn_params &p = a.p;
  bf16x8 *s_wq = reinterpret_cast<bf16x8 *>(smem);   // [SQ][ND cb][hi, lo][64]
  bf16x8 *s_w0 = s_wq + SQ * ND * 128;               // [S0][2 ND][2][64]
  bf16x8 *s_w2 = s_w0 + S0 * ND * 256;               // [4 ND][NOB][2][64]
  float *s_c = reinterpret_cast<float *>(s_w2 + ND * NOB * 512);   // bq | ln1 g | ln1 b : 3 x 64 | ln2 g | ln2 b : 2 x 32 NOB | (bfinal 128)
  f32x4 *s_p0 = reinterpret_cast<f32x4 *>(s_c + 448);    // [64] {w0x, w0y, w0z, b0}
  float *s_ks = reinterpret_cast<float *>(s_p0 + 64);    // [waves][64] key sums of the wave's current cloud
  const int tid = threadIdx.x, lane = tid & 63, j = lane & 31, h = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  {
    // (c1 not a multiple of 16: the padded image of mlp[0]; the packer pads wq's single step with zero columns itself)
    const f32x4 *wq = reinterpret_cast<const f32x4 *>(p.wq),
                *w0 = reinterpret_cast<const f32x4 *>((p.c1 & 15) ? p.wmlp0_bf_xpad : p.wmlp0),
                *w2 = reinterpret_cast<const f32x4 *>(p.wmlp2);
    f32x4 *dq = reinterpret_cast<f32x4 *>(s_wq), *d0 = reinterpret_cast<f32x4 *>(s_w0), *d2 = reinterpret_cast<f32x4 *>(s_w2);
    for (int e = tid; e < SQ * ND * 128; e += 64 * kApsWaves) dq[e] = wq[e];
    for (int e = tid; e < S0 * ND * 256; e += 64 * kApsWaves) d0[e] = w0[e];
    for (int e = tid; e < ND * NOB * 512; e += 64 * kApsWaves) d2[e] = w2[e];
    if (tid < D) {
      s_c[tid] = p.bq[tid];
      s_c[64 + tid] = p.ln1_g[tid];
      s_c[128 + tid] = p.ln1_b[tid];

      if (QPOS) s_p0[tid] = f32x4{p.pos0_w[3 * tid], p.pos0_w[3 * tid + 1], p.pos0_w[3 * tid + 2], p.pos0_b[tid]};
    }
    if (tid < 32 * NOB) {
      s_c[192 + tid] = p.ln2_g[tid];
      s_c[192 + 32 * NOB + tid] = p.ln2_b[tid];
    }
    if (CF && tid < 32 * CF) s_c[320 + tid] = p.bfinal[tid];
  }
  __syncthreads();
  const int nblk = p.Lq >> 5;
  const long nitem = (long)p.B * nblk;
  const int dh = D / p.nhead;
  const float skf = (float)p.Sk;
  float *ksw = s_ks + wave * 64;
  // constants of the lane's registers (cout = 32 cb + 8 g + 4 h + q): 16-byte reads at 32 cb + 8 g + 4 h
  auto cvec = [&](const float *base, int cb, int g) __attribute__((always_inline)) {
    return *reinterpret_cast<const f32x4 *>(base + 32 * cb + 8 * g + 4 * h);
  };
  auto to_ops = [&](const f32x16 &acc, int G, bf16x8 &oh, bf16x8 &ol) __attribute__((always_inline)) {
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; e++) v[e] = acc[8 * G + e];
    bf_split8(v, oh, ol, true);
  };
#ifdef PCR_SA_TRACE_BUILD
  const bool tracing = (a.dbg & 256) && lane == 0 && (wave == 0 || wave == 5) && blockIdx.x < kATraceWgs;
  unsigned long long *trace = g_attn_trace + (size_t)(2 * blockIdx.x + (wave ? 1 : 0)) * (kATraceRecs * kATraceMarks);
  int trace_it = 0;
#define PCR_AMARK(m)                                                                                       \
  do {                                                                                                     \
    if (tracing && trace_it < kATraceRecs) trace[trace_it * kATraceMarks + (m)] = __builtin_readcyclecounter(); \
  } while (0)
#define PCR_ANEXT() trace_it++
#else
#define PCR_AMARK(m) do { } while (0)
#define PCR_ANEXT() do { } while (0)
#endif
  // item walk: (cloud, block) dealt block by block over the waves -- or, POOL, cloud by cloud (all blocks of a cloud on one
  // wave, in order: the running maximum / sum of the cloud live in two registers)
  const long nwav = (long)gridDim.x * kApsWaves, gw = (long)blockIdx.x * kApsWaves + wave;
  long pc = gw;
  int pblk = 0;
  float pmax = -INFINITY, psum = 0.f;
  for (long it = gw; POOL ? pc < p.B : it < nitem; it += POOL ? 0 : nwav) {
    asm volatile("" ::: "memory");   // (weight reads stay inside the item loop)
    PCR_AMARK(0);
    const long b = POOL ? pc : it / nblk;
    const int blk = POOL ? pblk : (int)(it - b * nblk);
    const size_t bq_ = p.q_index ? (size_t)p.q_index[b] : (size_t)b;
    const size_t kb_ = p.kv_index ? (size_t)p.kv_index[b] : (size_t)b;
    const float *kvp = p.kv + kb_ * ((size_t)D * D + D);
    const int t = blk * 32 + j;
    const __amdgpu_buffer_rsrc_t rfeat = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(p.feat_q + bq_ * p.c1 * p.Lq), 0, p.c1 * p.Lq * 4, 0x00020000);   // (channels past c1 read as 0)
    const int vo = (4 * h * p.Lq + t) * 4;
    float xf[NX];
#pragma unroll
    for (int e = 0; e < NX; e++) {
      const int ch = 16 * (e >> 3) + bf_kpos(0, e & 7);   // + 4 h: in the lane offset
      xf[e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rfeat, vo, ch * p.Lq * 4, 0));
    }
    if (lane < D) ksw[lane] = kvp[(size_t)D * D + lane];
    bf16x8 bh[4 * ND], bl[4 * ND];
    if constexpr (QPOS) {
      const float *xyz = p.xyz_q + (bq_ * p.Lq + t) * 3;
      const float px = xyz[0], py = xyz[1], pz = xyz[2];
#pragma unroll
      for (int s2 = 0; s2 < SD; s2++) {
        float hv[8];
#pragma unroll
        for (int e = 0; e < 8; e++) {
          const f32x4 w = s_p0[16 * s2 + bf_kpos(0, e) + 4 * h];
          const float v = w[0] * px + w[1] * py + w[2] * pz + w[3];
          hv[e] = relu_i(v);
        }
        bf_split8(hv, bh[C1S + s2], bl[C1S + s2], true);
      }
    }
#pragma unroll
    for (int e = 0; e < NX; e++) asm volatile("" : "+v"(xf[e]));   // (the loads land here: see attn_kv_stream64_kernel)
#pragma unroll
    for (int s2 = 0; s2 < C1S; s2++) {
      float xv[8];
#pragma unroll
      for (int e = 0; e < 8; e++) xv[e] = xf[8 * s2 + e];
      bf_split8(xv, bh[s2], bl[s2], true);
    }
    PCR_AMARK(1);
    // ---- Q = elu(Wq [x ; h] + bq) + 1
    f32x16 q[ND];
#pragma unroll
    for (int cb = 0; cb < ND; cb++)
#pragma unroll
      for (int r = 0; r < 16; r++) q[cb][r] = 0.f;
    {
      const bf16x8 *wb = s_wq + lane;
#pragma unroll
      for (int s2 = 0; s2 < SQ; s2++) {
        bf16x8 wh[ND], wl[ND];
#pragma unroll
        for (int cb = 0; cb < ND; cb++) {
          wh[cb] = wb[((s2 * ND + cb) * 2) * 64];
          wl[cb] = wb[((s2 * ND + cb) * 2 + 1) * 64];
        }
#pragma unroll
        for (int cb = 0; cb < ND; cb++) q[cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh[cb], bh[s2], q[cb], 0, 0, 0);
#pragma unroll
        for (int cb = 0; cb < ND; cb++) q[cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh[cb], bl[s2], q[cb], 0, 0, 0);
#pragma unroll
        for (int cb = 0; cb < ND; cb++) q[cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wl[cb], bh[s2], q[cb], 0, 0, 0);
      }
    }
    PCR_AMARK(2);
    // the message-phase A operands (the cloud's matrix M): requested now, used after the normaliser
    bf16x8 mh[SD][ND], ml[SD][ND];
    {
      const bf16x8 *mb = reinterpret_cast<const bf16x8 *>(kvp) + lane;
#pragma unroll
      for (int s2 = 0; s2 < SD; s2++)
#pragma unroll
        for (int cb = 0; cb < ND; cb++) {
          mh[s2][cb] = mb[((s2 * ND + cb) * 2) * 64];
          ml[s2][cb] = mb[((s2 * ND + cb) * 2 + 1) * 64];
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // (the wave's own key-sum strip: written above, read below)
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // z[head][token] = Q_head . ksum_head; partial sums per 16-channel group (cb, G), then by head width
    float z16[2][2] = {{0.f, 0.f}, {0.f, 0.f}};
#pragma unroll
    for (int cb = 0; cb < ND; cb++)
#pragma unroll
      for (int g = 0; g < 4; g++) {
        const f32x4 bqv = cvec(s_c, cb, g), ksv = cvec(ksw, cb, g);
        float zz = 0.f;
#pragma unroll
        for (int qq = 0; qq < 4; qq++) {
          const float v = elu1(q[cb][4 * g + qq] + bqv[qq]);
          q[cb][4 * g + qq] = v;
          zz += v * ksv[qq];
        }
        if ((g & 1) == 0) z16[cb][g >> 1] = zz;
        else z16[cb][g >> 1] += zz;
      }
    float zs[2][2];
    {
      float za = z16[0][0], zb = z16[0][1], zc = z16[1][0], zd = z16[1][1];
      if (dh >= 32) {
        za += zb; zb = za;
        zc += zd; zd = zc;
      }
      if (ND == 2 && dh >= 64) {
        za += zc; zb = za; zc = za; zd = za;
      }
      za += __shfl_xor(za, 32, 64);
      zb += __shfl_xor(zb, 32, 64);
      zc += __shfl_xor(zc, 32, 64);
      zd += __shfl_xor(zd, 32, 64);
      zs[0][0] = (1.0f / (za + 1e-6f)) * skf;
      zs[0][1] = (1.0f / (zb + 1e-6f)) * skf;
      zs[1][0] = (1.0f / (zc + 1e-6f)) * skf;
      zs[1][1] = (1.0f / (zd + 1e-6f)) * skf;
    }
#pragma unroll
    for (int cb = 0; cb < ND; cb++)
#pragma unroll
      for (int r = 0; r < 16; r++) q[cb][r] *= zs[cb][r >> 3];
    // ---- message = M Q'
    bf16x8 ch_[SD], cl_[SD];
#pragma unroll
    for (int cb = 0; cb < ND; cb++)
#pragma unroll
      for (int G = 0; G < 2; G++) to_ops(q[cb], G, ch_[2 * cb + G], cl_[2 * cb + G]);
    f32x16 m[ND];
#pragma unroll
    for (int cb = 0; cb < ND; cb++)
#pragma unroll
      for (int r = 0; r < 16; r++) m[cb][r] = 0.f;
#pragma unroll
    for (int s2 = 0; s2 < SD; s2++) {
#pragma unroll
      for (int cb = 0; cb < ND; cb++) m[cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(mh[s2][cb], ch_[s2], m[cb], 0, 0, 0);
#pragma unroll
      for (int cb = 0; cb < ND; cb++) m[cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(mh[s2][cb], cl_[s2], m[cb], 0, 0, 0);
#pragma unroll
      for (int cb = 0; cb < ND; cb++) m[cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ml[s2][cb], ch_[s2], m[cb], 0, 0, 0);
    }
    // ---- LayerNorm over the 64 channels of a token (two passes, eps inside the sqrt: tile_layernorm's arithmetic)
    auto layernorm = [&](auto &v, auto nb_tag, const float *gam, const float *bet) __attribute__((always_inline)) {
      constexpr int NB = decltype(nb_tag)::value;
      float sm = 0.f;
#pragma unroll
      for (int cb = 0; cb < NB; cb++)
#pragma unroll
        for (int r = 0; r < 16; r++) sm += v[cb][r];
      sm += __shfl_xor(sm, 32, 64);
      const float mean = sm / (32.0f * NB);
      float vr = 0.f;
#pragma unroll
      for (int cb = 0; cb < NB; cb++)
#pragma unroll
        for (int r = 0; r < 16; r++) {
          const float dlt = v[cb][r] - mean;
          vr += dlt * dlt;
        }
      vr += __shfl_xor(vr, 32, 64);
      const float inv = 1.0f / sqrtf(vr / (32.0f * NB) + 1e-5f);
#pragma unroll
      for (int cb = 0; cb < NB; cb++)
#pragma unroll
        for (int g = 0; g < 4; g++) {
          const f32x4 gv = cvec(gam, cb, g), bv = cvec(bet, cb, g);
#pragma unroll
          for (int qq = 0; qq < 4; qq++) v[cb][4 * g + qq] = (v[cb][4 * g + qq] - mean) * inv * gv[qq] + bv[qq];
        }
    };
    PCR_AMARK(3);
    layernorm(m, std::integral_constant<int, ND>{}, s_c + 64, s_c + 128);
    PCR_AMARK(4);
    // ---- FFN0: relu(W0 [x ; msg]) (128 couts), operands: x re-converted from its f32 registers, msg from m
#pragma unroll
    for (int s2 = 0; s2 < C1S; s2++) {
      float xv[8];
#pragma unroll
      for (int e = 0; e < 8; e++) xv[e] = xf[8 * s2 + e];
      bf_split8(xv, bh[s2], bl[s2], true);
    }
#pragma unroll
    for (int cb = 0; cb < ND; cb++)
#pragma unroll
      for (int G = 0; G < 2; G++) to_ops(m[cb], G, bh[C1S + 2 * cb + G], bl[C1S + 2 * cb + G]);
    f32x16 f[SD];
#pragma unroll
    for (int cb = 0; cb < SD; cb++)
#pragma unroll
      for (int r = 0; r < 16; r++) f[cb][r] = 0.f;
    {
      const bf16x8 *wb = s_w0 + lane;
#pragma unroll
      for (int s2 = 0; s2 < S0; s2++) {
        bf16x8 wh[SD], wl[SD];
#pragma unroll
        for (int cb = 0; cb < SD; cb++) {
          wh[cb] = wb[((s2 * SD + cb) * 2) * 64];
          wl[cb] = wb[((s2 * SD + cb) * 2 + 1) * 64];
        }
#pragma unroll
        for (int cb = 0; cb < SD; cb++) f[cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh[cb], bh[s2], f[cb], 0, 0, 0);
#pragma unroll
        for (int cb = 0; cb < SD; cb++) f[cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh[cb], bl[s2], f[cb], 0, 0, 0);
#pragma unroll
        for (int cb = 0; cb < SD; cb++) f[cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wl[cb], bh[s2], f[cb], 0, 0, 0);
      }
    }
#pragma unroll
    for (int cb = 0; cb < SD; cb++) {
#pragma unroll
      for (int r = 0; r < 16; r++) f[cb][r] = relu_i(f[cb][r]);
#pragma unroll
      for (int G = 0; G < 2; G++) to_ops(f[cb], G, bh[2 * cb + G], bl[2 * cb + G]);
    }
    PCR_AMARK(5);
    // ---- FFN1 (64 couts), LayerNorm, residual, store
    f32x16 o[NOB];
#pragma unroll
    for (int cb = 0; cb < NOB; cb++)
#pragma unroll
      for (int r = 0; r < 16; r++) o[cb][r] = 0.f;
    {
      const bf16x8 *wb = s_w2 + lane;
#pragma unroll
      for (int s2 = 0; s2 < 4 * ND; s2++) {
        bf16x8 wh[NOB], wl[NOB];
#pragma unroll
        for (int cb = 0; cb < NOB; cb++) {
          wh[cb] = wb[((s2 * NOB + cb) * 2) * 64];
          wl[cb] = wb[((s2 * NOB + cb) * 2 + 1) * 64];
        }
#pragma unroll
        for (int cb = 0; cb < NOB; cb++) o[cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh[cb], bh[s2], o[cb], 0, 0, 0);
#pragma unroll
        for (int cb = 0; cb < NOB; cb++) o[cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh[cb], bl[s2], o[cb], 0, 0, 0);
#pragma unroll
        for (int cb = 0; cb < NOB; cb++) o[cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wl[cb], bh[s2], o[cb], 0, 0, 0);
      }
    }
    PCR_AMARK(6);
    layernorm(o, std::integral_constant<int, NOB>{}, s_c + 192, s_c + 192 + 32 * NOB);
    if constexpr (C1S == 2 * NOB) {   // cout == c1
      if (p.residual) {
#pragma unroll
        for (int cb = 0; cb < NOB; cb++)
#pragma unroll
          for (int r = 0; r < 16; r++) o[cb][r] += xf[16 * cb + r];
      }
    }
    if constexpr (POOL) {
      // lane (j, h) holds the block's token j in the 32 channels ch(e) = 16 (e >> 3) + bf_kpos(h, e & 7), e = 0 .. 31.  Step s
      // exchanges with lane j ^ (1 << s): the lane with bit s set keeps the upper half of its register list, its partner
      // the lower half (t = what I keep, u = what my partner keeps), so after five steps lane j holds the maximum and the
      // sum over the 32 tokens of ONE channel: e = bitreverse5(j).  (Sums: a fixed pairwise tree per block, blocks in order.)
      float vm[16], vs[16];
      {
        const bool up = (j & 1) != 0;
#pragma unroll
        for (int r = 0; r < 16; r++) {
          const float lo_ = o[r >> 4][r & 15], hi_ = o[(r + 16) >> 4][(r + 16) & 15];
          const float tk = up ? hi_ : lo_, uk = up ? lo_ : hi_;
          const float pu = __uint_as_float(pcr_sort_partner<2, 1>(__float_as_uint(uk), lane));
          vm[r] = fmaxf(tk, pu);
          vs[r] = tk + pu;
        }
      }
#define PCR_POOL_STEP(N, J, BIT)                                                                      \
      {                                                                                               \
        const bool up = (j & BIT) != 0;                                                               \
        _Pragma("unroll") for (int r = 0; r < N; r++) {                                               \
          const float tm = up ? vm[r + N] : vm[r], um = up ? vm[r] : vm[r + N];                       \
          const float ts = up ? vs[r + N] : vs[r], us = up ? vs[r] : vs[r + N];                       \
          vm[r] = fmaxf(tm, __uint_as_float(pcr_sort_partner<2 * J, J>(__float_as_uint(um), lane)));  \
          vs[r] = ts + __uint_as_float(pcr_sort_partner<2 * J, J>(__float_as_uint(us), lane));        \
        }                                                                                             \
      }
      PCR_POOL_STEP(8, 2, 2)
      PCR_POOL_STEP(4, 4, 4)
      PCR_POOL_STEP(2, 8, 8)
      PCR_POOL_STEP(1, 16, 16)
#undef PCR_POOL_STEP
      pmax = fmaxf(pmax, vm[0]);
      psum += vs[0];
      if (++pblk == nblk) {
        const int e = ((j & 1) << 4) | ((j & 2) << 2) | (j & 4) | ((j & 8) >> 2) | ((j & 16) >> 4);
        const int ch = 16 * (e >> 3) + bf_kpos(h, e & 7);
        float *po = p.pool_out + (size_t)b * 128;
        po[ch] = pmax;
        po[64 + ch] = psum;
        pmax = -INFINITY;
        psum = 0.f;
        pblk = 0;
        pc += nwav;
      }
    } else if constexpr (!CF) {
      const __amdgpu_buffer_rsrc_t rout =
          __builtin_amdgcn_make_buffer_rsrc(p.out + (size_t)b * (32 * NOB) * p.Lq, 0, 32 * NOB * p.Lq * 4, 0x00020000);
#pragma unroll
      for (int e = 0; e < 16 * NOB; e++) {
        const int ch = 16 * (e >> 3) + bf_kpos(0, e & 7);
        const float v = o[e >> 4][e & 15];   // (a copy: __builtin_bit_cast of a vector ELEMENT reads element 0)
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rout, vo, ch * p.Lq * 4, 0);
      }
    } else {
      // ---- cov_final (32 NOB -> 32 CF channels, bias), A operands from the global image
#pragma unroll
      for (int cb = 0; cb < NOB; cb++)
#pragma unroll
        for (int G = 0; G < 2; G++) to_ops(o[cb], G, bh[2 * cb + G], bl[2 * cb + G]);
      f32x16 fo[CF ? CF : 1];
#pragma unroll
      for (int cb = 0; cb < CF; cb++)
#pragma unroll
        for (int r = 0; r < 16; r++) fo[cb][r] = 0.f;
      const bf16x8 *wb = reinterpret_cast<const bf16x8 *>(p.wfinal) + lane;
#pragma unroll
      for (int s2 = 0; s2 < 2 * NOB; s2++) {
        bf16x8 wh[CF ? CF : 1], wl[CF ? CF : 1];
#pragma unroll
        for (int cb = 0; cb < CF; cb++) {
          wh[cb] = wb[((s2 * CF + cb) * 2) * 64];
          wl[cb] = wb[((s2 * CF + cb) * 2 + 1) * 64];
        }
#pragma unroll
        for (int cb = 0; cb < CF; cb++) fo[cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh[cb], bh[s2], fo[cb], 0, 0, 0);
#pragma unroll
        for (int cb = 0; cb < CF; cb++) fo[cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh[cb], bl[s2], fo[cb], 0, 0, 0);
#pragma unroll
        for (int cb = 0; cb < CF; cb++) fo[cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wl[cb], bh[s2], fo[cb], 0, 0, 0);
      }
      const __amdgpu_buffer_rsrc_t rout =
          __builtin_amdgcn_make_buffer_rsrc(p.out + (size_t)b * (32 * CF) * p.Lq, 0, 32 * CF * p.Lq * 4, 0x00020000);
#pragma unroll
      for (int cb = 0; cb < CF; cb++)
#pragma unroll
        for (int g = 0; g < 4; g++) {
          const f32x4 bv = cvec(s_c + 320, cb, g);
#pragma unroll
          for (int qq = 0; qq < 4; qq++) {
            const int e = 16 * cb + 4 * g + qq;
            const int ch = 16 * (e >> 3) + bf_kpos(0, e & 7);
            const float v = fo[cb][4 * g + qq] + bv[qq];
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rout, vo, ch * p.Lq * 4, 0);
          }
        }
    }
    __builtin_amdgcn_wave_barrier();   // (the key-sum strip is rewritten by the next item)
    PCR_AMARK(7);
    PCR_ANEXT();
  }
#undef PCR_AMARK
#undef PCR_ANEXT
}
#endif

}  // namespace

static int attn_check(const pcr_attn_params &p) {
  if (p.B < 0 || p.Lq < 1 || p.Sk < 1 || p.c1 < 1 || p.c2 < 1 || p.cout < 1 || p.nhead < 1) return 1;
  if (p.d < 32 || p.d > 512 || (p.d & 31) || p.d % p.nhead) return 1;  // d_model in {32,64,96,128} or wide (below)
  if (p.d > 128 && ((p.d & 63) || ((p.d / p.nhead) & 63) || p.d / p.nhead > 256)) return 1;   // wide: heads of 64 n <= 256
  if ((p.c2 & 7) || p.cout > 512 || p.cfinal > 512) return 1;
  if (!p.feat_q || !p.feat_k || !p.xyz_k || !p.kv || !p.pos0_w || !p.pos0_b || !p.wq || !p.bq || !p.wkv ||
      !p.bkv || !p.wmerge || !p.wmlp0 || !p.wmlp2 || !p.ln1_g || !p.ln1_b || !p.ln2_g || !p.ln2_b)
    return 1;
  if (p.q_pos && (!p.xyz_q || p.c1 != p.c2 || p.c1 != p.d)) return 1;
  if (p.residual && p.cout != p.c1) return 1;
  if (p.cfinal && (!p.wfinal || !p.bfinal || (p.cout & 7))) return 1;
  return 0;
}

static void attn_dump_trace(const char *path, const char *tag, int wgs, int B, int Sk) {
  static int launches = 0;
  static const int skip = pcr_tune_int("PCR_SA_TRACE_SKIP");
  launches++;
  if (launches <= skip || launches > skip + 8) return;
  static std::vector<unsigned long long> host(2 * kATraceWgs * kATraceRecs * kATraceMarks);
  if (hipDeviceSynchronize() != hipSuccess) return;
  if (hipMemcpyFromSymbol(host.data(), HIP_SYMBOL(g_attn_trace), host.size() * sizeof(unsigned long long)) != hipSuccess) return;
  FILE *f = fopen(path, "a");
  if (!f) return;
  const int n = 2 * (wgs < kATraceWgs ? wgs : kATraceWgs);
  fprintf(f, "launch %d kernel %s wgs %d B %d Sk %d\n", launches, tag, wgs, B, Sk);
  for (int w = 0; w < n; w++) {
    const unsigned long long *t = host.data() + (size_t)w * (kATraceRecs * kATraceMarks);
    fprintf(f, "wg %d", w);
    for (int i = 0; i < kATraceRecs * kATraceMarks; i++) fprintf(f, " %llu", t[i]);
    fprintf(f, "\n");
  }
  fclose(f);
}

// d_model <= 128: one workgroup per key-side cloud (both precisions; wq / wkv / ... are images of THIS unit's kind)
static int attn_kv_narrow(const pcr_attn_params *pp, pcr_stream_t stream) {
  AttnArgs a;
  a.p = *pp;
  a.dbg = 0;
  const int d = pp->d;
  if (pp->c2 < pp->d) return PCR_ERR_INVALID;   // the in-place K/V projection needs 2d <= c2 + d rows
  const int tb = d <= 64 ? 2 : 1, RP = 32 * tb + 1;
  size_t lds = (size_t)(pp->c2 + d + 3) * RP + 4 * d;
  const size_t lds2 = (size_t)d * (d + 1) + d;
  if (lds2 > lds) lds = lds2;
  lds = (lds + kThreads) * sizeof(float);
  if (lds > (size_t)kMaxDynLds) return PCR_ERR_INVALID;
  static bool ok = allow_big_lds(attn_kv_kernel<2, 1, 2, 1>) && allow_big_lds(attn_kv_kernel_o3<2, 1, 1, 1>) &&
                   allow_big_lds(attn_kv_kernel_o3<1, 2, 1, 4>) && allow_big_lds(attn_kv_kernel<1, 2, 0, 4>);
  (void)ok;
  const int ntile = (pp->Sk + 32 * tb - 1) / (32 * tb);
  int ns = pp->kv_splits > 1 ? pp->kv_splits : 1;
  if (ns > 1 && (!pp->kv_part || ns > 64)) return PCR_ERR_INVALID;
  if (ns > ntile) ns = ntile;                 // (kv_part is sized for the requested count: fewer splits use its head)
  a.p.kv_splits = ns;
  dim3 g(pp->B, ns), blk(kThreads);
  hipStream_t st = pcr_s(stream);
  constexpr bool kBfUnit = kAPrec != 0;
  const bool bfk = kBfUnit && pp->wkv_bf != nullptr;
  if (ns == 1 && d == 64 && (pp->c2 == 64 || (pp->c2 == 128 && bfk && pp->Sk >= 128)) && (pp->Sk & 31) == 0 && pp->nhead >= 1 &&
      64 % pp->nhead == 0) {
    // wave-autonomous form (shape-only choice; an explicit token split keeps the tile kernel)
    const bool bf = bfk;
    const bool wide = pp->c2 == 128;
    pcr_note_arith(bf ? PCR_PREC_BF16X3 : PCR_PREC_F32);   // (the projection; the KV accumulation and the fold are f32)
    static bool oks = allow_big_lds(attn_kv_stream64_kernel<true, false>) && allow_big_lds(attn_kv_stream64_kernel<false, false>) &&
                      allow_big_lds(attn_kv_stream64_kernel<true, kBfUnit>) && allow_big_lds(attn_kv_stream64_kernel<false, kBfUnit>) &&
                      allow_big_lds(attn_kv_stream64_kernel<true, kBfUnit, kBfUnit ? 8 : 4>) &&
                      allow_big_lds(attn_kv_stream64_kernel<false, kBfUnit, kBfUnit ? 8 : 4>);
    (void)oks;
    const int nblk = pp->Sk >> 5;
    const int minw = wide ? 4 : 2;
    int wpc2 = 1;
    while (wpc2 * 2 <= (nblk < 4 ? nblk : 4)) wpc2 *= 2;      // (at most four waves per cloud: the kernel's comment)
    if (wpc2 < minw) wpc2 = minw;
    // short key sets: one wave per cloud (ONEW; shape-only, the bf16 unit's image only)
    static const int no_onew = pcr_tune_int("PCR_ATTN_NO_ONEW");   // diagnostics
    const bool onew = kBfUnit && bf && !wide && nblk <= 4 && !no_onew;
    if (onew) wpc2 = 1;
    const int cpg = kKvsWaves / wpc2;
    const size_t wu = bf ? (size_t)((wide ? 8 : 4) + 4) * 512 : 4096;
    const size_t lds_s = (wu * 4 + 256 + 128 + 64 * 65 + (onew ? (size_t)0 : (size_t)cpg * (64 * 65 + 64))) * sizeof(float);
    const long rounds = ((long)pp->B + cpg - 1) / cpg;
    static const int ncu = [] {
      hipDeviceProp_t pr;
      int dev = 0;
      if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&pr, dev) != hipSuccess) return 256;
      return pr.multiProcessorCount > 0 ? pr.multiProcessorCount : 256;
    }();
    const int gs = (int)(rounds < ncu ? rounds : ncu);        // persistent: one workgroup per CU
    const dim3 gg(gs), bb(64 * kKvsWaves);
    if (bf && wide) {
      if (pp->nhead >= 2) hipLaunchKernelGGL((attn_kv_stream64_kernel<true, kBfUnit, kBfUnit ? 8 : 4>), gg, bb, lds_s, st, a);
      else hipLaunchKernelGGL((attn_kv_stream64_kernel<false, kBfUnit, kBfUnit ? 8 : 4>), gg, bb, lds_s, st, a);
    } else if (bf) {
      static const char *atrace = pcr_tune_str("PCR_ATTN_TRACE");
      if (atrace) a.dbg |= 256;
      if (onew) {
        static bool oko = allow_big_lds(attn_kv_stream64_kernel<true, kBfUnit, 4, kBfUnit>) &&
                          allow_big_lds(attn_kv_stream64_kernel<false, kBfUnit, 4, kBfUnit>);
        (void)oko;
        if (pp->nhead >= 2) hipLaunchKernelGGL((attn_kv_stream64_kernel<true, kBfUnit, 4, kBfUnit>), gg, bb, lds_s, st, a);
        else hipLaunchKernelGGL((attn_kv_stream64_kernel<false, kBfUnit, 4, kBfUnit>), gg, bb, lds_s, st, a);
      } else if (pp->nhead >= 2) hipLaunchKernelGGL((attn_kv_stream64_kernel<true, kBfUnit>), gg, bb, lds_s, st, a);
      else hipLaunchKernelGGL((attn_kv_stream64_kernel<false, kBfUnit>), gg, bb, lds_s, st, a);
      if (atrace) attn_dump_trace(atrace, "kv64", gs, (int)pp->B, pp->Sk);
    } else {
      if (pp->nhead >= 2) hipLaunchKernelGGL((attn_kv_stream64_kernel<true, false>), gg, bb, lds_s, st, a);
      else hipLaunchKernelGGL((attn_kv_stream64_kernel<false, false>), gg, bb, lds_s, st, a);
    }
    PCR_CHECK_LAUNCH();
    return PCR_OK;
  }
#if PCR_ATTN_PREC != 0
  if (ns == 1 && d == 32 && pp->c2 == 32 && pp->wkv_bf && (pp->Sk & 31) == 0 && pp->nhead >= 1 && 32 % pp->nhead == 0) {
    static bool ok32 = allow_big_lds(attn_kv_stream32_kernel);
    (void)ok32;
    pcr_note_arith(PCR_PREC_BF16X3);
    const int nblk = pp->Sk >> 5;
    int wpc2 = 1;
    while (wpc2 * 2 <= (nblk < kKvsWaves ? nblk : kKvsWaves)) wpc2 *= 2;
    if (wpc2 < 2) wpc2 = 2;
    const int cpg = kKvsWaves / wpc2;
    const size_t lds_s = (size_t)(4096 + 128 + 64 + 32 * 33 + cpg * (32 * 33 + 32)) * sizeof(float);
    const long rounds = ((long)pp->B + cpg - 1) / cpg;
    static const int ncu = [] {
      hipDeviceProp_t pr;
      int dev = 0;
      if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&pr, dev) != hipSuccess) return 256;
      return pr.multiProcessorCount > 0 ? pr.multiProcessorCount : 256;
    }();
    const long cap = 2L * ncu;                                // (small footprint: two workgroups per CU)
    hipLaunchKernelGGL(attn_kv_stream32_kernel, dim3((unsigned)(rounds < cap ? rounds : cap)), dim3(64 * kKvsWaves), lds_s, st, a);
    PCR_CHECK_LAUNCH();
    return PCR_OK;
  }
#endif
  pcr_note_arith(PCR_PREC_F32);   // the tile kernel projects in f32 in both units (only the form of M differs)
  if (d == 32) hipLaunchKernelGGL((attn_kv_kernel<2, 1, 2, 1>), g, blk, lds, st, a);        // 2d = 64: two cout blocks
  else if (d == 64) hipLaunchKernelGGL((attn_kv_kernel_o3<2, 1, 1, 1>), g, blk, lds, st, a);   // four, one per wave
  else if (d == 128) {
    // eight cout blocks, two rounds.  NOT the three-workgroups-per-CU form: this shape's fold buffer (d (d + 1) floats =
    // 66 KB) allows two workgroups per CU whatever the registers say, and held to a third of the register file the body
    // spilled 93 registers inside its tile loop for nothing (round 5, tools/kres.py)
    static bool ok2 = allow_big_lds(attn_kv_kernel_o2<1, 2, 1, 4>) && allow_big_lds(attn_kv_kernel_o2<1, 2, 1, 4, true>);
    (void)ok2;
    // (split-bf16 projection for key sets of >= 256 tokens -- a shape-only rule: the KV state is a mean over the key
    // tokens, so the projection's rounding averages out with their number; at Sk = 32 (the Point-Transformer @128) the
    // guard's sweep of tests/test_gpu_precision.py went from 4.5e-5 to 5.2e-5 with it, for 0.014 ms)
    if (kBfUnit && pp->wkv_bf && pp->Sk >= 256) hipLaunchKernelGGL((attn_kv_kernel_o2<1, 2, 1, 4, true>), g, blk, lds, st, a);
    else hipLaunchKernelGGL((attn_kv_kernel_o2<1, 2, 1, 4>), g, blk, lds, st, a);
  }
  else hipLaunchKernelGGL((attn_kv_kernel<1, 2, 0, 4>), g, blk, lds, st, a);                // d = 96: generic shape
  PCR_CHECK_LAUNCH();
  if (ns > 1) {
    static bool okf = allow_big_lds(attn_kv_fold_kernel);
    (void)okf;
    hipLaunchKernelGGL(attn_kv_fold_kernel, dim3(pp->B), blk, ((size_t)d * (d + 1) + d) * sizeof(float), st, a);
    PCR_CHECK_LAUNCH();
  }
  return PCR_OK;
}

static int attn_apply_launch(const pcr_attn_params *pp, pcr_stream_t stream) {
  const pcr_attn_params &p = *pp;
  AttnArgs a;
  a.p = p;
  a.dbg = 0;
  // (round 6, measured and dropped: 64-token tiles for d = 128 -- half the weight traffic from L2, half the waves per CU:
  // pt1024's attn_apply[d=128] 0.455 -> 0.518 ms, profiles/r06_inproc_ab.txt)
  const int tb = p.d <= 32 ? 4 : (p.d <= 64 ? 2 : 1), T = 32 * tb, RP = T + 1;
  const int catP = ceil8(p.c1 + p.d);
  int rowsU = catP > 2 * p.d ? catP : 2 * p.d;
  if (ceil32(p.cout) > rowsU) rowsU = ceil32(p.cout);
  if (ceil32(p.cfinal) > rowsU) rowsU = ceil32(p.cfinal);
  size_t lds = ((size_t)(rowsU + 3 + p.nhead) * RP + 2 * (kThreads / T) * T + 7 * p.d + 2 * p.cout) *
               sizeof(float);
  if (lds > (size_t)kMaxDynLds) return PCR_ERR_INVALID;
  dim3 g((p.Lq + T - 1) / T, p.B), blk(kThreads);
  hipStream_t st = pcr_s(stream);
  pcr_note_arith(kAPrec != 0 ? PCR_PREC_BF16X3 : PCR_PREC_F32);   // (the bf unit is split bf16 whatever was requested)
#if PCR_ATTN_PREC != 0
  const int aps_c1s = (p.c1 + 15) >> 4, aps_nd = p.d >> 5;
  const size_t aps_lds = (size_t)((aps_c1s + (p.q_pos ? 2 * aps_nd : 0)) * aps_nd * 128 + (aps_c1s + 2 * aps_nd) * aps_nd * 256 +
                                  aps_nd * (p.cout >> 5) * 512) * 16 +
                         (size_t)(448 + 256 + 64 * kApsWaves) * sizeof(float);
  if (p.d == 32 && p.c1 == 32 && p.cout == 32 && !p.cfinal && (p.Lq & 31) == 0 && (p.nhead == 1 || p.nhead == 2)) {
    // the SA1 self-attention (d_model 32): the same kernel with one 32-channel block
    const long nitem = (long)p.B * (p.Lq >> 5);
    const long nwg = (nitem + kApsWaves - 1) / kApsWaves;
    static const int ncu = [] {
      hipDeviceProp_t pr;
      int dev = 0;
      if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&pr, dev) != hipSuccess) return 256;
      return pr.multiProcessorCount > 0 ? pr.multiProcessorCount : 256;
    }();
    const dim3 gg((unsigned)(nwg < ncu ? nwg : ncu)), bb(64 * kApsWaves);
    static bool ok32 = allow_big_lds(attn_apply_stream64_kernel<true, 2, 0, 1, 1>) && allow_big_lds(attn_apply_stream64_kernel<false, 2, 0, 1, 1>);
    (void)ok32;
    if (p.q_pos) hipLaunchKernelGGL((attn_apply_stream64_kernel<true, 2, 0, 1, 1>), gg, bb, aps_lds, st, a);
    else hipLaunchKernelGGL((attn_apply_stream64_kernel<false, 2, 0, 1, 1>), gg, bb, aps_lds, st, a);
    PCR_CHECK_LAUNCH();
    return PCR_OK;
  }
  // shapes: (c1 = 64 | 32 | < 16 with the padded mlp[0] image) x (cout = 64 [+ cov_final 128] | cout = 128 | cout = 32 + cov_final 64)
  const bool aps_in = p.c1 == 64 || ((p.c1 == 32 || (p.c1 < 16 && pp->wmlp0_bf_xpad)) && !p.q_pos && !p.residual);
  const bool aps_out = (p.cout == 64 && (p.cfinal == 0 || p.cfinal == 128) && p.c1 >= 32) ||
                       (p.cout == 128 && !p.cfinal && !p.residual && p.c1 == 64) ||
                       (p.cout == 32 && p.cfinal == 64 && !p.residual && p.c1 < 16);
  if (p.d == 64 && aps_in && aps_out &&
      (p.Lq & 31) == 0 && (p.nhead == 1 || p.nhead == 2 || p.nhead == 4) && aps_lds <= (size_t)kMaxDynLds) {
    // wave-autonomous form (shape-only choice)
    const int c1s = aps_c1s, sq = c1s + (p.q_pos ? 4 : 0), s0 = c1s + 4;
    const int nob = p.cout >> 5;
    const size_t lds_s = (size_t)(sq * 256 + s0 * 512 + nob * 1024) * 16 + (size_t)(448 + 256 + 64 * kApsWaves) * sizeof(float);
    const long nitem = (long)p.B * (p.Lq >> 5);
    const long nwg = (nitem + kApsWaves - 1) / kApsWaves;
    static const int ncu = [] {
      hipDeviceProp_t pr;
      int dev = 0;
      if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&pr, dev) != hipSuccess) return 256;
      return pr.multiProcessorCount > 0 ? pr.multiProcessorCount : 256;
    }();
    const dim3 gg((unsigned)(nwg < ncu ? nwg : ncu)), bb(64 * kApsWaves);
#define PCR_APS(QP, C1Sv, CFv)                                                            \
  do {                                                                                    \
    static bool oks = allow_big_lds(attn_apply_stream64_kernel<QP, C1Sv, CFv>);           \
    (void)oks;                                                                            \
    hipLaunchKernelGGL((attn_apply_stream64_kernel<QP, C1Sv, CFv>), gg, bb, lds_s, st, a); \
  } while (0)
    const bool cf = p.cfinal != 0;
    static const char *aptrace = pcr_tune_str("PCR_ATTN_TRACE");
    if (aptrace) a.dbg = 256;
    if (pp->pool_out) {
      // pooled output (pcr_attn_apply_pool_ok said yes): whole clouds per wave
      static bool okp = allow_big_lds(attn_apply_stream64_kernel<false, 4, 0, 2, 2, true>);
      (void)okp;
      const long nwgp = ((long)p.B + kApsWaves - 1) / kApsWaves;
      const dim3 ggp((unsigned)(nwgp < ncu ? nwgp : ncu));
      hipLaunchKernelGGL((attn_apply_stream64_kernel<false, 4, 0, 2, 2, true>), ggp, bb, lds_s, st, a);
      PCR_CHECK_LAUNCH();
      return PCR_OK;
    }
    if (nob == 4) {
      static bool ok4 = allow_big_lds(attn_apply_stream64_kernel<true, 4, 0, 4>) && allow_big_lds(attn_apply_stream64_kernel<false, 4, 0, 4>);
      (void)ok4;
      if (p.q_pos) hipLaunchKernelGGL((attn_apply_stream64_kernel<true, 4, 0, 4>), gg, bb, lds_s, st, a);
      else hipLaunchKernelGGL((attn_apply_stream64_kernel<false, 4, 0, 4>), gg, bb, lds_s, st, a);
    } else if (p.c1 < 16) {
      static bool ok1 = allow_big_lds(attn_apply_stream64_kernel<false, 1, 2, 1>);
      (void)ok1;
      hipLaunchKernelGGL((attn_apply_stream64_kernel<false, 1, 2, 1>), gg, bb, lds_s, st, a);
    } else if (p.c1 == 32) { if (cf) PCR_APS(false, 2, 4); else PCR_APS(false, 2, 0); }
    else if (p.q_pos) { if (cf) PCR_APS(true, 4, 4); else PCR_APS(true, 4, 0); }
    else { if (cf) PCR_APS(false, 4, 4); else PCR_APS(false, 4, 0); }
#undef PCR_APS
    if (aptrace) attn_dump_trace(aptrace, "apply64", (int)gg.x, (int)p.B, p.Lq);
    PCR_CHECK_LAUNCH();
    return PCR_OK;
  }
#endif
  const bool wide = 2 * p.d > 128 || p.cout > 128 || p.cfinal > 128;   // some layer has > 4 cout blocks
#define PCR_APPLY(TBv, NRv)                                                          \
  do {                                                                               \
    static bool ok = allow_big_lds(attn_apply_kernel<TBv, NRv>);                     \
    (void)ok;                                                                        \
    hipLaunchKernelGGL((attn_apply_kernel<TBv, NRv>), g, blk, lds, st, a);           \
  } while (0)
  if (tb == 4) {
    if (wide) PCR_APPLY(4, 2);
    else PCR_APPLY(4, 1);
  } else if (tb == 2) {
    if (wide) PCR_APPLY(2, 2);
    else PCR_APPLY(2, 1);
  } else {
    // cout blocks of the widest layer: up to 8 -> two rounds per wave, up to 16 -> four, up to 32 (d_model 512) -> eight
    int widest = 2 * p.d > p.cout ? 2 * p.d : p.cout;
    if (p.cfinal > widest) widest = p.cfinal;
#if PCR_ATTN_PREC == 0
    if (widest > 512) PCR_APPLY(1, 8);
    else if (widest > 256) PCR_APPLY(1, 4);
    else PCR_APPLY(1, 2);
#else
    if (widest > 256) return PCR_ERR_INVALID;
    PCR_APPLY(1, 2);
#endif
  }
#undef PCR_APPLY
  PCR_CHECK_LAUNCH();
  return PCR_OK;
}

#if PCR_ATTN_PREC == 1
// (the caller -- attn_kernels.hip -- has validated the parameters and swapped the bf16 images into wq / wkv / ...)
int pcr_attn_kv_bf3(const pcr_attn_params *pp, pcr_stream_t stream) { return attn_kv_narrow(pp, stream); }
int pcr_attn_apply_bf3(const pcr_attn_params *pp, pcr_stream_t stream) { return attn_apply_launch(pp, stream); }
#else
int pcr_attn_kv_bf3(const pcr_attn_params *pp, pcr_stream_t stream);      // attn_kernels_bf3.hip
int pcr_attn_apply_bf3(const pcr_attn_params *pp, pcr_stream_t stream);

// precision != f32, d_model <= 128 and bf16 images given: the same parameters with the images swapped in
static bool attn_bf(const pcr_attn_params &p, pcr_attn_params &q) {
  const int widest = 2 * p.d > p.cout ? (2 * p.d > p.cfinal ? 2 * p.d : p.cfinal) : (p.cout > p.cfinal ? p.cout : p.cfinal);
  if (p.precision == 0 || p.d > 128 || widest > 256 || !p.wq_bf || !p.wmlp0_bf || !p.wmlp2_bf ||
      (p.cfinal && !p.wfinal_bf))
    return false;
  q = p;      // (wkv stays the f32 image: the kv kernel's projection is f32 in both units, only its output M changes form)
  q.wq = p.wq_bf; q.wmlp0 = p.wmlp0_bf; q.wmlp2 = p.wmlp2_bf; q.wfinal = p.wfinal_bf;
  return true;
}

// Suggested token split of a kv launch: a function of the launch SHAPE only (never of the batch size), so that a pair's
// logits do not depend on how many other pairs share its launch (tests/test_gpu_fullsize.py holds the bench batch to the
// logits of the same pairs run in small groups, bit for bit).  Measured on pt1024 (1024 clouds): 1 / 2 / 3 / 4 splits
// run the d = 64, Sk = 1024 launches in 0.495 / 0.484 / 0.476 / 0.492 ms -- a grid that fills the chip is bound by its
// per-tile work, so the split costs nothing there -- while a few clouds (gallery queries, small batches) gain the
// parallelism they lack.  d = 128 stays whole: its partial matrices are as large as its inputs (0.73 -> 0.85 ms at two).
PCR_EXPORT int pcr_attn_kv_splits(int B, int Sk, int d) {
  (void)B;
  if (Sk < 1 || d > 64) return 1;
  if ((d == 64 || d == 32) && (Sk & 31) == 0) return 1;   // the wave-autonomous kernels' shapes (attn_kv_stream64 / 32): whole clouds
  const int ntile = (Sk + 63) / 64;
  return ntile >= 16 ? 4 : (ntile >= 8 ? 2 : 1);
}

PCR_EXPORT int pcr_attn_kv_f32(const pcr_attn_params *pp, pcr_stream_t stream) {
  if (!pp || attn_check(*pp)) return PCR_ERR_INVALID;
  if (pp->B == 0) return PCR_OK;
  const int d = pp->d;
  if (d > 128) {   // wide: d / 64 workgroups per cloud (attn_kv_wide_kernel)
    AttnArgs a;
    a.p = *pp;
    a.dbg = 0;
    if (!pp->wkv_wide || !pp->bkv_wide || !pp->wmerge_packed || pp->B > 65535) return PCR_ERR_INVALID;
    pcr_note_arith(PCR_PREC_F32);
    const int dh = d / pp->nhead;
    size_t lds = ((size_t)(pp->c2 + d + 3) * 33 + 4 * d + kThreads) * sizeof(float);
    const size_t lds2 = (size_t)dh * 65 * sizeof(float);
    if (lds2 > lds) lds = lds2;
    if (lds > (size_t)kMaxDynLds) return PCR_ERR_INVALID;
    static bool okw = allow_big_lds(attn_kv_wide_kernel<2>) && allow_big_lds(attn_kv_wide_kernel<3>);
    (void)okw;
    dim3 g(d / 64, pp->B), blk(kThreads);
    if (64 + dh <= 256) hipLaunchKernelGGL((attn_kv_wide_kernel<2>), g, blk, lds, pcr_s(stream), a);
    else hipLaunchKernelGGL((attn_kv_wide_kernel<3>), g, blk, lds, pcr_s(stream), a);
    PCR_CHECK_LAUNCH();
    return PCR_OK;
  }
  pcr_attn_params q;
  if (attn_bf(*pp, q)) return pcr_attn_kv_bf3(&q, stream);
  return attn_kv_narrow(pp, stream);
}

static bool attn_pool_ok(const pcr_attn_params &p) {
  pcr_attn_params q;
  return p.d == 64 && p.c1 == 64 && p.cout == 64 && !p.cfinal && !p.q_pos && (p.Lq & 31) == 0 &&
         (p.nhead == 1 || p.nhead == 2 || p.nhead == 4) && attn_bf(p, q);
}

PCR_EXPORT int pcr_attn_apply_pool_ok(const pcr_attn_params *pp) {
  return (pp && !attn_check(*pp) && attn_pool_ok(*pp)) ? 1 : 0;
}

PCR_EXPORT int pcr_attn_apply_f32(const pcr_attn_params *pp, pcr_stream_t stream) {
  if (!pp || attn_check(*pp) || (!pp->out && !pp->pool_out)) return PCR_ERR_INVALID;
  if (pp->pool_out && !attn_pool_ok(*pp)) return PCR_ERR_INVALID;
  if (pp->B == 0) return PCR_OK;
  if (pp->B > 65535) return PCR_ERR_INVALID;
  pcr_attn_params q;
  if (attn_bf(*pp, q)) return pcr_attn_apply_bf3(&q, stream);
  return attn_apply_launch(pp, stream);
}
#endif
