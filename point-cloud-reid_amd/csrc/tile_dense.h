// Shared device helpers of the fused model kernels (included by sa_kernels.hip, attn_kernels.hip,
// model_kernels.hip): the MFMA dense tile, LayerNorm over an LDS tile, small utilities.
#pragma once
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
#ifndef PCR_PF
#define PCR_PF 2
#endif
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
struct DenseNoHook {
  __device__ __forceinline__ void operator()() const {}
};

// The first PF weight fragments of a dense call (what tile_dense_impl would load in its prologue), for
// callers that request them EARLY -- e.g. right after the k-loop of the previous layer, so that the L2
// round trip is over before the call starts -- and pass them in as `ring`.
template <int NR, int WAYS, int PF = PCR_PF>
__device__ __forceinline__ void tile_dense_ring_load(const float *__restrict__ wp, int CP, int OP,
                                                     f32x4 (&ring)[PF][NR]) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6) & 3;
  const int l31 = lane & 31, h = lane >> 5;
  const int nCB = OP >> 5, KB = CP >> 3;
  const int cb0 = WAYS == 1 ? wave : (WAYS == 2 ? (wave & 1) : 0);
  const size_t wstride = (size_t)OP * 2;
#pragma unroll
  for (int nr = 0; nr < NR; nr++) {
    int cb = cb0 + 4 * nr;
    cb = cb < nCB ? cb : nCB - 1;
    const f32x4 *wrow = reinterpret_cast<const f32x4 *>(wp) + (size_t)cb * 64 + (size_t)l31 * 2 + h;
#pragma unroll
    for (int i = 0; i < PF; i++) ring[i][nr] = wrow[(size_t)(i < KB ? i : KB - 1) * wstride];
  }
}

// ring: nullptr, or the fragments tile_dense_ring_load fetched for THIS call.  after_k(): called once between
// the k-loop and the epilogue (before the sync_epi barrier): the place to request the next call's ring.
// RES (resident weights): `ring` holds ALL k-blocks of the call (KB <= PF, narrow layers): no weight load inside
// the call at all -- a caller that runs the same layer on many tiles fetches the ring once.
// CIN / COUT (split contraction): the accumulators come from / go back to `carry` instead of the seeds / the
// epilogue, so a caller can walk a long cin extent chunk by chunk through a small LDS buffer.
template <int TB, int NR, int WAYS, bool TILE, class Epi, int PF = PCR_PF, class AfterK = DenseNoHook,
          bool RES = false, bool CIN = false, bool COUT = false>
// PF must be even (the B-operand double buffer alternates per k-block)
__device__ __forceinline__ void tile_dense_impl(const float *__restrict__ in, int CP,
                                                const float *__restrict__ wp, int OP, bool sync_epi,
                                                Epi epi, const float *__restrict__ init = nullptr,
                                                f32x4 (*ring)[NR] = nullptr, AfterK after_k = AfterK(),
                                                int opfull = 0,
                                                f32x16 (*carry)[(TB + WAYS - 1) / WAYS] = nullptr) {
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
#ifdef PCR_DIAG_WSTRIDE0
  const size_t wstride = 0;   // diagnostic build only: every k-block re-reads the first one (L1 hits)
#else
  // opfull: `wp` points at a WINDOW of OP cout rows inside a wider packed image of opfull padded couts
  const size_t wstride = (size_t)(opfull ? opfull : OP) * 2;
#endif
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
  // first PF weight fragments, requested BEFORE the accumulator seeds so that all round trips overlap
  f32x4 aw[PF][NR];
#pragma unroll
  for (int i = 0; i < PF; i++) {
    const int ki = i < KB ? i : KB - 1;
#pragma unroll
    for (int nr = 0; nr < NR; nr++) aw[i][nr] = ring ? ring[i][nr] : wrow[nr][(size_t)ki * wstride];
  }
  // `init` (OP floats, zero-padded) seeds the accumulators with the per-cout bias / folded BatchNorm
  // shift, so the epilogue needs no per-element constant loads.  The 16 accumulator rows of a lane are
  // four runs of four consecutive couts (8g + 4h .. +3), i.e. four 16-byte loads, issued together
  // with the first weight fragments.
  f32x16 acc[NR][TBW];
#pragma unroll
  for (int nr = 0; nr < NR; nr++) {
    int cbi = cb0 + 4 * nr;
    cbi = cbi < nCB ? cbi : nCB - 1;
    if constexpr (CIN) {
#pragma unroll
      for (int j = 0; j < TBW; j++) acc[nr][j] = carry[nr][j];
    } else if (init != nullptr) {
      const f32x4 *ip = reinterpret_cast<const f32x4 *>(init + cbi * 32 + 4 * h);
#pragma unroll
      for (int g = 0; g < 4; g++) {
        const f32x4 v4 = ip[2 * g];
#pragma unroll
        for (int q = 0; q < 4; q++)
#pragma unroll
          for (int j = 0; j < TBW; j++) acc[nr][j][4 * g + q] = v4[q];
      }
    } else {
#pragma unroll
      for (int r = 0; r < 16; r++)
#pragma unroll
        for (int j = 0; j < TBW; j++) acc[nr][j][r] = 0.f;
    }
  }
  // PF weight-fragment register sets in a ring: the 16-byte load for k-block kb+PF is issued right
  // after the last use of set (kb % PF) and has PF-1 blocks of MFMAs to land; the B operands (LDS) of
  // block kb+1 are requested before the MFMAs of block kb.  The scheduling barriers pin that order
  // (left alone, hipcc sinks every weight load to the end of the unrolled body, where its latency is
  // fully exposed, and issues each ds_read right before its first use).
  float xb[2][TBW][4];
  auto load_x = [&](float (&x)[TBW][4], int kb) {
#pragma unroll
    for (int j = 0; j < TBW; j++) {
      const float *bt = brow[j] + kb * 8 * RP;
      x[j][0] = bt[0];
      x[j][1] = bt[2 * RP];
      x[j][2] = bt[4 * RP];
      x[j][3] = bt[6 * RP];
    }
  };
  auto mma = [&](const f32x4 (&a)[NR], const float (&x)[TBW][4]) {
#pragma unroll
    for (int q = 0; q < 4; q++)
#pragma unroll
      for (int j = 0; j < TBW; j++)
#pragma unroll
        for (int nr = 0; nr < NR; nr++)
          acc[nr][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[nr][q], x[j][q], acc[nr][j], 0, 0, 0);
  };
  load_x(xb[0], 0);
  int kb = 0;
  if constexpr (RES) {
#pragma unroll
    for (int i = 0; i < PF; i++)
      if (i < KB) {
        load_x(xb[(i + 1) & 1], i + 1 < KB ? i + 1 : KB - 1);
        __builtin_amdgcn_sched_barrier(0);
        mma(aw[i], xb[i & 1]);
        __builtin_amdgcn_sched_barrier(0);
      }
    kb = KB;
  }
  // main groups: every prefetch address is inside the image (no clamps: plain pointer increments)
  for (; kb + 2 * PF <= KB; kb += PF) {
#pragma unroll
    for (int i = 0; i < PF; i++) {
      load_x(xb[(i + 1) & 1], kb + i + 1);
      __builtin_amdgcn_sched_barrier(0);
      mma(aw[i], xb[i & 1]);
#pragma unroll
      for (int nr = 0; nr < NR; nr++) aw[i][nr] = wrow[nr][(size_t)(kb + i + PF) * wstride];
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  // last whole group: the ring is drained, partial prefetch only for the tail blocks
  if (kb + PF <= KB) {
#pragma unroll
    for (int i = 0; i < PF; i++) {
      const int kx = kb + i + 1 < KB ? kb + i + 1 : KB - 1;
      load_x(xb[(i + 1) & 1], kx);
      __builtin_amdgcn_sched_barrier(0);
      mma(aw[i], xb[i & 1]);
      if (i + 1 < PF) {   // only PF - 1 tail blocks can exist
        const int kn = kb + i + PF < KB ? kb + i + PF : KB - 1;
#pragma unroll
        for (int nr = 0; nr < NR; nr++) aw[i][nr] = wrow[nr][(size_t)kn * wstride];
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    kb += PF;
  }
  {
    const int rem = KB - kb;   // < PF; aw[i] already holds k-block kb + i, xb[0] the operands of block kb
#pragma unroll
    for (int i = 0; i + 1 < PF; i++)
      if (i < rem) {
        const int kx = kb + i + 1 < KB ? kb + i + 1 : KB - 1;
        load_x(xb[(i + 1) & 1], kx);
        mma(aw[i], xb[i & 1]);
      }
  }
  if constexpr (COUT) {
#pragma unroll
    for (int nr = 0; nr < NR; nr++)
#pragma unroll
      for (int j = 0; j < TBW; j++) carry[nr][j] = acc[nr][j];
    return;
  }
  after_k();
  if (sync_epi) __syncthreads();
#pragma unroll
  for (int nr = 0; nr < NR; nr++) {
    const int cb = cb0 + 4 * nr;
    if (cb < nCB) {
#pragma unroll
      for (int j = 0; j < TBW; j++) {
        const int tb = tb0 + j * WAYS;
        if (tb < TB) {
          if constexpr (TILE) {
            epi(acc[nr][j], cb, tb, l31, h);   // whole 32x32 tile: lane = token l31 of block tb, 16 couts in regs
          } else {
            const int t = tb * 32 + l31;
#pragma unroll
            for (int r = 0; r < 16; r++) epi(acc[nr][j][r], cb * 32 + (r & 3) + 8 * (r >> 2) + 4 * h, t);
          }
        }
      }
    }
  }
}

// WSEL = 0: pick the wave/tile split from OP at run time (all three bodies are compiled in and the
// register allocation is their maximum); WSEL = 1 / 2 / 4: the caller guarantees OP/32 >= 3 / == 2 /
// == 1 and only that body is compiled (fewer registers => more waves per SIMD).
// rounds / ways of the explicit variants (what a caller's ring must be dimensioned with)
template <int NR, int WSEL>
struct DenseShape {
  static constexpr int ways = WSEL == 0 ? 1 : WSEL;
  static constexpr int nr = WSEL == 1 ? NR : 1;
};

template <int TB, int NR, int WSEL = 0, bool TILE = false, class Epi, class AfterK = DenseNoHook, int PFv = PCR_PF,
          bool RES = false>
__device__ __forceinline__ void tile_dense2(const float *__restrict__ in, int CP,
                                            const float *__restrict__ wp, int OP, bool sync_epi, Epi epi,
                                            const float *__restrict__ init = nullptr,
                                            f32x4 (*ring)[DenseShape<NR, WSEL>::nr] = nullptr,
                                            AfterK after_k = AfterK(), int opfull = 0) {
  if constexpr (WSEL == 1) {
    tile_dense_impl<TB, NR, 1, TILE, Epi, PFv, AfterK, RES>(in, CP, wp, OP, sync_epi, epi, init, ring, after_k, opfull);
  } else if constexpr (WSEL == 2) {
    tile_dense_impl<TB, 1, 2, TILE, Epi, PFv, AfterK, RES>(in, CP, wp, OP, sync_epi, epi, init, ring, after_k, opfull);
  } else if constexpr (WSEL == 4) {
    tile_dense_impl<TB, 1, 4, TILE, Epi, PFv, AfterK, RES>(in, CP, wp, OP, sync_epi, epi, init, ring, after_k, opfull);
  } else {
    after_k();   // (generic shape: no early ring, the hook still runs once)
    const int nCB = OP >> 5;
    const DenseNoHook nh;
    // rounds per wave for the layer at hand: NR of the caller, except that callers instantiated for the 512- / 1024-row
    // layers of the wide attention blocks (NR = 4 / 8) run their narrower layers with fewer rounds
    if constexpr (NR >= 8) {
      if (nCB > 16) {
        tile_dense_impl<TB, 8, 1, TILE>(in, CP, wp, OP, sync_epi, epi, init, nullptr, nh, opfull);
        return;
      }
    }
    if constexpr (NR >= 4) {
      if (nCB > 8) {
        tile_dense_impl<TB, 4, 1, TILE>(in, CP, wp, OP, sync_epi, epi, init, nullptr, nh, opfull);
        return;
      }
    }
    if (nCB > 4) tile_dense_impl<TB, (NR >= 4 ? 2 : NR), 1, TILE>(in, CP, wp, OP, sync_epi, epi, init, nullptr, nh, opfull);
    else if (nCB >= 3) tile_dense_impl<TB, 1, 1, TILE>(in, CP, wp, OP, sync_epi, epi, init, nullptr, nh, opfull);   // one round
    else if (nCB == 2) tile_dense_impl<TB, 1, 2, TILE>(in, CP, wp, OP, sync_epi, epi, init, nullptr, nh, opfull);
    else tile_dense_impl<TB, 1, 4, TILE>(in, CP, wp, OP, sync_epi, epi, init, nullptr, nh, opfull);
  }
}

// ---------------------------------------------------------------- bf16 matrix core (gfx950) ----
// The same dense tile on v_mfma_f32_32x32x16_bf16 (16x the f32-input MFMA's rate, f32 accumulate).  The LDS tile keeps
// its f32 [C][RP] layout -- every other phase of the callers (gathers, LayerNorm, pooling, stores) is untouched -- and the
// B operand is converted where it is consumed: a lane reads the 8 channels k = 16 s + 8 h .. + 7 of its token and forms
//   NS = 3 ("bf16x3", split bf16):  x = hi + lo + O(2^-18 |x|),  hi = bf16(x), lo = bf16(x - hi)  (round to nearest),
//            W x ~= W_hi x_hi + W_hi x_lo + W_lo x_hi : three MFMAs per product, relative error ~2^-17 per term
//            (measured end to end: logits within 2e-6 of the f32 path) at ~5x the f32-input rate;
//   NS = 1 ("bf16"): x_hi and W_hi only -- BASELINE config 2's "bf16 activations, f32 accumulate" taken literally.
// Weights: image [S = ceil16(cin)/16][OP/32][hi, lo][64 lanes][8 bf16] (pcr_pack_weight_bf16x2): one 16-byte load per
// lane, step and part, streamed through a ring of bf_pf(NR) steps.
// K ORDER inside a 16-channel step: element j of lane half h is channel 16 s + bf_kpos(h, j) = 16 s + 4 h + j (j < 4) |
// 16 s + 8 + 4 h + (j - 4) -- not 8 h + j.  That is the order in which a 32 x 32 accumulator tile hands its rows to a
// lane (four runs of four couts, 8 g + 4 h + q), so a producing layer can store its output already converted, as
// 16-byte bf16 pieces in accumulator order (bf_store_tile -> "bf image", below), and the consuming layer's B operand
// is one ds_read_b128 per part instead of eight ds_read_b32 and 24 conversion instructions per step, repeated by every
// wave that owns a cout block.  The f32-tile form (BIMG = false) reads its rows in the same order, so ONE weight image
// serves both.
//
// bf image of a [C][ROWS] activation tile (C a multiple of 32): 16-byte units, unit (2 P + part) ROWS + t holds piece P
// (8 channels: 32 (P >> 2) + 16 ((P >> 1) & 1) + bf_kpos(P & 1, j)) of token t, part 0 = hi, 1 = lo.  Step s of the
// consumer reads pieces 2 s (lanes 0..31) and 2 s + 1 (lanes 32..63): 512 contiguous bytes per half, conflict-free; the
// producer's lane (token l31, half h) writes pieces 4 cb + 2 gp + h, gp = 0, 1: 128 contiguous bytes per 8 lanes.
// The image takes C ROWS 4 bytes (hi + lo), what the f32 tile takes.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
// ring depth in 16-channel steps: a step is 3 TBW NR MFMAs of 32 cycles, so two cout-block rounds per wave halve the
// depth an L2 round trip needs (and the 8 registers per step and round are what decides the residency)
constexpr int bf_pf(int nr) { return nr >= 2 ? 2 : 4; }

__host__ __device__ constexpr int bf_kpos(int h, int j) { return j < 4 ? 4 * h + j : 8 + 4 * h + (j - 4); }

// x - hi for a converted pair (exact: hi is x rounded to 8 significant bits).  Per eight values the split is 4 v_cvt_pk_bf16_f32,
// 8 v_and / v_lshlrev (hi back to f32), 4 v_pk_add_f32, 4 v_cvt_pk_bf16_f32.  (Round 6, measured and dropped: the residual as
// v_dot2c_f32_bf16 (hi . {-1, 0} + x; 16 instructions instead of 20) -- not full rate on gfx950 (tools/probe_split.hip: 3 % on a
// split-only loop, profiles/r06k_split_probe.txt), and hipcc 7.2 folds the two packed constants {-1, 0} and {0, -1} into the same
// operand (the inline constant -1.0 is the f32 pattern 0xbf800000 = {0, -1}), so the even element comes out wrong.)
__device__ __forceinline__ f32x2 bf_residual2(const f32x2 v, const bf16x2 h2) {
  return v - __builtin_convertvector(h2, f32x2);
}

__device__ __forceinline__ void bf_split8(const float (&x)[8], bf16x8 &hi, bf16x8 &lo, bool want_lo) {
#pragma unroll
  for (int q = 0; q < 4; q++) {
    const f32x2 v = {x[2 * q], x[2 * q + 1]};
    const bf16x2 h2 = __builtin_convertvector(v, bf16x2);
    hi[2 * q] = h2[0];
    hi[2 * q + 1] = h2[1];
    if (want_lo) {
      const f32x2 r = bf_residual2(v, h2);
      const bf16x2 l2 = __builtin_convertvector(r, bf16x2);
      lo[2 * q] = l2[0];
      lo[2 * q + 1] = l2[1];
    }
  }
}

// relu(acc) of one 32 x 32 accumulator tile (cout block cb, token block tb; lane = token l31, half h) -> bf image
template <bool LO>
__device__ __forceinline__ void bf_store_tile(float *img, int ROWS, const f32x16 &acc, int cb, int tb, int l31, int h) {
  bf16x8 *u = reinterpret_cast<bf16x8 *>(img);
#pragma unroll
  for (int gp = 0; gp < 2; gp++) {
    float x[8];
#pragma unroll
    for (int q = 0; q < 8; q++) {
      const int b = __float_as_int(acc[8 * gp + q]);
      x[q] = __int_as_float(b > 0 ? b : 0);
    }
    bf16x8 hi, lo;
    bf_split8(x, hi, lo, LO);
    const int P = 4 * cb + 2 * gp + h;
    u[(2 * P) * ROWS + tb * 32 + l31] = hi;
    if constexpr (LO) u[(2 * P + 1) * ROWS + tb * 32 + l31] = lo;
  }
}

// The first PF steps of a bf16 dense call's weight stream (what tile_dense_bf_impl loads in its prologue), for callers
// that request them EARLY -- at the top of the kernel, or right after the previous call's k-loop -- so that the L2 round
// trip is over before the call starts.  RES callers hold ALL steps of a narrow layer this way (KS <= PF).
template <int PF, int NR>
struct BfRing {
  bf16x8 h[PF][NR], l[PF][NR];
};

template <int NR, int WAYS, int PF, int NS>
__device__ __forceinline__ void bf_ring_load(const float *__restrict__ wp_, int CP, int OP, BfRing<PF, NR> &ring,
                                             int opfull = 0) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6) & 3;
  const int nCB = OP >> 5, KS = (CP + 15) >> 4;
  const int cb0 = WAYS == 1 ? wave : (WAYS == 2 ? (wave & 1) : 0);
  const size_t wstride = (size_t)((opfull ? opfull : OP) >> 5) * 128;
#pragma unroll
  for (int nr = 0; nr < NR; nr++) {
    int cb = cb0 + 4 * nr;
    cb = cb < nCB ? cb : nCB - 1;
    const bf16x8 *wrow = reinterpret_cast<const bf16x8 *>(wp_) + (size_t)cb * 128 + lane;
#pragma unroll
    for (int i = 0; i < PF; i++) {
      const int ki = i < KS ? i : KS - 1;
      ring.h[i][nr] = wrow[(size_t)ki * wstride];
      if constexpr (NS == 3) ring.l[i][nr] = wrow[(size_t)ki * wstride + 64];
    }
  }
}

// BIMG: `in` is a bf image ([CP / 8 pieces][2][32 TB tokens] 16-byte units), not an f32 [CP][RP] tile
// ring: null, or the steps bf_ring_load fetched for THIS call; RES: the ring holds every step (no weight load inside)
// CIN / COUT (split contraction, as tile_dense_impl): the accumulators come from / go back to `carry`, so a caller walks a
// long cin extent chunk by chunk through a small LDS image (dense_bf_kernel, model_kernels.hip)
// BSWZ (bf images only): token t of the image sits at unit position bf_tok_swz(t) = t ^ ((t >> 4) & 3) -- a permutation inside
// aligned groups of four, so a consumer's 32 lanes still read 32 consecutive units, while a PRODUCER whose lane holds four
// consecutive tokens (a 16-byte global load per channel) writes, per store instruction, units that fall into 16 different
// 16-byte bank groups instead of 4 (an 8-way -> 2-way conflict on every ds_write_b128 of the fill)
__host__ __device__ constexpr int bf_tok_swz(int t) { return t ^ ((t >> 4) & 3); }

template <int TB, int NR, int WAYS, bool TILE, int NS, class Epi, int PF = bf_pf(NR), class AfterK = DenseNoHook,
          bool BIMG = false, bool RES = false, bool CIN = false, bool COUT = false, bool BSWZ = false>
__device__ __forceinline__ void tile_dense_bf_impl(const float *__restrict__ in, int CP,
                                                   const float *__restrict__ wp_, int OP, bool sync_epi, Epi epi,
                                                   const float *__restrict__ init = nullptr,
                                                   AfterK after_k = AfterK(), int opfull = 0,
                                                   const BfRing<PF, NR> *ring = nullptr,
                                                   f32x16 (*carry)[(TB + WAYS - 1) / WAYS] = nullptr) {
  static_assert((PF & 1) == 0, "the B-operand double buffer alternates per step");
  constexpr int RP = 32 * TB + 1;
  constexpr int TBW = (TB + WAYS - 1) / WAYS;
  constexpr bool kLo = NS == 3;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int l31 = lane & 31, h = lane >> 5;
  const int nCB = OP >> 5, KS = (CP + 15) >> 4;
  const bool tail = (CP & 8) != 0;          // the last step's upper half (k = 8 .. 15) lies beyond the tile's rows
  const int cb0 = WAYS == 1 ? wave : (WAYS == 2 ? (wave & 1) : 0);
  const int tb0 = WAYS == 1 ? 0 : (WAYS == 2 ? (wave >> 1) : wave);
  const size_t wstride = (size_t)((opfull ? opfull : OP) >> 5) * 128;   // bf16x8 units per step
  const bf16x8 *wrow[NR];
#pragma unroll
  for (int nr = 0; nr < NR; nr++) {
    int cb = cb0 + 4 * nr;
    cb = cb < nCB ? cb : nCB - 1;
    wrow[nr] = reinterpret_cast<const bf16x8 *>(wp_) + (size_t)cb * 128 + lane;
  }
  const float *brow[TBW];
  const bf16x8 *bimg[TBW];
#pragma unroll
  for (int j = 0; j < TBW; j++) {
    int tb = tb0 + j * WAYS;
    tb = tb < TB ? tb : TB - 1;
    brow[j] = in + 4 * h * RP + tb * 32 + l31;                                            // rows bf_kpos(h, .) of a step
    bimg[j] = reinterpret_cast<const bf16x8 *>(in) + 2 * h * (32 * TB) +
              (BSWZ ? bf_tok_swz(tb * 32 + l31) : tb * 32 + l31);                         // piece 2 s + h of a step
  }
  bf16x8 ah[PF][NR], al[PF][NR];
#pragma unroll
  for (int i = 0; i < PF; i++) {
    const int ki = i < KS ? i : KS - 1;
#pragma unroll
    for (int nr = 0; nr < NR; nr++) {
      if (ring) {
        ah[i][nr] = ring->h[i][nr];
        if constexpr (kLo) al[i][nr] = ring->l[i][nr];
      } else {
        ah[i][nr] = wrow[nr][(size_t)ki * wstride];
        if constexpr (kLo) al[i][nr] = wrow[nr][(size_t)ki * wstride + 64];
      }
    }
  }
  f32x16 acc[NR][TBW];
#pragma unroll
  for (int nr = 0; nr < NR; nr++) {
    int cbi = cb0 + 4 * nr;
    cbi = cbi < nCB ? cbi : nCB - 1;
    if constexpr (CIN) {
#pragma unroll
      for (int j = 0; j < TBW; j++) acc[nr][j] = carry[nr][j];
    } else if (init != nullptr) {
      const f32x4 *ip = reinterpret_cast<const f32x4 *>(init + cbi * 32 + 4 * h);
#pragma unroll
      for (int g = 0; g < 4; g++) {
        const f32x4 v4 = ip[2 * g];
#pragma unroll
        for (int q = 0; q < 4; q++)
#pragma unroll
          for (int j = 0; j < TBW; j++) acc[nr][j][4 * g + q] = v4[q];
      }
    } else {
#pragma unroll
      for (int r = 0; r < 16; r++)
#pragma unroll
        for (int j = 0; j < TBW; j++) acc[nr][j][r] = 0.f;
    }
  }
  // operands of one step: raw f32 rows (converted in mma) or, from a bf image, the hi / lo pieces themselves
  struct XB {
    float r[BIMG ? 1 : 8];
    bf16x8 hi, lo;
  };
  XB xr[2][TBW];
  auto load_x = [&](XB (&x)[TBW], int s) {
#pragma unroll
    for (int j = 0; j < TBW; j++) {
      if constexpr (BIMG) {
        const bf16x8 *bt = bimg[j] + (size_t)s * (4 * 32 * TB);
        x[j].hi = bt[0];
        if constexpr (kLo) x[j].lo = bt[32 * TB];
      } else {
        const float *bt = brow[j] + s * 16 * RP;
#pragma unroll
        for (int q = 0; q < 8; q++) x[j].r[q] = bt[(q < 4 ? q : q + 4) * RP];
      }
    }
  };
  auto mma = [&](int i, XB (&x)[TBW], bool last) {
    if constexpr (!BIMG) {
      if (last && tail) {   // channels 16 s + 8 .. + 15 lie beyond the tile: elements 4 .. 7 of both lane halves
#pragma unroll
        for (int j = 0; j < TBW; j++)
#pragma unroll
          for (int q = 4; q < 8; q++) x[j].r[q] = 0.f;
      }
    }
#pragma unroll
    for (int j = 0; j < TBW; j++) {
      bf16x8 bh, bl;
      if constexpr (BIMG) {
        bh = x[j].hi;
        bl = x[j].lo;
      } else {
        bf_split8(x[j].r, bh, bl, kLo);
      }
#pragma unroll
      for (int nr = 0; nr < NR; nr++) {
        if constexpr (kLo) {
          acc[nr][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i][nr], bl, acc[nr][j], 0, 0, 0);
          acc[nr][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i][nr], bh, acc[nr][j], 0, 0, 0);
        }
        acc[nr][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i][nr], bh, acc[nr][j], 0, 0, 0);
      }
    }
  };
  auto refill = [&](int i, int ks) {
    if constexpr (RES) return;
#pragma unroll
    for (int nr = 0; nr < NR; nr++) {
      ah[i][nr] = wrow[nr][(size_t)ks * wstride];
      if constexpr (kLo) al[i][nr] = wrow[nr][(size_t)ks * wstride + 64];
    }
  };
  load_x(xr[0], 0);
  int s = 0;
  for (; s + 2 * PF <= KS; s += PF) {
#pragma unroll
    for (int i = 0; i < PF; i++) {
      load_x(xr[(i + 1) & 1], s + i + 1);
      __builtin_amdgcn_sched_barrier(0);
      mma(i, xr[i & 1], false);
      refill(i, s + i + PF);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  if (s + PF <= KS) {
#pragma unroll
    for (int i = 0; i < PF; i++) {
      const int sx = s + i + 1 < KS ? s + i + 1 : KS - 1;
      load_x(xr[(i + 1) & 1], sx);
      __builtin_amdgcn_sched_barrier(0);
      mma(i, xr[i & 1], s + i == KS - 1);
      if (i + 1 < PF) {
        const int sn = s + i + PF < KS ? s + i + PF : KS - 1;
        refill(i, sn);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    s += PF;
  }
  {
    const int rem = KS - s;   // < PF; ring slot i holds step s + i, xr[0] the operands of step s
#pragma unroll
    for (int i = 0; i + 1 < PF; i++)
      if (i < rem) {
        const int sx = s + i + 1 < KS ? s + i + 1 : KS - 1;
        load_x(xr[(i + 1) & 1], sx);
        mma(i, xr[i & 1], s + i == KS - 1);
      }
  }
  if constexpr (COUT) {
#pragma unroll
    for (int nr = 0; nr < NR; nr++)
#pragma unroll
      for (int j = 0; j < TBW; j++) carry[nr][j] = acc[nr][j];
    return;
  }
  after_k();
  if (sync_epi) __syncthreads();
#pragma unroll
  for (int nr = 0; nr < NR; nr++) {
    const int cb = cb0 + 4 * nr;
    if (cb < nCB) {
#pragma unroll
      for (int j = 0; j < TBW; j++) {
        const int tb = tb0 + j * WAYS;
        if (tb < TB) {
          if constexpr (TILE) {
            epi(acc[nr][j], cb, tb, l31, h);
          } else {
            const int t = tb * 32 + l31;
#pragma unroll
            for (int r = 0; r < 16; r++) epi(acc[nr][j][r], cb * 32 + (r & 3) + 8 * (r >> 2) + 4 * h, t);
          }
        }
      }
    }
  }
}

// PREC 0: f32-input MFMA (exact fmaf chains); 1: split bf16 (three MFMAs per product); 2: plain bf16.  The bf16 forms
// exist for the explicit wave / tile splits only (WSEL 1 / 2 / 4).
template <int NR, int WSEL>
using BfRingOf = BfRing<bf_pf(DenseShape<NR, WSEL>::nr), DenseShape<NR, WSEL>::nr>;

template <int PREC, int TB, int NR, int WSEL = 0, bool TILE = false, bool BIMG = false, class Epi,
          class AfterK = DenseNoHook, int PFv = PCR_PF, bool RES = false>
__device__ __forceinline__ void tile_dense2p(const float *__restrict__ in, int CP, const float *__restrict__ wp, int OP,
                                             bool sync_epi, Epi epi, const float *__restrict__ init = nullptr,
                                             f32x4 (*ring)[DenseShape<NR, WSEL>::nr] = nullptr,
                                             AfterK after_k = AfterK(), int opfull = 0,
                                             const BfRingOf<NR, WSEL> *bfring = nullptr) {
  if constexpr (PREC == 0) {
    static_assert(!BIMG, "bf images feed the bf16 tile only");
    tile_dense2<TB, NR, WSEL, TILE, Epi, AfterK, PFv, RES>(in, CP, wp, OP, sync_epi, epi, init, ring, after_k, opfull);
  } else {
    constexpr int NS = PREC == 1 ? 3 : 1;
    if constexpr (WSEL != 0) {
      tile_dense_bf_impl<TB, DenseShape<NR, WSEL>::nr, DenseShape<NR, WSEL>::ways, TILE, NS, Epi,
                         bf_pf(DenseShape<NR, WSEL>::nr), AfterK, BIMG>(in, CP, wp, OP, sync_epi, epi, init, after_k,
                                                                        opfull, bfring);
    } else {
      // generic shape: the wave / tile split follows OP at run time, as in tile_dense2 (up to two cout-block rounds)
      static_assert(NR <= 2, "bf16 dense tiles cover up to eight cout blocks");
      after_k();
      const int nCB = OP >> 5;
      const DenseNoHook nh;
      if (nCB > 4) tile_dense_bf_impl<TB, NR, 1, TILE, NS, Epi, bf_pf(NR), DenseNoHook, BIMG>(in, CP, wp, OP, sync_epi, epi, init, nh, opfull);
      else if (nCB >= 3) tile_dense_bf_impl<TB, 1, 1, TILE, NS, Epi, bf_pf(1), DenseNoHook, BIMG>(in, CP, wp, OP, sync_epi, epi, init, nh, opfull);
      else if (nCB == 2) tile_dense_bf_impl<TB, 1, 2, TILE, NS, Epi, bf_pf(1), DenseNoHook, BIMG>(in, CP, wp, OP, sync_epi, epi, init, nh, opfull);
      else tile_dense_bf_impl<TB, 1, 4, TILE, NS, Epi, bf_pf(1), DenseNoHook, BIMG>(in, CP, wp, OP, sync_epi, epi, init, nh, opfull);
    }
  }
}

// the early request of a bf16 call's first steps, in terms of the caller's (NR, WSEL)
template <int PREC, int NR, int WSEL>
__device__ __forceinline__ void bf_ring_load2(const float *__restrict__ wp, int CP, int OP, BfRingOf<NR, WSEL> &ring) {
  bf_ring_load<DenseShape<NR, WSEL>::nr, DenseShape<NR, WSEL>::ways, bf_pf(DenseShape<NR, WSEL>::nr), PREC == 1 ? 3 : 1>(
      wp, CP, OP, ring);
}

// elu(x) + 1 = x + 1 (x > 0) | exp(x) (x <= 0); hardware exp2 (v_exp_f32, ~1 ulp) instead of the libm
// expansion: this runs once per projected Q/K element and was a third of the attention kernels' VALU time
__device__ __forceinline__ float elu1(float x) { return x > 0.f ? x + 1.0f : __expf(x); }
// ReLU on the bit pattern (signed integer max; sa_kernels_impl.h relu_bits): the same value for every non-NaN input, one
// INTEGER VALU instruction -- no canonicalisation prefix, and an integer instruction runs under another wave's MFMAs
// where an f32 one does not.  (The same swap for elu1's compare measured nothing: profiles/r06_attn_relu_ab.txt.)
__device__ __forceinline__ float relu_i(float v) {
  const int b = __float_as_int(v);
  return __int_as_float(b > 0 ? b : 0);
}

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

// loads a [C][T] tile of a (B,C,L) tensor into LDS rows [0,CP), zero beyond C or beyond L.
// Every thread issues a BATCH of independent loads from clamped (always valid) addresses and selects
// afterwards: a plain `dst = ok ? src[..] : 0` loop compiles to one load + s_waitcnt vmcnt(0) per
// element, i.e. CP*T/256 serial memory round trips per workgroup.  16-byte loads when the tile is whole
// and aligned (t0 is a multiple of 32 in every caller).
__device__ __forceinline__ void load_tile(float *dst, int RP, const float *src, int C, int CP, int L,
                                          int t0, int T) {
  const bool vec = ((L & 3) == 0) && (t0 + T <= L) && ((reinterpret_cast<size_t>(src) & 15) == 0);
  if (vec) {
    const int Q = T >> 2, totq = CP * Q;
    for (int e0 = threadIdx.x; e0 < totq; e0 += 4 * kThreads) {
      f32x4 v[4];
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const int e = e0 + u * kThreads;
        const int c = e / Q, q = e - c * Q;
        const bool ok = e < totq && c < C;
        const f32x4 x = *reinterpret_cast<const f32x4 *>(src + (size_t)(ok ? c : 0) * L + t0 + 4 * q);
        v[u] = ok ? x : f32x4{0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int u = 0; u < 4; u++) {
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
    for (int e0 = threadIdx.x; e0 < total; e0 += 8 * kThreads) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; u++) {
        const int e = e0 + u * kThreads;
        const int c = e / T, t = e - c * T;
        const bool ok = e < total && c < C && t0 + t < L;
        const float x = src[ok ? (size_t)c * L + t0 + t : 0];
        v[u] = ok ? x : 0.f;
      }
#pragma unroll
      for (int u = 0; u < 8; u++) {
        const int e = e0 + u * kThreads;
        if (e < total) {
          const int c = e / T, t = e - c * T;
          dst[c * RP + t] = v[u];
        }
      }
    }
  }
}

// the same tile from a POINT-major tensor: src (L,C) of one cloud, element (c,t) = src[(t0+t)*C + c]
__device__ __forceinline__ void load_tile_pm(float *dst, int RP, const float *src, int C, int CP, int L,
                                             int t0, int T) {
  if ((C & 3) == 0 && (reinterpret_cast<size_t>(src) & 15) == 0) {
    const int Q = CP >> 2, totq = T * Q;    // CP is a multiple of 8
    // piece e = (token e / Q, channel quad e % Q); e advances by kThreads per piece: ONE division per thread, then
    // increments (a runtime division is ~20 instructions, and there would be one per piece and pass)
    const int dt = kThreads / Q, dq = kThreads - dt * Q;
    int t = threadIdx.x / Q, q = threadIdx.x - t * Q;
    for (int e0 = threadIdx.x; e0 < totq; e0 += 4 * kThreads) {
      f32x4 v[4];
      int tt[4], qq[4];
#pragma unroll
      for (int u = 0; u < 4; u++) {
        tt[u] = t;
        qq[u] = q;
        const bool ok = e0 + u * kThreads < totq && 4 * q < C && t0 + t < L;
        const f32x4 x = *reinterpret_cast<const f32x4 *>(src + (ok ? (size_t)(t0 + t) * C + 4 * q : 0));
        v[u] = ok ? x : f32x4{0.f, 0.f, 0.f, 0.f};
        t += dt;
        q += dq;
        if (q >= Q) { q -= Q; t++; }
      }
#pragma unroll
      for (int u = 0; u < 4; u++) {
        if (e0 + u * kThreads < totq) {
          float *d = dst + 4 * qq[u] * RP + tt[u];
          d[0] = v[u][0];
          d[RP] = v[u][1];
          d[2 * RP] = v[u][2];
          d[3 * RP] = v[u][3];
        }
      }
    }
  } else {
    const int total = T * CP;
    for (int e0 = threadIdx.x; e0 < total; e0 += 8 * kThreads) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; u++) {
        const int e = e0 + u * kThreads;
        const int t = e / CP, c = e - t * CP;
        const bool ok = e < total && c < C && t0 + t < L;
        const float x = src[ok ? (size_t)(t0 + t) * C + c : 0];
        v[u] = ok ? x : 0.f;
      }
#pragma unroll
      for (int u = 0; u < 8; u++) {
        const int e = e0 + u * kThreads;
        if (e < total) {
          const int t = e / CP, c = e - t * CP;
          dst[c * RP + t] = v[u];
        }
      }
    }
  }
}

template <class Kern>
bool allow_big_lds(Kern k) {
  return hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize,
                             kMaxDynLds) == hipSuccess;
}

}  // namespace
