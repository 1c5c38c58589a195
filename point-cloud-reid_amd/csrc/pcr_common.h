// Shared helpers for the gfx950 kernels of libpcr_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "pcr.h"

#define PCR_EXPORT extern "C" __attribute__((visibility("default")))

#define PCR_CHECK_LAUNCH()                        \
  do {                                            \
    if (hipGetLastError() != hipSuccess) return PCR_ERR_LAUNCH; \
  } while (0)

static inline hipStream_t pcr_s(pcr_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

constexpr int kWave = 64;

// Diagnostics for bench.py's roofline object: the arithmetic (PCR_PREC_*) of the matrix phases of the model launch the
// calling thread issued last (pcr_last_launch_arith, probe_kernels.hip).  Every dispatcher of section B notes what the
// kernel it launches really runs -- the requested precision is only a request (shapes a bf16 unit does not hold run f32).
void pcr_note_arith(int prec);

// Tuning / ablation knobs (PCR_SA_DBG, PCR_TD_VARIANT, ...) are read from the environment ONLY in builds made with
// -DPCR_TUNING=1 (PCR_EXTRA_HIPCC_FLAGS); the shipped library ignores the environment on its launch paths.
#ifndef PCR_TUNING
#define PCR_TUNING 0
#endif
#include <stdlib.h>
static inline const char *pcr_tune_str(const char *name) { return PCR_TUNING ? getenv(name) : nullptr; }
static inline int pcr_tune_int(const char *name) {
  const char *v = pcr_tune_str(name);
  return v ? atoi(v) : 0;
}

// (x2-x1)^2+(y2-y1)^2+(z2-z1)^2, left to right, never contracted into fma: the file is built
// with -ffp-contract=off and the products are kept in separate statements.
__device__ __forceinline__ float pcr_sqdist3(float x1, float y1, float z1, float x2, float y2,
                                             float z2) {
  float dx = x2 - x1, dy = y2 - y1, dz = z2 - z1;
  float a = dx * dx;
  float b = dy * dy;
  float c = dz * dz;
  float s = a + b;
  return s + c;
}

// Monotone float -> uint32 map (total order of finite floats, -0 < +0).
__device__ __forceinline__ uint32_t pcr_orderable(float v) {
  uint32_t u = __float_as_uint(v);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

__device__ __forceinline__ unsigned long long pcr_wave_max_u64(unsigned long long k) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) {
    unsigned long long o = __shfl_xor(k, m, 64);
    k = o > k ? o : k;
  }
  return k;
}

__device__ __forceinline__ unsigned long long pcr_wave_min_u64(unsigned long long k) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) {
    unsigned long long o = __shfl_xor(k, m, 64);
    k = o < k ? o : k;
  }
  return k;
}

// ---- sorting network over the 64 lanes of a wave (bitonic, 21 compare-exchange steps) -------------------------
// Used where a wave ranks <= 64 keys (kNN selection): a rank count over v_readlane broadcasts costs 64 x 3
// instructions for 32-bit keys and ~64 x 7 for 64-bit keys; the network costs 21 x (exchange + compare + select).
// Exchanges: xor 1 / 2 and the mirrors inside 4 / 8 / 16 lanes are DPP modifiers (no extra instruction latency),
// xor 4 / 8 / 16 and the mirror inside 32 lanes are ds_swizzle bit-mode patterns, the one cross-half step uses
// gfx950's v_permlane32_swap.  Step (K, 0) "flips" a sorted block of K/2 against its mirror neighbour, steps (K, J)
// are the xor-J merges below it; after the (64, *) group the keys ascend with the lane index.
template <int CTRL>
__device__ __forceinline__ uint32_t pcr_dpp_u32(uint32_t v) {
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xF, 0xF, true);
}

__device__ __forceinline__ uint32_t pcr_lane_xor32(uint32_t v, bool upper_half) {
  const auto r = __builtin_amdgcn_permlane32_swap(v, v, false, false);   // {lanes 0..31 twice, lanes 32..63 twice}
  return upper_half ? r[0] : r[1];
}

template <int K, int J>
__device__ __forceinline__ uint32_t pcr_sort_partner(uint32_t v, int lane) {
  if constexpr (J == 0) {
    if constexpr (K == 2) return pcr_dpp_u32<0xB1>(v);                 // quad_perm [1,0,3,2]
    else if constexpr (K == 4) return pcr_dpp_u32<0x1B>(v);            // quad_perm [3,2,1,0]
    else if constexpr (K == 8) return pcr_dpp_u32<0x141>(v);           // row_half_mirror
    else if constexpr (K == 16) return pcr_dpp_u32<0x140>(v);          // row_mirror
    else if constexpr (K == 32) return (uint32_t)__builtin_amdgcn_ds_swizzle((int)v, 0x7C1F);   // xor 31
    else return pcr_lane_xor32((uint32_t)__builtin_amdgcn_ds_swizzle((int)v, 0x7C1F), lane >= 32);   // xor 63
  } else {
    if constexpr (J == 1) return pcr_dpp_u32<0xB1>(v);
    else if constexpr (J == 2) return pcr_dpp_u32<0x4E>(v);            // quad_perm [2,3,0,1]
    else if constexpr (J == 4) return (uint32_t)__builtin_amdgcn_ds_swizzle((int)v, 0x101F);
    else if constexpr (J == 8) return (uint32_t)__builtin_amdgcn_ds_swizzle((int)v, 0x201F);
    else return (uint32_t)__builtin_amdgcn_ds_swizzle((int)v, 0x401F);
  }
}

#define PCR_SORT_NETWORK(STEP)                                                                      \
  STEP(2, 0, 1)                                                                                     \
  STEP(4, 0, 2) STEP(4, 1, 1)                                                                       \
  STEP(8, 0, 4) STEP(8, 2, 2) STEP(8, 1, 1)                                                         \
  STEP(16, 0, 8) STEP(16, 4, 4) STEP(16, 2, 2) STEP(16, 1, 1)                                       \
  STEP(32, 0, 16) STEP(32, 8, 8) STEP(32, 4, 4) STEP(32, 2, 2) STEP(32, 1, 1)                       \
  STEP(64, 0, 32) STEP(64, 16, 16) STEP(64, 8, 8) STEP(64, 4, 4) STEP(64, 2, 2) STEP(64, 1, 1)

// ascending over the lanes; the lower lane of a pair keeps the smaller key
__device__ __forceinline__ uint32_t pcr_wave_sort_u32(uint32_t v, int lane) {
#define PCR_STEP32(K, J, BIT)                                   \
  {                                                             \
    const uint32_t o = pcr_sort_partner<K, J>(v, lane);         \
    const bool up = (lane & BIT) != 0;                          \
    v = ((o < v) != up) ? o : v;                                \
  }
  PCR_SORT_NETWORK(PCR_STEP32)
#undef PCR_STEP32
  return v;
}

// The same network for keys that are the BITS OF NON-NEGATIVE, NON-NaN FLOATS (denormals included: the kernels run
// with float_denorm_mode_32 = 3), two VALU instructions per step instead of three plus their vcc round trip.  A
// lane that keeps the larger key of its pair holds the key NEGATED: then both lanes of a pair execute the same
// x = min(x, -partner(x)) -- for the keeper of the minimum that is min(x, y), for the keeper of the maximum
// min(-y, -x) = -max(x, y) -- and the negation of the partner is a source modifier of v_min_f32 (on the DPP operand
// where the exchange is a DPP pattern).  Between two steps a lane whose role changes flips its sign (v_cndmask with a
// negated source under a constant lane mask).  Float order on these keys is their unsigned order.
constexpr unsigned long long pcr_lane_bit_mask(int bit) {   // the lanes whose bit `bit` (a power of two, 0 = none) is set
  return bit == 1 ? 0xAAAAAAAAAAAAAAAAull : bit == 2 ? 0xCCCCCCCCCCCCCCCCull : bit == 4 ? 0xF0F0F0F0F0F0F0F0ull
       : bit == 8 ? 0xFF00FF00FF00FF00ull : bit == 16 ? 0xFFFF0000FFFF0000ull : bit == 32 ? 0xFFFFFFFF00000000ull : 0ull;
}

template <int K, int J, int BIT>
__device__ __forceinline__ uint32_t pcr_sortf_step(uint32_t v, int lane) {
  // lanes with bit BIT set keep the larger key in this step; in the previous step it was bit PREV
  constexpr int PREV = J == 0 ? (K == 2 ? 0 : 1) : 2 * BIT;
  const unsigned long long flip = pcr_lane_bit_mask(PREV) ^ pcr_lane_bit_mask(BIT);
#define PCR_SORTF_DPP(CTRL)                                                                              \
  asm("v_cndmask_b32_e64 %0, %0, -%0, %1\n\ts_nop 1\n\t"                                                \
      "v_min_f32_dpp %0, -%0, %0 " CTRL " row_mask:0xf bank_mask:0xf" : "+v"(v) : "s"(flip))
  constexpr bool dpp = (J == 0 && K <= 16) || J == 1 || J == 2 || J == 8;
  if constexpr (J == 4) {
    // xor 4 inside a row of 16: the lanes of banks 0 / 2 take lane + 4, those of banks 1 / 3 lane - 4 -- two DPP
    // minima under bank masks instead of a ds_swizzle round trip
    uint32_t w;
    asm("v_cndmask_b32_e64 %1, %1, -%1, %2\n\ts_nop 1\n\t"
        "v_min_f32_dpp %0, -%1, %1 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
        "v_min_f32_dpp %0, -%1, %1 row_shr:4 row_mask:0xf bank_mask:0xa"
        : "=&v"(w), "+v"(v) : "s"(flip));
    v = w;
  } else if constexpr (dpp) {
    if constexpr (J == 1 || (J == 0 && K == 2)) PCR_SORTF_DPP("quad_perm:[1,0,3,2]");
    else if constexpr (J == 2) PCR_SORTF_DPP("quad_perm:[2,3,0,1]");
    else if constexpr (J == 8) PCR_SORTF_DPP("row_ror:8");
    else if constexpr (K == 4) PCR_SORTF_DPP("quad_perm:[3,2,1,0]");
    else if constexpr (K == 8) PCR_SORTF_DPP("row_half_mirror");
    else PCR_SORTF_DPP("row_mirror");
  } else {
    asm("v_cndmask_b32_e64 %0, %0, -%0, %1" : "+v"(v) : "s"(flip));
    const uint32_t o = pcr_sort_partner<K, J>(v, lane);
    asm("v_min_f32_e64 %0, %1, -%2" : "=v"(v) : "v"(v), "v"(o));
  }
#undef PCR_SORTF_DPP
  return v;
}

__device__ __forceinline__ uint32_t pcr_wave_sort_posf32(uint32_t v, int lane) {
#define PCR_STEPF(K, J, BIT) v = pcr_sortf_step<K, J, BIT>(v, lane);
  PCR_SORT_NETWORK(PCR_STEPF)
#undef PCR_STEPF
  return v & 0x7FFFFFFFu;   // the last step's keepers of the maximum (odd lanes) still hold their key negated
}

__device__ __forceinline__ unsigned long long pcr_wave_sort_u64(unsigned long long key, int lane) {
  uint32_t hi = (uint32_t)(key >> 32), lo = (uint32_t)key;
#define PCR_STEP64(K, J, BIT)                                                        \
  {                                                                                  \
    const uint32_t ohi = pcr_sort_partner<K, J>(hi, lane), olo = pcr_sort_partner<K, J>(lo, lane); \
    const bool less = (((unsigned long long)ohi << 32) | olo) < (((unsigned long long)hi << 32) | lo); \
    const bool up = (lane & BIT) != 0;                                               \
    const bool take = less != up;                                                    \
    hi = take ? ohi : hi;                                                            \
    lo = take ? olo : lo;                                                            \
  }
  PCR_SORT_NETWORK(PCR_STEP64)
#undef PCR_STEP64
  return ((unsigned long long)hi << 32) | lo;
}

// ---- MFMA token (round 6).  The counters say the wave-autonomous kernels overlap almost nothing: matrix pipe busy 0.57 +
// VALU issue 0.31 of a launch's cycles with 0.10 of them in both states (SQ_VALU_MFMA_COEXEC_CYCLES; 0.014 for the kv
// kernel) -- the two waves of a SIMD run the same phases on equal blocks and fall into step, so the VALU phase of one
// does not sit under the MFMA phase of the other.  A token per SIMD (an LDS word: the waves of a workgroup that share a
// SIMD are found by HW_ID.SIMD_ID) makes the matrix phases of the SIMD's waves mutually exclusive: while one wave holds
// it and runs layers 2 / 3, the other gathers, seeds, splits, reduces -- or sleeps at the gate.
// Measured (tools/scratch/probe_coexec*.hip, one wave per SIMD with an instruction-level interleave): under a running
// v_mfma_f32_32x32x16_bf16 integer / transcendental / LDS instructions are nearly free, plain f32 VALU instructions keep
// about two of their three cycles, packed f32 instructions (v_pk_add / mul / fma_f32) hide almost nothing -- the matrix pipe and
// the f32 vector lanes are one datapath.  So the token buys what the integer part of the other wave's VALU phase is worth:
// sa_stream_kernel<4,4> 2.20-2.26 -> 2.06-2.09 ms (pt1024 SA3, same bits); the attention stream kernels (four short matrix
// phases per block, f32-heavy LayerNorm / normaliser between them) gain nothing from it (measured: 0 to -2 %): not used there.
__device__ __forceinline__ int pcr_simd_id() { return __builtin_amdgcn_s_getreg((1 << 11) | (4 << 6) | 4); }   // HW_ID[5:4]
__device__ __forceinline__ void mfma_token_acquire(int *tok, int lane) {
  if (!tok) return;
  __builtin_amdgcn_sched_barrier(0);
  int busy;
  do {
    int old = 1;
    if (lane == 0) old = atomicCAS(tok, 0, 1);
    busy = __builtin_amdgcn_readfirstlane(old);
    if (busy) __builtin_amdgcn_s_sleep(1);
  } while (busy);
  __builtin_amdgcn_sched_barrier(0);
}
__device__ __forceinline__ void mfma_token_release(int *tok, int lane) {
  if (!tok) return;
  __builtin_amdgcn_sched_barrier(0);
  if (lane == 0) __hip_atomic_store(tok, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  __builtin_amdgcn_sched_barrier(0);
}

