// Point ops of the ReID hot path for gfx950: FPS, ball query, heap kNN, gather / group,
// three-NN and three-interpolate.  Semantics (and every index output, bit for bit) follow the
// reference's dormant CUDA ops under mmdet3d/ops (cited in include/pcr.h); the kernels
// themselves are designed for CDNA4: 64-wide waves, cloud-resident LDS tiles with broadcast
// reads, one barrier per FPS step, register-resident running distances.
//
// Build with -ffp-contract=off (distance expressions must not be fused; see pcr_common.h).
#include <type_traits>

#include "pcr_common.h"

namespace {

// ------------------------------------------------------------------------------- FPS ----
// One workgroup per cloud.  Thread tid owns the candidates k = tid, tid+block, ... exactly as
// in the reference (furthest_point_sample_cuda.cu:56-71) so its per-thread candidate (first
// maximum in increasing k, initial best=-1/besti=0) is identical; the cross-thread merge is a
// max over a packed 64-bit key that encodes the total order of the reference's halving tree
// (__update, :17-23): larger value wins, ties go to the slot that survives the tree, which is
// the tid with the smallest BIT-REVERSED value (stride-s step keeps tid over tid+s).
constexpr int kFpsPpt = 4;        // candidates per thread held in registers (N <= 4*block)
constexpr int kFpsLdsPts = 4096;  // clouds up to this size are staged in LDS (48 KiB)

// `start` / `pytie`: the Python twin of the model path (models/pointnet2_utils.py:116-137): the first pick is start[cloud]
// (the reference draws it with torch.randint) and equal distances go to the LOWEST index (torch.max).
template <bool DIST, bool REG>
__global__ void fps_kernel(const float *__restrict__ data, float *__restrict__ temp,
                           int *__restrict__ idxs, int n, int m, int block, int logb,
                           int stage_xyz, const int *__restrict__ start, int pytie) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  unsigned long long *skey = reinterpret_cast<unsigned long long *>(smem_raw);  // [2][16]
  float *sx = reinterpret_cast<float *>(smem_raw + 256);

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int nw = (blockDim.x + 63) >> 6;
  const size_t cloud = blockIdx.x;
  data += DIST ? cloud * n * n : cloud * n * 3;
  temp += cloud * n;
  idxs += cloud * m;
  if (m <= 0) return;

  if (!DIST && stage_xyz)
    for (int i = tid; i < 3 * n; i += blockDim.x) sx[i] = data[i];

  float px[kFpsPpt], py[kFpsPpt], pz[kFpsPpt], t[kFpsPpt];
  if (REG) {
#pragma unroll
    for (int p = 0; p < kFpsPpt; p++) {
      int k = tid + p * block;
      bool ok = tid < block && k < n;
      t[p] = ok ? temp[k] : 0.f;
      px[p] = py[p] = pz[p] = 0.f;
      if (!DIST && ok) { px[p] = data[k * 3]; py[p] = data[k * 3 + 1]; pz[p] = data[k * 3 + 2]; }
    }
  }
  __syncthreads();

  const uint32_t rev = logb ? (__brev((uint32_t)tid) >> (32 - logb)) : 0u;
  const uint32_t tie = pytie ? 0u : (uint32_t)(block - 1) - rev;  // larger = preferred by the merge tree
  int old = start ? start[cloud] : 0;
  old = (unsigned)old < (unsigned)n ? old : 0;   // (a start outside the cloud has no point to begin from: clamped, the wrapper validates)
  if (tid == 0) idxs[0] = old;

  for (int j = 1; j < m; j++) {
    float x1 = 0.f, y1 = 0.f, z1 = 0.f;
    if (!DIST) {
      if (stage_xyz) { x1 = sx[old * 3]; y1 = sx[old * 3 + 1]; z1 = sx[old * 3 + 2]; }
      else { x1 = data[old * 3]; y1 = data[old * 3 + 1]; z1 = data[old * 3 + 2]; }
    }
    float best = -1.f;
    int besti = 0;
    if (tid < block) {
      if (REG) {
#pragma unroll
        for (int p = 0; p < kFpsPpt; p++) {
          int k = tid + p * block;
          if (k < n) {
            float d = DIST ? data[(size_t)old * n + k] : pcr_sqdist3(x1, y1, z1, px[p], py[p], pz[p]);
            float d2 = fminf(d, t[p]);
            t[p] = d2;
            besti = d2 > best ? k : besti;
            best = d2 > best ? d2 : best;
          }
        }
      } else {
        for (int k = tid; k < n; k += block) {
          float d;
          if (DIST) d = data[(size_t)old * n + k];
          else if (stage_xyz) d = pcr_sqdist3(x1, y1, z1, sx[k * 3], sx[k * 3 + 1], sx[k * 3 + 2]);
          else d = pcr_sqdist3(x1, y1, z1, data[k * 3], data[k * 3 + 1], data[k * 3 + 2]);
          float d2 = fminf(d, temp[k]);
          temp[k] = d2;
          besti = d2 > best ? k : besti;
          best = d2 > best ? d2 : best;
        }
      }
    }
    unsigned long long key = 0ull;
    if (tid < block)
      key = ((unsigned long long)pcr_orderable(best + 0.0f) << 32) |
            (unsigned long long)((tie << 22) | (pytie ? 0x3FFFFFu - (uint32_t)besti : (uint32_t)besti));
    key = pcr_wave_max_u64(key);
    if (nw > 1) {
      unsigned long long *slot = skey + (j & 1) * 16;
      if (lane == 0) slot[wave] = key;
      __syncthreads();
      int w = lane & 15;
      key = w < nw ? slot[w] : 0ull;
#pragma unroll
      for (int s = 8; s >= 1; s >>= 1) {
        unsigned long long o = __shfl_xor(key, s, 64);
        key = o > key ? o : key;
      }
    }
    old = (int)(key & 0x3FFFFFull);
    if (pytie) old = 0x3FFFFF - old;
    if (tid == 0) idxs[j] = old;
  }

  if (REG) {
#pragma unroll
    for (int p = 0; p < kFpsPpt; p++) {
      int k = tid + p * block;
      if (tid < block && k < n) temp[k] = t[p];
    }
  }
}

// Fast D-FPS for N <= 4096: 256 threads per cloud, PPT points per thread in registers, the cloud SoA
// in LDS for the broadcast read of the last pick, one barrier per step.  Same result as fps_kernel:
// the reference's outcome is the maximum of the total order (distance, merge-tree preference of
// k mod block, smaller k); each point carries that order in a packed 64-bit key, reduced with
// in-register DPP moves inside 16-lane rows, v_readlane across rows and a 4-entry LDS slot across
// waves.  (Only for coordinates: there d2 >= 0 > -1, so the reference's "best = -1" start never
// matters; the distance-matrix variant keeps the literal per-thread scan of fps_kernel.)
template <int CTRL>
__device__ __forceinline__ uint32_t dpp_u32(uint32_t v) {
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xF, 0xF, true);
}

template <int PPT, int NT>
__global__ __launch_bounds__(NT) void fps_fast_kernel(const float *__restrict__ xyz, float *__restrict__ temp,
                                                       int *__restrict__ idxs, int n, int m, int block,
                                                       int logb) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  unsigned long long *skey = reinterpret_cast<unsigned long long *>(smem_raw);  // [2][4]
  float *sx = reinterpret_cast<float *>(smem_raw + 64);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const size_t cloud = blockIdx.x;
  xyz += cloud * n * 3;
  temp += cloud * n;
  idxs += cloud * m;
  for (int i = tid; i < 3 * n; i += NT) sx[i] = xyz[i];
  __syncthreads();
  float px[PPT], py[PPT], pz[PPT], t[PPT];
  uint32_t low[PPT];
#pragma unroll
  for (int p = 0; p < PPT; p++) {
    const int k = tid + NT * p;
    const bool ok = k < n;
    px[p] = ok ? sx[3 * k] : 0.f;
    py[p] = ok ? sx[3 * k + 1] : 0.f;
    pz[p] = ok ? sx[3 * k + 2] : 0.f;
    t[p] = ok ? temp[k] : 0.f;
    const uint32_t tr = (uint32_t)k & (uint32_t)(block - 1);
    const uint32_t rev = logb ? (__brev(tr) >> (32 - logb)) : 0u;
    low[p] = ok ? ((((uint32_t)(block - 1) - rev) << 22) | (0x3FFFFFu - (uint32_t)k)) : 0u;
  }
  int old = 0;
  if (tid == 0) idxs[0] = 0;
  for (int j = 1; j < m; j++) {
    const float x1 = sx[3 * old], y1 = sx[3 * old + 1], z1 = sx[3 * old + 2];
    uint32_t hi = 0u, lo = 0u;
#pragma unroll
    for (int p = 0; p < PPT; p++) {
      const float d = pcr_sqdist3(x1, y1, z1, px[p], py[p], pz[p]);
      const float d2 = fminf(d, t[p]);
      t[p] = d2;
      const uint32_t h2 = low[p] ? pcr_orderable(d2 + 0.0f) : 0u;   // points beyond n never win
      const bool take = h2 > hi || (h2 == hi && low[p] > lo);
      hi = take ? h2 : hi;
      lo = take ? low[p] : lo;
    }
#define PCR_FPS_STEP(CTRL)                                         \
    {                                                              \
      const uint32_t oh = dpp_u32<CTRL>(hi), ol = dpp_u32<CTRL>(lo); \
      const bool take = oh > hi || (oh == hi && ol > lo);          \
      hi = take ? oh : hi;                                         \
      lo = take ? ol : lo;                                         \
    }
    PCR_FPS_STEP(0xB1)   // lane ^ 1
    PCR_FPS_STEP(0x4E)   // lane ^ 2
    PCR_FPS_STEP(0x141)  // row_half_mirror
    PCR_FPS_STEP(0x140)  // row_mirror: every lane of a 16-lane row now holds the row maximum
#undef PCR_FPS_STEP
    unsigned long long key = 0ull;
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const unsigned long long kr = ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)hi, 16 * r) << 32) |
                                    (uint32_t)__builtin_amdgcn_readlane((int)lo, 16 * r);
      key = kr > key ? kr : key;
    }
    if (NT > 64) {   // single-wave workgroups (NT == 64) need no exchange and no barrier at all
      unsigned long long *slot = skey + (j & 1) * 4;
      if (lane == 0) slot[wave] = key;
      __syncthreads();
#pragma unroll
      for (int w = 0; w < NT / 64; w++) {
        const unsigned long long kw = slot[w];
        key = kw > key ? kw : key;
      }
    }
    old = (int)(0x3FFFFFu - (uint32_t)(key & 0x3FFFFFull));
    if (tid == 0) idxs[j] = old;
  }
#pragma unroll
  for (int p = 0; p < PPT; p++) {
    const int k = tid + NT * p;
    if (k < n) temp[k] = t[p];
  }
}

// One WAVE per cloud, up to 1024 points, no barrier and no cross-wave exchange: lane l keeps points
// l, l+64, ... in registers as pairs (packed f32 subtract / multiply / add: same IEEE operations as
// pcr_sqdist3, two points per instruction), the running maximum of the updated distances comes from a
// v_max3 chain, and the reference's tie order (see fps_fast_kernel) is only evaluated among the points
// that ATTAIN the wave maximum.  Coordinates must be finite (as everywhere in this file).
typedef float f32x2v __attribute__((ext_vector_type(2)));
// Two points side by side.  PCR_POINT_PACKED=1 maps the arithmetic onto v_pk_{add,mul}_f32; 0 keeps plain scalar
// instructions (same IEEE operations either way; which is faster is a measurement)
#ifndef PCR_POINT_PACKED
#define PCR_POINT_PACKED 1
#endif
#if PCR_POINT_PACKED
typedef f32x2v f32x2;
#else
struct f32x2 {
  float a, b;
  __device__ __forceinline__ float &operator[](int i) { return i ? b : a; }
  __device__ __forceinline__ float operator[](int i) const { return i ? b : a; }
};
__device__ __forceinline__ f32x2 operator-(const f32x2 &x, const f32x2 &y) { return f32x2{x.a - y.a, x.b - y.b}; }
__device__ __forceinline__ f32x2 operator+(const f32x2 &x, const f32x2 &y) { return f32x2{x.a + y.a, x.b + y.b}; }
__device__ __forceinline__ f32x2 operator*(const f32x2 &x, const f32x2 &y) { return f32x2{x.a * y.a, x.b * y.b}; }
#endif
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t dpp_max_u32(uint32_t v) {
  uint32_t o;
  o = dpp_u32<0xB1>(v); v = o > v ? o : v;
  o = dpp_u32<0x4E>(v); v = o > v ? o : v;
  o = dpp_u32<0x141>(v); v = o > v ? o : v;
  o = dpp_u32<0x140>(v); v = o > v ? o : v;
  const uint32_t a = (uint32_t)__builtin_amdgcn_readlane((int)v, 0), b = (uint32_t)__builtin_amdgcn_readlane((int)v, 16);
  const uint32_t c = (uint32_t)__builtin_amdgcn_readlane((int)v, 32), d = (uint32_t)__builtin_amdgcn_readlane((int)v, 48);
  const uint32_t ab = a > b ? a : b, cd = c > d ? c : d;
  return ab > cd ? ab : cd;
}

// lane `l` (wave-uniform) of a register := a wave-uniform value (v_writelane_b32; this clang has no builtin for it, and on
// gfx9 a second scalar operand must be M0: one scalar register per vector instruction)
__device__ __forceinline__ int pcr_writelane(int v, int l, int into) {
  asm("s_mov_b32 m0, %2\n\ts_nop 0\n\tv_writelane_b32 %0, %1, m0" : "+v"(into) : "s"(v), "s"(l) : "m0");
  return into;
}
__device__ __forceinline__ void pcr_writelane4(int v0, int v1, int v2, int v3, int l, int &a0, int &a1, int &a2, int &a3) {
  asm("s_mov_b32 m0, %8\n\ts_nop 0\n\tv_writelane_b32 %0, %4, m0\n\tv_writelane_b32 %1, %5, m0\n\t"
      "v_writelane_b32 %2, %6, m0\n\tv_writelane_b32 %3, %7, m0"
      : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3)
      : "s"(v0), "s"(v1), "s"(v2), "s"(v3), "s"(l)
      : "m0");
}

// Which of a lane's NS slots hold the wave maximum `best`?  Every lane shifts one compare per slot into a private word
// (v_cmp_eq + v_addc_co: word = 2 word + carry, slot p ends up at bit NS - 1 - p) and ONE v_readlane takes lane L's word
// to the scalar side: 2 NS vector + 3 scalar instructions.  (Before: a ballot per slot and a scalar bit test of each mask
// -- 85 scalar instructions per pick of a 1024-point cloud, more than its distance arithmetic; a wave's scalar and
// vector instructions share its issue slots)
template <int NS>
__device__ __forceinline__ uint32_t fps_slot_bits(const uint32_t (&t)[NS], uint32_t best, int L) {
  static_assert(NS % 2 == 0, "slots come in pairs");
  uint32_t w = 0u;
  // (one asm statement per group of slots: the compiler pads every statement's end with an s_nop)
#define PCR_SB1(a) "v_cmp_eq_u32 vcc, %[b], %[" #a "]\n\tv_addc_co_u32 %[w], vcc, %[w], %[w], vcc\n\t"
  if constexpr (NS % 8 == 0) {
#pragma unroll
    for (int p = 0; p < NS; p += 8)
      asm(PCR_SB1(t0) PCR_SB1(t1) PCR_SB1(t2) PCR_SB1(t3) PCR_SB1(t4) PCR_SB1(t5) PCR_SB1(t6) PCR_SB1(t7)
          : [w] "+v"(w)
          : [b] "v"(best), [t0] "v"(t[p]), [t1] "v"(t[p + 1]), [t2] "v"(t[p + 2]), [t3] "v"(t[p + 3]), [t4] "v"(t[p + 4]),
            [t5] "v"(t[p + 5]), [t6] "v"(t[p + 6]), [t7] "v"(t[p + 7])
          : "vcc");
  } else {
#pragma unroll
    for (int p = 0; p < NS; p += 2)
      asm(PCR_SB1(t0) PCR_SB1(t1) : [w] "+v"(w) : [b] "v"(best), [t0] "v"(t[p]), [t1] "v"(t[p + 1]) : "vcc");
  }
#undef PCR_SB1
  return (uint32_t)__builtin_amdgcn_readlane((int)w, L);
}

// The picked point's coordinates come straight out of the register file: the pick `old` is wave-uniform, so its slot
// (old / 64) selects a REGISTER -- a wave-uniform dynamic index into the per-axis vector of a lane's coordinates, which
// the compiler turns into relative register addressing (s_set_gpr_idx_on + v_mov_b32) -- and v_readlane fetches lane
// old % 64 of it: ~12 instructions, no branch.  (Rounds 2-4, and still for clouds of <= 512 points: a scalar `switch` over
// the slot, which the structuriser compiles into a chain of fall-through flags -- ~35 scalar instructions and eight
// branches per pick of a 1024-point cloud.)  (Round 2 kept a
// {x,y,z,-} copy of the cloud in LDS for one broadcast read per pick: 16 KB per wave, which held a CU to ten waves --
// 2.5 per SIMD, 4096 clouds = 1.6 rounds -- and put an LDS round trip on every pick's critical path; without it four
// waves share a SIMD and the chip takes 4096 clouds in one round.)
template <int PP>   // point PAIRS per lane: n <= 128 * PP
__global__ __launch_bounds__(64) void fps_wave_kernel(const float *__restrict__ xyz, float *__restrict__ temp,
                                                      int *__restrict__ idxs, int n, int m, int block, int logb) {
  const int lane = threadIdx.x;
  const size_t cloud = blockIdx.x;
  xyz += cloud * n * 3;
  temp += cloud * n;
  idxs += cloud * m;
  // the running minimum distances are kept as BITS: for floats >= +0 the unsigned order of the bit
  // patterns is the float order, and integer min / max need no NaN canonicalisation instructions
  // a lane's coordinates as ONE vector per axis: slot p = element p (pairs (2 p, 2 p + 1) feed the packed arithmetic; the
  // pick reads element `slot` with a wave-uniform DYNAMIC index, i.e. relative register addressing, see fps_pick_coord)
  typedef float fvec __attribute__((ext_vector_type(2 * PP)));
  fvec vx, vy, vz;
  // the vector forms of the slot search and of the coordinate fetch move ~100 scalar instructions per pick of a
  // 1024-point cloud to ~40 vector ones; clouds of <= 512 points have half the slots, run at higher occupancy (scalar
  // and vector instructions of different waves issue side by side) and are faster with the scalar forms (measured)
  constexpr bool kVecPick = PP >= 8;
  uint32_t t[2 * PP], low[2 * PP];
#pragma unroll
  for (int p = 0; p < 2 * PP; p++) {
    const int k = lane + 64 * p;
    const bool ok = k < n;
    const float *q = xyz + (size_t)(ok ? k : 0) * 3;
    const float x = q[0], y = q[1], z = q[2], tk = temp[ok ? k : 0];
    vx[p] = ok ? x : 0.f;
    vy[p] = ok ? y : 0.f;
    vz[p] = ok ? z : 0.f;
    t[p] = ok ? __float_as_uint(tk) : 0u;   // min(d, 0) = 0: a point beyond n never beats a real one (low = 0)
    const uint32_t tr = (uint32_t)k & (uint32_t)(block - 1);
    const uint32_t rev = logb ? (__brev(tr) >> (32 - logb)) : 0u;
    low[p] = ok ? ((((uint32_t)(block - 1) - rev) << 22) | (0x3FFFFFu - (uint32_t)k)) : 0u;
    asm volatile("" : "+v"(low[p]));   // keep the value in a register (else it is re-derived from masks every step)
  }
  int old = 0;
  if (lane == 0) idxs[0] = 0;
  for (int j = 1; j < m; j++) {
    const int slot = __builtin_amdgcn_readfirstlane(old >> 6), ln = __builtin_amdgcn_readfirstlane(old & 63);
    float ox = 0.f, oy = 0.f, oz = 0.f;
    if constexpr (kVecPick) {
      ox = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(vx[slot]), ln));
      oy = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(vy[slot]), ln));
      oz = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(vz[slot]), ln));
    } else {
#define PCR_FPS_PICK(P)                                                               \
  case P:                                                                             \
    if constexpr (P < 2 * PP) {                                                       \
      ox = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(vx[P]), ln));      \
      oy = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(vy[P]), ln));      \
      oz = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(vz[P]), ln));      \
    }                                                                                 \
    break;
      switch (slot) {
        PCR_FPS_PICK(0) PCR_FPS_PICK(1) PCR_FPS_PICK(2) PCR_FPS_PICK(3) PCR_FPS_PICK(4) PCR_FPS_PICK(5) PCR_FPS_PICK(6)
        PCR_FPS_PICK(7)
        default: break;
      }
#undef PCR_FPS_PICK
    }
    const f32x2 x1 = {ox, ox}, y1 = {oy, oy}, z1 = {oz, oz};
    uint32_t mx = 0u;
#pragma unroll
    for (int p = 0; p < PP; p++) {
      const f32x2 dx = f32x2{vx[2 * p], vx[2 * p + 1]} - x1, dy = f32x2{vy[2 * p], vy[2 * p + 1]} - y1,
                  dz = f32x2{vz[2 * p], vz[2 * p + 1]} - z1;
      const f32x2 a = dx * dx;
      const f32x2 b = dy * dy;
      const f32x2 c = dz * dz;
      const f32x2 s = a + b;
      const f32x2 d = s + c;
      const uint32_t d0 = __float_as_uint(d[0]), d1 = __float_as_uint(d[1]);
      t[2 * p] = d0 < t[2 * p] ? d0 : t[2 * p];
      t[2 * p + 1] = d1 < t[2 * p + 1] ? d1 : t[2 * p + 1];
      const uint32_t m01 = t[2 * p] > t[2 * p + 1] ? t[2 * p] : t[2 * p + 1];
      mx = m01 > mx ? m01 : mx;
    }
    const uint32_t best = dpp_max_u32(mx);
    // which point holds it?  Almost always ONE lane and, in it, ONE slot: then the pick is 64 slot + lane, found by a scalar
    // walk over the slots' equality masks (16 compares + scalar bit tests instead of the 40-instruction tie-break pass and
    // its second wave reduction).  Anything else -- two lanes, or two slots of the lane -- takes the reference's tie rule
    // below, unchanged.
    const unsigned long long lm = __ballot(mx == best);
    bool unique = __popcll(lm) == 1;
    if (unique) {
      const int L = (int)__builtin_ctzll(lm);
      if constexpr (kVecPick) {
        const uint32_t sb = fps_slot_bits<2 * PP>(t, best, L);
        if (__popc(sb) == 1) old = 64 * (2 * PP - 1 - (int)__builtin_ctz(sb)) + L;
        else unique = false;
      } else {
        int hits = 0, slot_found = 0;
#pragma unroll
        for (int p = 0; p < 2 * PP; p++) {
          const unsigned long long mp = __ballot(t[p] == best);
          const int bit = (int)((mp >> L) & 1ull);
          hits += bit;
          slot_found = bit ? p : slot_found;
        }
        if (hits == 1) old = 64 * slot_found + L;
        else unique = false;
      }
    }
    if (!unique) {
      uint32_t lo = 0u;
#pragma unroll
      for (int p = 0; p < PP; p++) {   // (pairs: the two selects feed one three-way max)
        const uint32_t l0 = t[2 * p] == best ? low[2 * p] : 0u;
        const uint32_t l1 = t[2 * p + 1] == best ? low[2 * p + 1] : 0u;
        const uint32_t l01 = l0 > l1 ? l0 : l1;
        lo = l01 > lo ? l01 : lo;
      }
      lo = dpp_max_u32(lo);
      old = (int)(0x3FFFFFu - (lo & 0x3FFFFFu));
    }
    if (lane == 0) idxs[j] = old;
  }
#pragma unroll
  for (int p = 0; p < 2 * PP; p++) {
    const int k = lane + 64 * p;
    if (k < n) temp[k] = __uint_as_float(t[p]);
  }
}

// D-FPS and the ball query of the picked centres in ONE pass (round 5).  A pick's distances to every point of the cloud
// are exactly what the ball query of that centre needs -- (p - c)^2 summed in the same order, so the bits are those of
// ball_query_reg_kernel's (c - p)^2 -- and fps_wave_kernel computes them anyway for its running minimum: the fused
// kernel tests them against the radius while they are in registers (one compare per point) and compacts the few hits
// exactly as ball_query_reg_kernel<.., kRows> does.  Output: the pick order (idxs), the centres' coordinates
// (new_xyz: no gather launch), the hit counts and the SA kernel's row table -- entry for entry what the two separate
// launches write (tests/test_gpu_fps_bq.py).  One wave per cloud, up to 1024 points, K even, min radius 0.
template <int PP>
__global__ __launch_bounds__(64) void fps_bq_wave_kernel(const float *__restrict__ xyz, float *__restrict__ temp,
                                                         int *__restrict__ idxs, float *__restrict__ new_xyz,
                                                         int *__restrict__ cnt_out, f32x4 *__restrict__ rows_out,
                                                         int n, int m, int block, int logb, float max_r2, int K) {
  const int lane = threadIdx.x;
  const size_t cloud = blockIdx.x;
  xyz += cloud * n * 3;
  temp += cloud * n;
  idxs += cloud * m;
  new_xyz += cloud * m * 3;
  cnt_out += cloud * m;
  constexpr int cpw = 16;
  const int nitems = (m + cpw - 1) / cpw;
  f32x4 *rcloud = rows_out + cloud * (size_t)nitems * (size_t)(cpw * K);
  // a lane's coordinates as ONE vector per axis: slot p = element p (pairs (2 p, 2 p + 1) feed the packed arithmetic; the
  // pick reads element `slot` with a wave-uniform DYNAMIC index, i.e. relative register addressing, see fps_pick_coord)
  typedef float fvec __attribute__((ext_vector_type(2 * PP)));
  fvec vx, vy, vz;
  // the vector forms of the slot search and of the coordinate fetch move ~100 scalar instructions per pick of a
  // 1024-point cloud to ~40 vector ones; clouds of <= 512 points have half the slots, run at higher occupancy (scalar
  // and vector instructions of different waves issue side by side) and are faster with the scalar forms (measured)
  constexpr bool kVecPick = PP >= 8;
  uint32_t t[2 * PP];
#pragma unroll
  for (int p = 0; p < 2 * PP; p++) {
    const int k = lane + 64 * p;
    const bool ok = k < n;
    const float *q = xyz + (size_t)(ok ? k : 0) * 3;
    const float x = q[0], y = q[1], z = q[2], tk = temp[ok ? k : 0];
    vx[p] = ok ? x : INFINITY;    // a point at infinity is never inside a ball (and min(inf, 0) = 0 below)
    vy[p] = ok ? y : INFINITY;
    vz[p] = ok ? z : INFINITY;
    t[p] = ok ? __float_as_uint(tk) : 0u;     // min(d, 0) = 0: a point beyond n never beats a real one
  }
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  // one buffer descriptor for the cloud's whole row region (items of cpw K rows follow each other; a descriptor per item
  // was a 64-bit multiply + rebuild per pick); roff = the next free row counted from the cloud's first
  const __amdgpu_buffer_rsrc_t rrows =
      __builtin_amdgcn_make_buffer_rsrc(rcloud, 0, (int)((size_t)nitems * (size_t)(cpw * K) * 16), 0x00020000);
  int old = 0, roff = 0;
  int acc_i = 0, acc_x = 0, acc_y = 0, acc_z = 0, acc_c = 0;
  for (int j = 0; j < m; j++) {
    if (j == m - 1) {   // the running minima as pcr_fps_f32 leaves them: the last centre does not enter them
#pragma unroll
      for (int p = 0; p < 2 * PP; p++) {
        const int k = lane + 64 * p;
        if (k < n) temp[k] = __uint_as_float(t[p]);
      }
    }
    // ---- centre j = point `old`: its coordinates out of the register file
    const int slot = __builtin_amdgcn_readfirstlane(old >> 6), ln = __builtin_amdgcn_readfirstlane(old & 63);
    float ox = 0.f, oy = 0.f, oz = 0.f;
    if constexpr (kVecPick) {
      ox = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(vx[slot]), ln));
      oy = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(vy[slot]), ln));
      oz = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(vz[slot]), ln));
    } else {
#define PCR_FPS_PICK(P)                                                               \
  case P:                                                                             \
    if constexpr (P < 2 * PP) {                                                       \
      ox = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(vx[P]), ln));      \
      oy = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(vy[P]), ln));      \
      oz = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(vz[P]), ln));      \
    }                                                                                 \
    break;
      switch (slot) {
        PCR_FPS_PICK(0) PCR_FPS_PICK(1) PCR_FPS_PICK(2) PCR_FPS_PICK(3) PCR_FPS_PICK(4) PCR_FPS_PICK(5) PCR_FPS_PICK(6)
        PCR_FPS_PICK(7)
        default: break;
      }
#undef PCR_FPS_PICK
    }
    // (what a pick leaves behind -- index, centre, hit count -- is parked in lane j % 64 of five registers by v_writelane
    // and leaves as whole stores every 64 picks: a lane-0 store per pick and value was ~20 instructions behind an exec
    // mask)
    const int jl = j & 63;
    pcr_writelane4(__builtin_amdgcn_readfirstlane(old), __float_as_int(ox), __float_as_int(oy), __float_as_int(oz), jl, acc_i,
                   acc_x, acc_y, acc_z);
    // ---- every point's distance to it: hit masks, the running minimum and its lane maximum
    const f32x2 x1 = {ox, ox}, y1 = {oy, oy}, z1 = {oz, oz};
    unsigned long long mk[2 * PP];
    bool hit[2 * PP];
    uint32_t mx = 0u;
#pragma unroll
    for (int p = 0; p < PP; p++) {
      const f32x2 dx = f32x2{vx[2 * p], vx[2 * p + 1]} - x1, dy = f32x2{vy[2 * p], vy[2 * p + 1]} - y1,
                  dz = f32x2{vz[2 * p], vz[2 * p + 1]} - z1;
      const f32x2 a = dx * dx;
      const f32x2 b = dy * dy;
      const f32x2 c = dz * dz;
      const f32x2 s = a + b;
      const f32x2 d = s + c;
      hit[2 * p] = d[0] < max_r2;         // (kept as lane masks: the hit blocks below take "my point" from them)
      hit[2 * p + 1] = d[1] < max_r2;
      mk[2 * p] = __ballot(hit[2 * p]);
      mk[2 * p + 1] = __ballot(hit[2 * p + 1]);
      const uint32_t d0 = __float_as_uint(d[0]), d1 = __float_as_uint(d[1]);
      t[2 * p] = d0 < t[2 * p] ? d0 : t[2 * p];
      t[2 * p + 1] = d1 < t[2 * p + 1] ? d1 : t[2 * p + 1];
      // (pins the update and the running maximum HERE: left alone the compiler sinks the minima and the maximum chain
      // below the ball section and keeps the 2 PP distances -- or a second copy of t -- alive across it)
      asm volatile("" : "+v"(t[2 * p]), "+v"(t[2 * p + 1]));
      const uint32_t m01 = t[2 * p] > t[2 * p + 1] ? t[2 * p] : t[2 * p + 1];
      mx = m01 > mx ? m01 : mx;
      asm volatile("" : "+v"(mx));
    }
    // ---- the ball of centre j: ordered compaction of the hits into the item's row region (ball_query_reg_kernel).
    // (opaque copies of the centre for the hit branch: with the same values the compiler keeps all 6 PP differences of the
    // distance loop alive for the rare rows -- 48 registers, half the occupancy)
    {
      int oxi = __float_as_int(ox), oyi = __float_as_int(oy), ozi = __float_as_int(oz);
      asm volatile("" : "+s"(oxi), "+s"(oyi), "+s"(ozi));
      const float cx = __int_as_float(oxi), cy = __int_as_float(oyi), cz = __int_as_float(ozi);
      auto put_row = [&](int rslot, int soff, int i, float dx, float dy, float dz) __attribute__((always_inline)) {
        // (scalar-offset field 0 on purpose: see ball_query_reg_kernel)
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, f32x4{__int_as_float(i), dx, dy, dz}), rrows,
                                               (rslot + soff) * 16, 0, 0);
      };
      int cnt = 0, first = 0;
      float fdx = 0.f, fdy = 0.f, fdz = 0.f;
#pragma unroll
      for (int q = 0; q < 2 * PP; q++) {
        if (__builtin_expect(mk[q] != 0ull, 0)) {   // wave-uniform: hits are rare (a few per 1024 points; out of line)
          const int pos = cnt + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(mk[q] >> 32),
                                                                __builtin_amdgcn_mbcnt_lo((uint32_t)mk[q], 0u));
          const bool mine = hit[q];
          // (rounds 5a-5f routed the point's coordinates through an opaque copy inside the branch, because the compiler
          // speculated the row data of all 2 PP slots above the wave-uniform branches; with the hit blocks out of line
          // it no longer does, and the three copies per non-empty slot are gone)
          const float qx = vx[q], qy = vy[q], qz = vz[q];
          if (mine && pos < K) put_row(pos, roff, q * 64 + lane, qx - cx, qy - cy, qz - cz);
          if (cnt == 0) {
            const int fl = (int)__builtin_ctzll(mk[q]);
            first = q * 64 + fl;
            fdx = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(qx), fl)) - cx;
            fdy = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(qy), fl)) - cy;
            fdz = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(qz), fl)) - cz;
          }
          cnt += __popcll(mk[q]);
        }
      }
      if (cnt > K) cnt = K;
      if (cnt == 0) {   // (cannot happen for a centre that is a point of the cloud; kept for the generic contract)
        fdx = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(vx[0]), 0)) - cx;
        fdy = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(vy[0]), 0)) - cy;
        fdz = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(vz[0]), 0)) - cz;
      }
      const int nrow = cnt < 1 ? 2 : ((cnt + 1) & ~1);
      if (lane < nrow - cnt) put_row(lane, roff + cnt, first, fdx, fdy, fdz);
      roff += nrow;
      acc_c = pcr_writelane(cnt, jl, acc_c);
      if (jl == 63 || j == m - 1) {
        const int k = (j & ~63) + lane;
        if (k <= j) {
          idxs[k] = acc_i;
          new_xyz[3 * k] = __int_as_float(acc_x);
          new_xyz[3 * k + 1] = __int_as_float(acc_y);
          new_xyz[3 * k + 2] = __int_as_float(acc_z);
          cnt_out[k] = acc_c;
        }
      }
      if ((j & (cpw - 1)) == cpw - 1 || j == m - 1) {   // the item is complete: zero entries up to a whole 32-row block
        if (roff + lane < ((roff + 31) & ~31)) put_row(lane, roff, 0, 0.f, 0.f, 0.f);
        roff = ((j >> 4) + 1) * (cpw * K);   // (cpw K is a multiple of 32: K is even)
      }
    }
    if (j + 1 >= m) break;
    // ---- the next pick (fps_wave_kernel): the maximum of the running minima, the reference's tie rule
    const uint32_t best = dpp_max_u32(mx);
    const unsigned long long lm = __ballot(mx == best);
    bool unique = __popcll(lm) == 1;
    if (unique) {
      const int L = (int)__builtin_ctzll(lm);
      if constexpr (kVecPick) {
        const uint32_t sb = fps_slot_bits<2 * PP>(t, best, L);
        if (__popc(sb) == 1) old = 64 * (2 * PP - 1 - (int)__builtin_ctz(sb)) + L;
        else unique = false;
      } else {
        int hits = 0, slot_found = 0;
#pragma unroll
        for (int p = 0; p < 2 * PP; p++) {
          const unsigned long long mp = __ballot(t[p] == best);
          const int bit = (int)((mp >> L) & 1ull);
          hits += bit;
          slot_found = bit ? p : slot_found;
        }
        if (hits == 1) old = 64 * slot_found + L;
        else unique = false;
      }
    }
    if (!unique) {   // rare: the tie keys are derived here, not kept in registers (the opaque lane keeps them from
      // being hoisted out of the pick loop again)
      uint32_t lo = 0u;
      int lane_o = lane;
      asm volatile("" : "+v"(lane_o));
#pragma unroll
      for (int p = 0; p < 2 * PP; p++) {
        const int k = lane_o + 64 * p;
        const uint32_t tr = (uint32_t)k & (uint32_t)(block - 1);
        const uint32_t rev = logb ? (__brev(tr) >> (32 - logb)) : 0u;
        const uint32_t lowp = k < n ? ((((uint32_t)(block - 1) - rev) << 22) | (0x3FFFFFu - (uint32_t)k)) : 0u;
        const uint32_t l0 = t[p] == best ? lowp : 0u;
        lo = l0 > lo ? l0 : lo;
      }
      lo = dpp_max_u32(lo);
      old = (int)(0x3FFFFFu - (lo & 0x3FFFFFu));
    }
  }
}

int fps_launch(bool dist, const float *data, float *temp, int *idx, int B, int N, int M,
               hipStream_t st) {
  if (!data || !temp || !idx || B < 0 || N < 1 || N >= (1 << 22) || M < 0) return PCR_ERR_INVALID;
  if (B == 0 || M == 0) return PCR_OK;
  int logb = 0;
  while ((2 << logb) <= N && logb < 10) logb++;  // block = min(1024, 2^floor(log2 N))
  int block = 1 << logb;
  if (!dist && N <= 1024 && M > 1) {
    const size_t lds_wave = 0;
    dim3 gw(B), bw(64);
#define PCR_FPS_WAVE(PPv) hipLaunchKernelGGL((fps_wave_kernel<PPv>), gw, bw, lds_wave, st, data, temp, idx, N, M, block, logb)
    if (N <= 128) PCR_FPS_WAVE(1);
    else if (N <= 256) PCR_FPS_WAVE(2);
    else if (N <= 512) PCR_FPS_WAVE(4);
    else PCR_FPS_WAVE(8);
#undef PCR_FPS_WAVE
    PCR_CHECK_LAUNCH();
    return PCR_OK;
  }
  if (!dist && N <= 4096 && M > 1) {
    const size_t lds_fast = 64 + (size_t)3 * N * sizeof(float);
    dim3 gf(B);
#define PCR_FPS_FAST(P, T) hipLaunchKernelGGL((fps_fast_kernel<P, T>), gf, dim3(T), lds_fast, st, data, temp, idx, N, M, block, logb)
    if (N <= 256) PCR_FPS_FAST(1, 256);
    else if (N <= 512) PCR_FPS_FAST(2, 256);
    else if (N <= 1024) PCR_FPS_FAST(4, 256);
    else if (N <= 2048) PCR_FPS_FAST(8, 256);
    else PCR_FPS_FAST(16, 256);
#undef PCR_FPS_FAST
    PCR_CHECK_LAUNCH();
    return PCR_OK;
  }
  int threads = block < 64 ? 64 : block;
  bool reg = N <= kFpsPpt * block;
  int stage = (!dist && N <= kFpsLdsPts) ? 1 : 0;
  size_t lds = 256 + (stage ? (size_t)3 * N * sizeof(float) : 0);
  dim3 g(B), b(threads);
  if (dist) {
    if (reg) hipLaunchKernelGGL((fps_kernel<true, true>), g, b, lds, st, data, temp, idx, N, M, block, logb, stage, (const int *)nullptr, 0);
    else hipLaunchKernelGGL((fps_kernel<true, false>), g, b, lds, st, data, temp, idx, N, M, block, logb, stage, (const int *)nullptr, 0);
  } else {
    if (reg) hipLaunchKernelGGL((fps_kernel<false, true>), g, b, lds, st, data, temp, idx, N, M, block, logb, stage, (const int *)nullptr, 0);
    else hipLaunchKernelGGL((fps_kernel<false, false>), g, b, lds, st, data, temp, idx, N, M, block, logb, stage, (const int *)nullptr, 0);
  }
  PCR_CHECK_LAUNCH();
  return PCR_OK;
}

// ------------------------------------------------------------------ pairwise distances ----
// calc_square_dist (ops/furthest_point_sample/utils.py:4-32): (|a_i|^2 + |b_j|^2) - 2 <a_i, b_j> with every sum taken
// left to right over the channels and no fma (the order oracle/pcr_oracle.c:pcr_oracle_pairwise_sqdist restates).
// One thread per (i, j); the a row is a wave-uniform broadcast, b rows are C floats apart (C is 3 + a few features).
__global__ __launch_bounds__(256) void pairwise_sqdist_kernel(const float *__restrict__ a, const float *__restrict__ b,
                                                              float *__restrict__ out, int N, int M, int C, int norm) {
  const size_t bb = blockIdx.z;
  const int i = blockIdx.y;
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= M) return;
  const float *ai = a + (bb * N + i) * C;
  const float *bj = b + (bb * M + j) * C;
  float a2 = 0.f, b2 = 0.f, dot = 0.f;
  for (int c = 0; c < C; c++) {
    const float x = ai[c], y = bj[c];
    const float sa = x * x, sb = y * y, sd = x * y;
    a2 = a2 + sa;
    b2 = b2 + sb;
    dot = dot + sd;
  }
  const float t = a2 + b2;
  const float u = 2.0f * dot;
  float d = t - u;
  if (norm) d = sqrtf(d) / (float)C;
  out[(bb * N + i) * M + j] = d;
}

// ------------------------------------------------------------------------ ball query ----
// One thread per centre; the cloud streams through a 1024-point LDS tile that every lane reads
// at the same address (broadcast, conflict-free), instead of N uncoalesced AoS global loads per
// thread as in ball_query_cuda.cu:38-52.  Hits are written straight to the output row, the
// padding (first hit) once at the end.
constexpr int kBqTile = 1024;

// PY: the Python twin query_ball_point (models/pointnet2_utils.py:218-240): square_distance's expanded form
// (-2 <c,p> + |c|^2) + |p|^2 (:169-188; the dot product is summed left to right here, the reference leaves that order
// to its matmul), a point is kept unless d > r^2, and a row without any hit is filled with N as in the reference.
template <bool PY>
__global__ __launch_bounds__(256) void ball_query_kernel(const float *__restrict__ centres,
                                                         const float *__restrict__ xyz,
                                                         int *__restrict__ idx, int n, int m,
                                                         float min_r2, float max_r2, int K,
                                                         int *__restrict__ cnt_out) {
  __shared__ float tile[3 * kBqTile];
  const int tid = threadIdx.x;
  const size_t b = blockIdx.y;
  const int p = blockIdx.x * 256 + tid;
  const bool valid = p < m;
  float cx = 0.f, cy = 0.f, cz = 0.f;
  if (valid) {
    const float *c = centres + (b * m + p) * 3;
    cx = c[0]; cy = c[1]; cz = c[2];
  }
  int *out = idx + (b * m + (valid ? p : 0)) * K;
  const float *cloud = xyz + b * n * 3;
  int cnt = 0, first = PY ? n : 0;
  float cc = 0.f;
  if (PY) {
    const float a2 = cx * cx, b2 = cy * cy, c2 = cz * cz;
    const float s2 = a2 + b2;
    cc = s2 + c2;
  }
  bool done = !valid;
  for (int base = 0; base < n; base += kBqTile) {
    int tn = n - base < kBqTile ? n - base : kBqTile;
    __syncthreads();
    for (int i = tid; i < 3 * tn; i += 256) tile[i] = cloud[(size_t)base * 3 + i];
    __syncthreads();
    if (!done) {
      for (int k = 0; k < tn; k++) {
        bool hit;
        if (PY) {
          const float x = tile[3 * k], y = tile[3 * k + 1], z = tile[3 * k + 2];
          const float m0 = cx * x, m1 = cy * y, m2 = cz * z;
          const float dot0 = m0 + m1;
          const float dot = dot0 + m2;
          const float x2 = x * x, y2 = y * y, z2 = z * z;
          const float p0 = x2 + y2;
          const float pp = p0 + z2;
          const float t0 = -2.0f * dot;
          const float t1 = t0 + cc;
          const float d = t1 + pp;
          hit = !(d > max_r2);
        } else {
          const float d2 = pcr_sqdist3(tile[3 * k], tile[3 * k + 1], tile[3 * k + 2], cx, cy, cz);
          hit = d2 == 0.f || (d2 >= min_r2 && d2 < max_r2);
        }
        if (hit) {
          if (cnt == 0) first = base + k;
          out[cnt] = base + k;
          if (++cnt >= K) { done = true; break; }
        }
      }
    }
    if (__syncthreads_and(done ? 1 : 0)) break;
  }
  if (valid)
    for (int l = cnt; l < K; l++) out[l] = first;  // first == 0 when nothing was hit
  if (valid && cnt_out) cnt_out[b * m + p] = cnt;  // number of genuine hits; rows [cnt,K) are copies of row 0
}

// Register-resident variant for clouds of up to 1024 points: a WAVE owns a run of centres and keeps the
// whole cloud in registers (lane l holds points l, l+64, ...), so a centre costs PPL distance
// evaluations per lane instead of N serial LDS round trips per thread; hits are rare (a few per 1024
// points), so the ordered compaction (ballot + prefix popcount) sits behind a wave-uniform branch.
// Same hit predicate, same "first K in index order, pad with the first hit" result as above.
// kRows (round 4): the wave ALSO writes the compact row table of its 16 centres -- what the wave-autonomous ragged SA
// kernel otherwise rebuilds per item from cnt -> prefix -> idx -> xyz (three dependent loads deep): entries {neighbour index,
// dx, dy, dz} (point - centre), a centre's first rag rows = ceil2(max(cnt, 1)) back to back (odd counts and empty balls are
// padded with the row's first entry, as the index tensor pads), the run of the wave's centres (= one SA item) in its own
// 16 K-entry region, zero entries up to the next multiple of 32 rows.  idx may then be NULL (nobody reads it).
template <int PPL, bool kSimple, bool kRows = false>   // kSimple: min_radius = 0 < max_radius (every configuration in the reference)
__global__ __launch_bounds__(256) void ball_query_reg_kernel(const float *__restrict__ centres,
                                                             const float *__restrict__ xyz,
                                                             int *__restrict__ idx, int n, int m,
                                                             float min_r2, float max_r2, int K,
                                                             int *__restrict__ cnt_out, int cpw, int nchunk,
                                                             int nclouds, f32x4 *__restrict__ rows_out = nullptr) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // XCD-aware order: consecutive workgroup ids go round-robin over the eight XCDs (each with its own L2), so the
  // chunks of ONE cloud -- which all read that cloud's 12 KB -- are given ids that are congruent modulo 8: the cloud
  // is fetched from HBM once, not once per chunk (measured before: 424 MB fetched per launch for 50 MB of clouds)
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int chunk = slot % nchunk;
  const size_t b = (size_t)(slot / nchunk) * 8 + xcd;
  if (b >= (size_t)nclouds) return;
  const float *cloud = xyz + b * n * 3;
  // two points per register pair (j = 2 t, 2 t + 1): the distance arithmetic is packed f32 -- the same IEEE
  // subtract / multiply / add per component as pcr_sqdist3, two points per instruction
  static_assert(PPL % 2 == 0, "points per lane come in pairs");
  f32x2 px[PPL / 2], py[PPL / 2], pz[PPL / 2];
#pragma unroll
  for (int j = 0; j < PPL; j++) {
    const int i = j * 64 + lane;
    const bool ok = i < n;
    const float *q = cloud + (size_t)(ok ? i : 0) * 3;
    const float x = q[0], y = q[1], z = q[2];
    px[j >> 1][j & 1] = ok ? x : INFINITY;   // a point at infinity is never inside a ball
    py[j >> 1][j & 1] = ok ? y : INFINITY;
    pz[j >> 1][j & 1] = ok ? z : INFINITY;
  }
  const int c0 = (chunk * 4 + wave) * cpw;
  const int c1 = c0 + cpw < m ? c0 + cpw : m;
  // the wave's centres (cpw <= 21 of them: 3 cpw floats) in ONE register, a float per lane; a centre's coordinates
  // are three v_readlane instead of a dependent scalar load per centre
  float cv = 0.f;
  if (c0 < c1 && lane < 3 * (c1 - c0)) cv = centres[(b * m + c0) * 3 + lane];
  // K <= 64: a centre's row is assembled in a wave-private LDS strip and leaves as ONE store of K consecutive ints
  // (hits and padding together).  Written hit by hit, a row was several partial-line stores; the L2 fetched the lines
  // it did not own in full: 697 MB fetched per ssg1024 launch pair against 344 MB of algorithmic traffic.
  __shared__ int s_row[4][64];
  int *row = s_row[wave];
  const bool staged = K <= 64;
  // (kRows) this wave's region of the row table and its running row offset
  // (buffer stores: the region's base is a scalar, a lane contributes 16 x its row -- no 64-bit address arithmetic per
  // hit.  The scalar-offset field stays 0 ON PURPOSE: with a REGISTER there, this compiler schedules a write to the
  // store's data registers directly behind a 16-byte store -- it assumes the store-data hazard does not exist in that
  // form -- and gfx950 then stores the overwritten value now and then: ~100 wrong dx per 2 M rows, found the hard way)
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  f32x4 *rbase = kRows ? rows_out + ((size_t)b * ((m + cpw - 1) / cpw) + (size_t)(chunk * 4 + wave)) * (size_t)(cpw * K) : nullptr;
  const __amdgpu_buffer_rsrc_t rrows = __builtin_amdgcn_make_buffer_rsrc(rbase, 0, kRows ? cpw * K * 16 : 0, 0x00020000);
  auto put_row = [&](int slot, int soff, int i, float dx, float dy, float dz) __attribute__((always_inline)) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, f32x4{__int_as_float(i), dx, dy, dz}), rrows,
                                           (slot + soff) * 16, 0, 0);
  };
  int roff = 0;
  for (int c = c0; c < c1; c++) {
    const int co = 3 * (c - c0);
    const float cx = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, cv), co));
    const float cy = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, cv), co + 1));
    const float cz = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, cv), co + 2));
    int *gout = idx ? idx + (b * m + c) * (size_t)K : nullptr;
    int cnt = 0, first = 0;
    float fdx = 0.f, fdy = 0.f, fdz = 0.f;   // (kRows) point - centre of the first hit
    {
      // min_radius = 0: d2 >= 0 always holds and d2 == 0 implies d2 < max_r2, so the predicate is a single compare
      auto inside = [&](float d2) __attribute__((always_inline)) {
        return kSimple ? d2 < max_r2 : (bool)((d2 == 0.f) | ((d2 >= min_r2) & (d2 < max_r2)));
      };
      // pass 1, straight line: all PPL distances and their hit masks (scalar registers); hits are rare (a few per
      // 1024 points), so pass 2 -- the ordered compaction -- visits only the non-empty masks behind scalar tests
      const f32x2 x1 = {cx, cx}, y1 = {cy, cy}, z1 = {cz, cz};
      bool hit[PPL];   // (lane masks in scalar registers, like their ballots: "my point is inside" costs nothing below)
      unsigned long long mk[PPL];
#pragma unroll
      for (int t = 0; t < PPL / 2; t++) {
        const f32x2 dx = x1 - px[t], dy = y1 - py[t], dz = z1 - pz[t];   // pcr_sqdist3(p, centre): centre - p
        const f32x2 a2 = dx * dx;
        const f32x2 b2 = dy * dy;
        const f32x2 c2 = dz * dz;
        const f32x2 sab = a2 + b2;
        const f32x2 dd = sab + c2;
        hit[2 * t] = inside(dd[0]);
        hit[2 * t + 1] = inside(dd[1]);
        mk[2 * t] = __ballot(hit[2 * t]);
        mk[2 * t + 1] = __ballot(hit[2 * t + 1]);
      }
#pragma unroll
      for (int j = 0; j < PPL; j++) {
        if (mk[j] != 0ull) {   // wave-uniform, ONE scalar test per j (the scalar unit is shared by the CU's four SIMDs)
          const int pos = cnt + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(mk[j] >> 32),
                                                                __builtin_amdgcn_mbcnt_lo((uint32_t)mk[j], 0u));
          if (hit[j] && pos < K) {
            if (gout) {
              if (staged) row[pos] = j * 64 + lane;   // (kept apart: one pointer for both would be a flat store)
              else gout[pos] = j * 64 + lane;
            }
            if constexpr (kRows)
              put_row(pos, roff, j * 64 + lane, px[j >> 1][j & 1] - cx, py[j >> 1][j & 1] - cy, pz[j >> 1][j & 1] - cz);
          }
          if (cnt == 0) {
            const int fl = (int)__builtin_ctzll(mk[j]);
            first = j * 64 + fl;
            if constexpr (kRows) {
              fdx = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(px[j >> 1][j & 1]), fl)) - cx;
              fdy = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(py[j >> 1][j & 1]), fl)) - cy;
              fdz = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(pz[j >> 1][j & 1]), fl)) - cz;
            }
          }
          cnt += __popcll(mk[j]);
        }
      }
    }
    if (cnt > K) cnt = K;
    if constexpr (kRows) {
      if (cnt == 0) {   // nothing inside: the index tensor's row is all `first` = point 0
        fdx = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(px[0][0]), 0)) - cx;
        fdy = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(py[0][0]), 0)) - cy;
        fdz = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(pz[0][0]), 0)) - cz;
      }
      const int nrow = cnt < 1 ? 2 : ((cnt + 1) & ~1);
      if (lane < nrow - cnt) put_row(lane, roff + cnt, first, fdx, fdy, fdz);
      roff += nrow;
    }
    if (gout && staged) {
      // (a wave's LDS operations complete in order: the hit lanes' writes above are visible to the read below)
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      const int v = lane < cnt ? row[lane < K ? lane : 0] : first;   // first == 0 when nothing was hit
      if (lane < K) idx[(b * m + c) * (size_t)K + lane] = v;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
    } else if (gout) {
      for (int l = cnt + lane; l < K; l += 64) gout[l] = first;
    }
    if (cnt_out && lane == 0) cnt_out[b * m + c] = cnt;
  }
  if constexpr (kRows) {   // whole 32-row blocks: the consumer reads them without a clamp (and skips the padding's pairs)
    if (c0 < c1 && roff + lane < ((roff + 31) & ~31)) put_row(lane, roff, 0, 0.f, 0.f, 0.f);
  }
}

static void ball_query_launch(const float *centres, const float *xyz, int *idx, int *cnt, int B, int N, int M,
                              float min_r2, float max_r2, int K, hipStream_t st, float *rows = nullptr) {
  if (rows) {   // (validated by the caller: N <= 1024, min radius 0, K even)
    const int cpw = 16, nchunk = (M + 4 * cpw - 1) / (4 * cpw);
    const dim3 grid((unsigned)((B + 7) / 8) * 8 * nchunk), blk(256);
    f32x4 *r4 = reinterpret_cast<f32x4 *>(rows);
#define PCR_BQR(PPLv)                                                                                            \
  hipLaunchKernelGGL((ball_query_reg_kernel<PPLv, true, true>), grid, blk, 0, st, centres, xyz, idx, N, M, min_r2, \
                     max_r2, K, cnt, cpw, nchunk, B, r4)
    if (N <= 256) PCR_BQR(4);
    else if (N <= 512) PCR_BQR(8);
    else PCR_BQR(16);
#undef PCR_BQR
    return;
  }
  if (N <= 1024) {
    const int cpw = 16;                                  // centres per wave
    const int nchunk = (M + 4 * cpw - 1) / (4 * cpw);
    const dim3 grid((unsigned)((B + 7) / 8) * 8 * nchunk), blk(256);
    const bool simple = min_r2 == 0.f && max_r2 > 0.f;
#define PCR_BQ(PPLv, SIMPLEv)                                                                                    \
  hipLaunchKernelGGL((ball_query_reg_kernel<PPLv, SIMPLEv>), grid, blk, 0, st, centres, xyz, idx, N, M, min_r2, \
                     max_r2, K, cnt, cpw, nchunk, B)
    if (N <= 256) { if (simple) PCR_BQ(4, true); else PCR_BQ(4, false); }
    else if (N <= 512) { if (simple) PCR_BQ(8, true); else PCR_BQ(8, false); }
    else { if (simple) PCR_BQ(16, true); else PCR_BQ(16, false); }
#undef PCR_BQ
  } else {
    hipLaunchKernelGGL(ball_query_kernel<false>, dim3((M + 255) / 256, B), dim3(256), 0, st, centres, xyz, idx, N, M,
                       min_r2, max_r2, K, cnt);
  }
}

// --------------------------------------------------------------------------- heap kNN ----
// Same max-heap insertion and heap-sort sequence as knn_cuda.cu:27-94, so equal distances leave
// in the reference's order.  The per-thread heaps live in LDS as [slot][thread] (bank =
// thread, conflict-free whatever slot each lane touches) instead of 800 B of scratch per thread.
constexpr int kKnnTile = 256;

__global__ __launch_bounds__(64) void knn_heap_kernel(const float *__restrict__ xyz,
                                                      const float *__restrict__ centres,
                                                      int *__restrict__ idx,
                                                      float *__restrict__ dist2, int n, int m,
                                                      int K) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  float *hd = reinterpret_cast<float *>(smem_raw);
  int *hi = reinterpret_cast<int *>(smem_raw) + K * 64;
  float *tile = reinterpret_cast<float *>(smem_raw) + 2 * K * 64;
  const int tid = threadIdx.x;
  const size_t b = blockIdx.y;
  const int p = blockIdx.x * 64 + tid;
  const bool valid = p < m;
#define HD(i) hd[(i) * 64 + tid]
#define HI(i) hi[(i) * 64 + tid]
  float cx = 0.f, cy = 0.f, cz = 0.f;
  if (valid) {
    const float *c = centres + (b * m + p) * 3;
    cx = c[0]; cy = c[1]; cz = c[2];
  }
  for (int i = 0; i < K; i++) { HD(i) = 1e10f; HI(i) = 0; }
  const float *cloud = xyz + b * n * 3;

  auto reheap = [&](int k) {
    int root = 0, child = 1;
    while (child < k) {
      if (child + 1 < k && HD(child + 1) > HD(child)) child++;
      float dr = HD(root), dc = HD(child);
      if (dr > dc) return;
      int ir = HI(root), ic = HI(child);
      HD(root) = dc; HD(child) = dr;
      HI(root) = ic; HI(child) = ir;
      root = child;
      child = root * 2 + 1;
    }
  };

  for (int base = 0; base < n; base += kKnnTile) {
    int tn = n - base < kKnnTile ? n - base : kKnnTile;
    __syncthreads();
    for (int i = tid; i < 3 * tn; i += 64) tile[i] = cloud[(size_t)base * 3 + i];
    __syncthreads();
    if (valid) {
      for (int k = 0; k < tn; k++) {
        float d2 = pcr_sqdist3(tile[3 * k], tile[3 * k + 1], tile[3 * k + 2], cx, cy, cz);
        if (d2 < HD(0)) {
          HD(0) = d2;
          HI(0) = base + k;
          reheap(K);
        }
      }
    }
  }
  if (valid) {
    for (int i = K - 1; i > 0; i--) {
      float d0 = HD(0), di = HD(i);
      int i0 = HI(0), ii = HI(i);
      HD(0) = di; HD(i) = d0;
      HI(0) = ii; HI(i) = i0;
      reheap(i);
    }
    int *oi = idx + (b * m + p) * K;
    float *od = dist2 + (b * m + p) * K;
    for (int i = 0; i < K; i++) { oi[i] = HI(i); od[i] = HD(i); }
  }
#undef HD
#undef HI
}

// ------------------------------------------------------------------- gather / group ----
// Elementwise over the OUTPUT (coalesced stores along the last axis; the gathered reads hit
// one (b,c) row of at most 4N bytes that stays in L2).
__global__ void gather_fwd_kernel(const float *__restrict__ feat, const int *__restrict__ idx,
                                  float *__restrict__ out, int C, int N, int M, size_t total) {
  for (size_t e = blockIdx.x * (size_t)blockDim.x + threadIdx.x; e < total;
       e += (size_t)gridDim.x * blockDim.x) {
    size_t bc = e / M;
    int p = (int)(e - bc * M);
    size_t b = bc / C;
    out[e] = feat[bc * N + idx[b * M + p]];
  }
}

// Scatter-add for clouds too large for the owner-computes kernel below (N > 16384): float atomics as in
// gather_points_cuda.cu:69 / group_points_cuda.cu:30 (the summation order is unspecified there too).
__global__ void gather_bwd_kernel(const float *__restrict__ grad_out, const int *__restrict__ idx,
                                  float *__restrict__ grad_feat, int C, int N, int M,
                                  size_t total) {
  for (size_t e = blockIdx.x * (size_t)blockDim.x + threadIdx.x; e < total;
       e += (size_t)gridDim.x * blockDim.x) {
    size_t bc = e / M;
    int p = (int)(e - bc * M);
    size_t b = bc / C;
    atomicAdd(grad_feat + bc * N + idx[b * M + p], grad_out[e]);
  }
}

// Deterministic scatter-add (the backward of gather / group / three_interpolate): grad_feat[b][c][idx[b][e]] +=
// src[b][c][e / DIV] * w[b][e] for e < M.  One workgroup per (cloud, chunk of CS channels); the chunk's accumulators
// [CS][N] live in LDS and every (channel, point) accumulator has ONE owner thread -- (cl = tid % CS, part = tid / CS)
// owns the points congruent to `part` -- that adds its entries in increasing e: no atomics, the same bits every run.
template <int CS>
__global__ __launch_bounds__(256) void scatter_owner_kernel(const float *__restrict__ src, const int *__restrict__ idx,
                                                            const float *__restrict__ w, float *__restrict__ grad_feat,
                                                            int C, int N, int M, int DIV) {
  constexpr int PARTS = 256 / CS, TR = 64;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int NP = N | 1;
  float *acc = smem;                 // [CS][NP]
  float *st = acc + CS * NP;         // [CS][TR + 1]
  float *wt = st + CS * (TR + 1);    // [TR]
  int *it = reinterpret_cast<int *>(wt + TR);
  const int tid = threadIdx.x, cl = tid % CS, part = tid / CS;
  const size_t b = blockIdx.y;
  const int c0 = blockIdx.x * CS;
  const int Msrc = M / DIV;
  for (int e = tid; e < CS * NP; e += 256) acc[e] = 0.f;
  __syncthreads();
  for (int e0 = 0; e0 < M; e0 += TR) {
    const int nr = M - e0 < TR ? M - e0 : TR;
    if (tid < nr) {
      it[tid] = idx[b * M + e0 + tid];
      wt[tid] = w ? w[b * M + e0 + tid] : 1.0f;
    }
    for (int e = tid; e < CS * TR; e += 256) {
      const int c = e / TR, rr = e - c * TR;
      st[c * (TR + 1) + rr] = (rr < nr && c0 + c < C) ? src[(b * C + c0 + c) * Msrc + (e0 + rr) / DIV] : 0.f;
    }
    __syncthreads();
    if (c0 + cl < C) {
      const float *row = st + cl * (TR + 1);
      for (int rr = 0; rr < nr; rr++) {
        const int i = it[rr];
        // (an index outside [0, N) -- e.g. the Python-twin ball query's N sentinel for a row without hits -- has no
        // accumulator: it is skipped, as the reference's gather would have faulted on it)
        if ((unsigned)i < (unsigned)N && (i % PARTS) == part) acc[cl * NP + i] += row[rr] * wt[rr];
      }
    }
    __syncthreads();
  }
  for (int e = tid; e < CS * N; e += 256) {
    const int c = e / N, i = e - c * N;
    if (c0 + c < C) grad_feat[(b * C + c0 + c) * N + i] += acc[c * NP + i];
  }
}

// -> false when the cloud is too large for the LDS accumulators (caller falls back to the atomic kernel)
static bool scatter_owner_launch(const float *src, const int *idx, const float *w, float *grad_feat, int B, int C, int N,
                                 int M, int DIV, hipStream_t st) {
  int cs = 32;
  while (cs > 1 && (size_t)cs * (N | 1) * 4 > 64 * 1024) cs >>= 1;
  if ((size_t)cs * (N | 1) * 4 > 96 * 1024 || B > 65535) return false;
  if (cs > C) {
    cs = 1;
    while (cs * 2 <= C) cs *= 2;
  }
  const size_t lds = ((size_t)cs * (N | 1) + (size_t)cs * 65 + 64 + 64) * sizeof(float);
  const dim3 grid((C + cs - 1) / cs, B);
#define PCR_SC(CSv)                                                                                              \
  do {                                                                                                           \
    static bool ok = hipFuncSetAttribute(reinterpret_cast<const void *>(scatter_owner_kernel<CSv>),              \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess;  \
    (void)ok;                                                                                                    \
    hipLaunchKernelGGL(scatter_owner_kernel<CSv>, grid, dim3(256), lds, st, src, idx, w, grad_feat, C, N, M, DIV); \
  } while (0)
  switch (cs) {
    case 32: PCR_SC(32); break;
    case 16: PCR_SC(16); break;
    case 8: PCR_SC(8); break;
    case 4: PCR_SC(4); break;
    case 2: PCR_SC(2); break;
    default: PCR_SC(1); break;
  }
#undef PCR_SC
  return true;
}

int gather_launch(bool bwd, const float *a, const int *idx, float *o, int B, int C, int N, int M,
                  hipStream_t st) {
  if (!a || !idx || !o || B < 0 || C < 0 || N < 1 || M < 0) return PCR_ERR_INVALID;
  size_t total = (size_t)B * C * M;
  if (total == 0) return PCR_OK;
  size_t blocks = (total + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  if (bwd) {
    if (!scatter_owner_launch(a, idx, nullptr, o, B, C, N, M, 1, st))
      hipLaunchKernelGGL(gather_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, st, a, idx, o, C, N, M, total);
  } else hipLaunchKernelGGL(gather_fwd_kernel, dim3((unsigned)blocks), dim3(256), 0, st, a, idx, o, C, N, M, total);
  PCR_CHECK_LAUNCH();
  return PCR_OK;
}

// ------------------------------------------------------------------------- three NN ----
constexpr int kNnTile = 1024;

__global__ __launch_bounds__(256) void three_nn_kernel(const float *__restrict__ unknown,
                                                       const float *__restrict__ known,
                                                       float *__restrict__ dist2,
                                                       int *__restrict__ idx, int n, int m) {
  __shared__ float tile[3 * kNnTile];
  const int tid = threadIdx.x;
  const size_t b = blockIdx.y;
  const int p = blockIdx.x * 256 + tid;
  const bool valid = p < n;
  float ux = 0.f, uy = 0.f, uz = 0.f;
  if (valid) {
    const float *u = unknown + (b * n + p) * 3;
    ux = u[0]; uy = u[1]; uz = u[2];
  }
  double best1 = 1e40, best2 = 1e40, best3 = 1e40;
  int i1 = 0, i2 = 0, i3 = 0;
  const float *cloud = known + b * m * 3;
  for (int base = 0; base < m; base += kNnTile) {
    int tn = m - base < kNnTile ? m - base : kNnTile;
    __syncthreads();
    for (int i = tid; i < 3 * tn; i += 256) tile[i] = cloud[(size_t)base * 3 + i];
    __syncthreads();
    if (valid) {
      for (int k = 0; k < tn; k++) {
        float d = pcr_sqdist3(tile[3 * k], tile[3 * k + 1], tile[3 * k + 2], ux, uy, uz);
        if (d < best1) {
          best3 = best2; i3 = i2; best2 = best1; i2 = i1; best1 = d; i1 = base + k;
        } else if (d < best2) {
          best3 = best2; i3 = i2; best2 = d; i2 = base + k;
        } else if (d < best3) {
          best3 = d; i3 = base + k;
        }
      }
    }
  }
  if (valid) {
    float *o = dist2 + (b * n + p) * 3;
    int *oi = idx + (b * n + p) * 3;
    o[0] = (float)best1; o[1] = (float)best2; o[2] = (float)best3;
    oi[0] = i1; oi[1] = i2; oi[2] = i3;
  }
}

__global__ void three_interp_fwd_kernel(const float *__restrict__ feat,
                                        const int *__restrict__ idx,
                                        const float *__restrict__ w, float *__restrict__ out,
                                        int C, int M, int N, size_t total) {
  for (size_t e = blockIdx.x * (size_t)blockDim.x + threadIdx.x; e < total;
       e += (size_t)gridDim.x * blockDim.x) {
    size_t bc = e / N;
    int p = (int)(e - bc * N);
    size_t b = bc / C;
    const int *i = idx + (b * N + p) * 3;
    const float *ww = w + (b * N + p) * 3;
    const float *f = feat + bc * M;
    float a0 = ww[0] * f[i[0]];
    float a1 = ww[1] * f[i[1]];
    float a2 = ww[2] * f[i[2]];
    float s = a0 + a1;
    out[e] = s + a2;
  }
}

__global__ void three_interp_bwd_kernel(const float *__restrict__ grad_out,
                                        const int *__restrict__ idx,
                                        const float *__restrict__ w,
                                        float *__restrict__ grad_feat, int C, int N, int M,
                                        size_t total) {
  for (size_t e = blockIdx.x * (size_t)blockDim.x + threadIdx.x; e < total;
       e += (size_t)gridDim.x * blockDim.x) {
    size_t bc = e / N;
    int p = (int)(e - bc * N);
    size_t b = bc / C;
    const int *i = idx + (b * N + p) * 3;
    const float *ww = w + (b * N + p) * 3;
    float *g = grad_feat + bc * M;
    float go = grad_out[e];
    atomicAdd(g + i[0], go * ww[0]);
    atomicAdd(g + i[1], go * ww[1]);
    atomicAdd(g + i[2], go * ww[2]);
  }
}

// inclusive prefix sum over the 64 lanes (row shifts inside the 16-lane rows, then the two row broadcasts)
__device__ __forceinline__ int pcr_wave_incl_scan_i32(int x) {
  x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xF, 0xF, false);   // row_shr:1
  x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xF, 0xF, false);   // row_shr:2
  x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xF, 0xF, false);   // row_shr:4
  x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xF, 0xF, false);   // row_shr:8
  x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xA, 0xF, false);   // row_bcast:15 -> rows 1, 3
  x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xC, 0xF, false);   // row_bcast:31 -> rows 2, 3
  return x;
}

constexpr int kKnnPThreads = 256;
constexpr int kKnnCap = 256;  // candidates per query the fast path can rank (4 per lane)

// Ranks 0 .. K-1 of `total` <= kKnnCap candidates {index, distance bits} (= 64-bit (distance, index) keys) in the wave's
// LDS strip `cand` -> out[0 .. K), in (distance, index) order.  Shared by the register and the LDS kNN kernels.
// out_b / Kb (optional): the first Kb <= K ranks are ALSO written there (pcr_knn_prefix2_f32: the K-nearest list of a
// query is the prefix of its K2-nearest list, so one ranking serves two set-abstraction levels on the same cloud)
__device__ __forceinline__ void knn_emit_from_candidates(unsigned long long *cand, int total, int K, int lane, int *out,
                                                         int *out_b = nullptr, int Kb = 0) {
  uint32_t *cand32 = reinterpret_cast<uint32_t *>(cand);
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  if (total > 64 && total <= 128) {
    // 2b. a few candidates too many for one per lane (K = 48: 62-72 of them): the same bound once more, on the
    // candidates -- the minimum of a lane's two candidates, rank K-1 of those 64 (K distinct candidates are
    // <= it) -- and the survivors compacted in place (every lane holds its two before the first write).
    const uint32_t i0 = cand32[2 * lane], d0 = cand32[2 * lane + 1];
    const bool has1 = lane + 64 < total;
    const uint32_t i1 = cand32[2 * lane + 128], d1r = cand32[2 * lane + 129];
    const uint32_t d1 = has1 ? d1r : 0x7F7FFFFFu;
    const uint32_t dmr = d0 < d1 ? d0 : d1;
    const uint32_t dm = dmr < 0x7F7FFFFFu ? dmr : 0x7F7FFFFFu;   // (+inf / NaN bits: the tagged key must stay a finite float)
    const uint32_t ts2 =
        (uint32_t)__builtin_amdgcn_readlane((int)pcr_wave_sort_posf32((dm & ~63u) | (uint32_t)lane, lane), K - 1);
    const uint32_t tau2 = ts2 | 63u;
    __builtin_amdgcn_wave_barrier();
    const bool p0 = d0 <= tau2, p1 = has1 && d1 <= tau2;
    const unsigned long long m0 = __ballot(p0), m1 = __ballot(p1);
    const int s0 = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m0 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m0, 0u));
    const int n0 = __popcll(m0);
    const int s1 = n0 + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m1 >> 32),
                                                        __builtin_amdgcn_mbcnt_lo((uint32_t)m1, 0u));
    if (p0) {
      cand32[2 * s0] = i0;
      cand32[2 * s0 + 1] = d0;
    }
    if (p1) {
      cand32[2 * s1] = i1;
      cand32[2 * s1 + 1] = d1;
    }
    total = n0 + __popcll(m1);
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  }
  if (total <= 64) {
    // 3a. one candidate per lane.  Fast form: sort 32-bit keys (distance bits with the low 6 bits replaced by
    // the SLOT) with the float network; exact unless two of the first K + 1 ranks share their truncated
    // distance (~1 % of the queries), which is detected on the sorted keys and sent through the 64-bit sort.
    // (a distance that overflowed to +inf, or a NaN, would make the tagged key a NaN pattern, which the float-min
    // network mis-orders: clamped to the largest finite float -- such candidates then share their truncated key and
    // the tie detector below sends the query through the exact 64-bit path, which sorts the RAW bits)
    const uint32_t djr = cand32[2 * lane + 1];
    const uint32_t dj = djr < 0x7F7FFFFFu ? djr : 0x7F7FFFFFu;
    const uint32_t key = lane < total ? ((dj & ~63u) | (uint32_t)lane) : (0x7F7FFFC0u | (uint32_t)lane);
    const uint32_t sk = pcr_wave_sort_posf32(key, lane);
    const uint32_t nx = (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)sk, 0x130, 0xF, 0xF, false);   // lane + 1
    const bool tie = lane < K && ((sk ^ nx) < 64u);
    if (__ballot(tie) == 0ull) {
      const int v = (int)cand32[2 * (sk & 63u)];
      if (lane < K) out[lane] = v;
      if (out_b && lane < Kb) out_b[lane] = v;
    } else {
      const unsigned long long own = pcr_wave_sort_u64(lane < total ? cand[lane] : ~0ull, lane);
      if (lane < K) out[lane] = (int)(uint32_t)own;
      if (out_b && lane < Kb) out_b[lane] = (int)(uint32_t)own;
    }
  } else {
    // 3b. up to kKnnCap candidates: four per lane, broadcast LDS reads
    unsigned long long own[kKnnCap / 64];
    int rk[kKnnCap / 64];
#pragma unroll
    for (int u = 0; u < kKnnCap / 64; u++) {
      own[u] = lane + 64 * u < total ? cand[lane + 64 * u] : ~0ull;
      rk[u] = 0;
    }
    for (int j = 0; j < total; j++) {
      const unsigned long long cj = cand[j];
#pragma unroll
      for (int u = 0; u < kKnnCap / 64; u++) rk[u] += cj < own[u] ? 1 : 0;
    }
#pragma unroll
    for (int u = 0; u < kKnnCap / 64; u++)
      if (lane + 64 * u < total && rk[u] < K) {
        out[rk[u]] = (int)(own[u] & 0xFFFFFFFFull);
        if (out_b && rk[u] < Kb) out_b[rk[u]] = (int)(own[u] & 0xFFFFFFFFull);
      }
  }
}

// -------------------------------------------------------------- PT neighbour search ----
// knn_point(K, xyz, xyz[:, :S]) of the Point-Transformer (pointnet2_utils.py:205-216): for each of the first S
// points its K nearest points in (distance, index) order, distance = pcr_sqdist3.  One wave per query; selection
// without sorting N values:
//   1. tau = an upper bound of the K-th smallest distance from the 64 per-lane minima (64 distinct points);
//   2. every point with d <= tau (typically 1.3-1.5 K of them) becomes a packed (distance, index) key;
//   3. each candidate's rank among the candidates is counted; ranks < K are the answer, already ordered.
// Clouds of up to 1024 points: the wave keeps the whole cloud in REGISTERS as packed
// pairs (no LDS reads per query; packed f32 subtract / multiply / add = pcr_sqdist3's operations, two points per
// instruction), and every ranking step is a count over v_readlane broadcasts -- three instructions per
// comparand -- on keys whose order is the required one:
//   tau : lane minima as 32-bit keys (distance bits with the low 6 bits replaced by the lane: unique, and
//         monotone up to 64 ulps); the lane of rank K-1 gives tau = its distance with those 6 bits SET, still an
//         upper bound of the K-th smallest distance (K lanes have a minimum <= tau);
//   rank: the candidates d <= tau (a few more than K) are compacted one per lane and ranked by their exact
//         64-bit (distance, index) keys; ranks < K are the answer in (distance, index) order.
// More than 64 candidates (duplicate-heavy clouds) are ranked four per lane over broadcast LDS reads, more than
// kKnnCap go through a K-round argmin fallback; all paths produce the same output.
template <int TP>   // point PAIRS per lane: n <= 128 * TP
__global__ __launch_bounds__(kKnnPThreads) void knn_prefix_reg_kernel(const float *__restrict__ xyz,
                                                                  int *__restrict__ idx, int n, int S,
                                                                  int K0, int qpw, int *__restrict__ idx2, int S2, int K2) {
  constexpr int T = 2 * TP;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float *sx = smem, *sy = smem + n, *sz = smem + 2 * n;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // candidates {index, distance bits} (= a 64-bit (distance, index) key); a round of the compaction adds at most 64
  constexpr int CAPW = kKnnCap + 64;
  unsigned long long *cand = reinterpret_cast<unsigned long long *>(smem + 3 * n + (n & 1)) +
                             __builtin_amdgcn_readfirstlane(wave) * CAPW;   // (a scalar: slot addresses fold it)
  uint32_t *cand32 = reinterpret_cast<uint32_t *>(cand);
  const size_t b = blockIdx.y;
  const float *cloud = xyz + b * n * 3;
  for (int i = tid; i < n; i += kKnnPThreads) {
    sx[i] = cloud[3 * i];
    sy[i] = cloud[3 * i + 1];
    sz[i] = cloud[3 * i + 2];
  }
  __syncthreads();
  f32x2 px[TP], py[TP], pz[TP];
#pragma unroll
  for (int t = 0; t < T; t++) {
    const int i = lane + 64 * t;
    const bool ok = i < n;
    px[t >> 1][t & 1] = ok ? sx[i] : INFINITY;   // a point at infinity is never among the K nearest (K <= n)
    py[t >> 1][t & 1] = ok ? sy[i] : INFINITY;
    pz[t >> 1][t & 1] = ok ? sz[i] : INFINITY;
  }
  const int q0 = blockIdx.x * qpw;
  const int q1 = (q0 + qpw < S) ? q0 + qpw : S;
  for (int q = q0 + wave; q < q1; q += kKnnPThreads / 64) {
    // (pcr_knn_prefix2_f32: the first S2 queries rank K2 >= K0 neighbours and write both lists; wave-uniform)
    const bool two = q < S2;
    const int K = two ? K2 : K0;
    const float qx = sx[q], qy = sy[q], qz = sz[q];
    const f32x2 x1 = {qx, qx}, y1 = {qy, qy}, z1 = {qz, qz};
    uint32_t d[T];   // distances as BITS (>= +0: unsigned order = float order; +inf = 0x7f800000 sorts last)
    uint32_t m = 0xFFFFFFFFu;
#pragma unroll
    for (int t = 0; t < TP; t++) {
      const f32x2 dx = px[t] - x1, dy = py[t] - y1, dz = pz[t] - z1;   // pcr_sqdist3(q, p): p - q
      const f32x2 a = dx * dx;
      const f32x2 bb = dy * dy;
      const f32x2 c = dz * dz;
      const f32x2 sab = a + bb;
      const f32x2 dd = sab + c;
      d[2 * t] = __float_as_uint(dd[0]);
      d[2 * t + 1] = __float_as_uint(dd[1]);
      const uint32_t m01 = d[2 * t] < d[2 * t + 1] ? d[2 * t] : d[2 * t + 1];
      m = m01 < m ? m01 : m;
    }
    // 1. the (truncated, lane-tagged) lane minimum of rank K-1: sort the 64 keys across the lanes (21-step network
    // on float bits, pcr_common.h; a lane without a valid point holds +inf: clamped to the largest finite float so
    // that the tagged key is not a NaN pattern)
    const uint32_t mc = m < 0x7F7FFFFFu ? m : 0x7F7FFFFFu;
    const uint32_t mkey = (mc & ~63u) | (uint32_t)lane;
    const uint32_t ts = (uint32_t)__builtin_amdgcn_readlane((int)pcr_wave_sort_posf32(mkey, lane), K - 1);
    // (rank K-1 at the clamp = distances overflowed: let every point pass, the rounds below rank them)
    const uint32_t tau = ts >= 0x7F7FFFC0u ? 0x7F800000u : (ts | 63u);
    // 2. candidates d <= tau, compacted in (t, lane) order: the slot of a candidate is the number of candidates
    // before it -- the running total of the earlier t (scalar) + the passing lanes below it (mbcnt of the ballot)
    int total = 0;
    if constexpr (T >= 16) {
      // (round 5, clouds of 513-1024 points: the lane-local form of knn_prefix_lds_kernel's pass 2 -- one compare word per
      // lane, a wave scan of the counts, the indices written by a short divergent loop and the candidates' distances
      // recomputed one per lane from the staged cloud: ~100 issue slots instead of 12 per register = 192)
      static_assert(T == 16, "one 16-bit compare word per lane");
      uint32_t w = 0u;
#define PCR_KW1(a) "v_cmp_ge_u32 vcc, %[tau], %[" #a "]\n\tv_addc_co_u32 %[w], vcc, %[w], %[w], vcc\n\t"
#pragma unroll
      for (int g = 1; g >= 0; g--) {   // (registers in descending order: the last one shifted in is bit 0)
        const int t0 = 8 * g;
        asm(PCR_KW1(t7) PCR_KW1(t6) PCR_KW1(t5) PCR_KW1(t4) PCR_KW1(t3) PCR_KW1(t2) PCR_KW1(t1) PCR_KW1(t0)
            : [w] "+v"(w)
            : [tau] "s"(tau), [t0] "v"(d[t0]), [t1] "v"(d[t0 + 1]), [t2] "v"(d[t0 + 2]), [t3] "v"(d[t0 + 3]),
              [t4] "v"(d[t0 + 4]), [t5] "v"(d[t0 + 5]), [t6] "v"(d[t0 + 6]), [t7] "v"(d[t0 + 7])
            : "vcc");
      }
#undef PCR_KW1
      const int mycnt = __popc(w);
      const int incl = pcr_wave_incl_scan_i32(mycnt);
      total = __builtin_amdgcn_readlane(incl, 63);
      if (total <= kKnnCap) {
        int pos = incl - mycnt;
        while (w) {
          const int t = (int)__builtin_ctz(w);
          w &= w - 1u;
          cand32[2 * pos] = (uint32_t)(lane + 64 * t);
          pos++;
        }
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        for (int c = lane; c < total; c += 64) {
          const int i = (int)cand32[2 * c];
          const bool ok = i < n;                       // (a lane's padding points sit at infinity: distance +inf)
          const int ii = ok ? i : 0;
          const float dx = sx[ii] - qx, dy = sy[ii] - qy, dz = sz[ii] - qz;
          const float a = dx * dx;
          const float bb = dy * dy;
          const float cc = dz * dz;
          const float sab = a + bb;
          cand32[2 * c + 1] = ok ? __float_as_uint(sab + cc) : 0x7F800000u;
        }
      }
    } else {
#pragma unroll
    for (int t = 0; t < T; t++) {
      const bool pass = d[t] <= tau;
      const unsigned long long mask = __ballot(pass);
      const int pos = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32),
                                                     __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
      uint32_t *wp = cand32 + 2 * total;   // (scalar)
      if (total <= kKnnCap && pass) {   // (scalar test: past kKnnCap the candidates are not used, only counted)
        wp[2 * pos] = (uint32_t)(lane + 64 * t);
        wp[2 * pos + 1] = d[t];
      }
      total += __popcll(mask);
    }
    }
    int *out0 = idx + (b * S + q) * K0;
    int *out = two ? idx2 + (b * S2 + q) * K2 : out0;   // the list of K ranks; out_b: its first K0 (two lists only)
    int *out_b = two ? out0 : nullptr;
    if (total <= kKnnCap) {
      knn_emit_from_candidates(cand, total, K, lane, out, out_b, K0);
      __builtin_amdgcn_wave_barrier();
    } else {
      int mine = 0;
      for (int k = 0; k < K; k++) {
        uint32_t bd = d[0];
        int bt = 0;
#pragma unroll
        for (int t = 1; t < T; t++)
          if (d[t] < bd) { bd = d[t]; bt = t; }
        unsigned long long key = ((unsigned long long)(bd | 0x80000000u) << 32) | (unsigned)(lane + 64 * bt);
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
      if (out_b && lane < K0) out_b[lane] = mine;
    }
  }
}

// Larger clouds (1024 < n <= 4096): 32-64 points per lane do not fit in registers next to their distances, so the
// cloud sits in LDS (as the point PAIRS a lane evaluates with packed arithmetic) and only the T distances of a query
// live in registers: pass 1 computes them and the lane minimum (-> tau as above), pass 2 compacts the candidates
// d <= tau chunk by chunk (ballot + prefix popcount, in index order).  (Rounds 2-3 recomputed the distances in pass 2
// from a second sweep of the LDS copy: 128 KB of LDS reads per query, the kernel's bound; pt4096's 4096 x 4096 launch
// 5.9 -> 3.8 ms with the distances kept; the packed pairs changed nothing measurable)  Ranking as in the register kernel; if more
// than kKnnCap points pass (heavily duplicated clouds) the K nearest are emitted one per round as "the smallest
// (distance, index) key above the previous one", which needs no per-point state either.
template <int T, int NT>   // points per lane: n <= 64 * T; NT threads (NT / 64 waves share the cloud in LDS)
__global__ __launch_bounds__(NT) void knn_prefix_lds_kernel(const float *__restrict__ xyz,
                                                                  int *__restrict__ idx, int n, int S,
                                                                  int K0, int qpw, int *__restrict__ idx2, int S2, int K2) {
  static_assert(T % 2 == 0, "points per lane come in pairs");
  constexpr int TP = T / 2;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  // the cloud in LDS as the PAIRS a lane evaluates together (points lane + 64 (2 tp) and lane + 64 (2 tp + 1)):
  // {x0, x1, y0, y1} + {z0, z1} -- register pairs for packed f32 arithmetic (the same IEEE operations as pcr_sqdist3,
  // two points per instruction), 24 bytes per pair instead of two 16-byte points; padded with points at infinity
  f32x4 *sP = reinterpret_cast<f32x4 *>(smem);            // [TP][64]
  f32x2 *sZ = reinterpret_cast<f32x2 *>(sP + 64 * TP);    // [TP][64]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  unsigned long long *cand = reinterpret_cast<unsigned long long *>(sZ + 64 * TP) + wave * kKnnCap;
  const size_t b = blockIdx.y;
  const float *cloud = xyz + b * n * 3;
  for (int i = tid; i < 64 * T; i += NT) {
    const bool ok = i < n;
    const float *q = cloud + (size_t)(ok ? i : 0) * 3;
    const float x = ok ? q[0] : INFINITY, y = ok ? q[1] : INFINITY, z = ok ? q[2] : INFINITY;
    const int t = i >> 6, l = i & 63, unit = (t >> 1) * 64 + l, sl = t & 1;
    reinterpret_cast<float *>(sP + unit)[sl] = x;
    reinterpret_cast<float *>(sP + unit)[2 + sl] = y;
    reinterpret_cast<float *>(sZ + unit)[sl] = z;
  }
  __syncthreads();
  const unsigned long long lt = (1ull << lane) - 1ull;
  const int q0 = blockIdx.x * qpw;
  const int q1 = (q0 + qpw < S) ? q0 + qpw : S;
  for (int q = q0 + wave; q < q1; q += NT / 64) {
    const bool two = q < S2;            // (pcr_knn_prefix2_f32, as in the register kernel)
    const int K = two ? K2 : K0;
    const int qu = ((q >> 7) * 64) + (q & 63), qs = (q >> 6) & 1;
    const float qx = reinterpret_cast<const float *>(sP + qu)[qs], qy = reinterpret_cast<const float *>(sP + qu)[2 + qs],
                qz = reinterpret_cast<const float *>(sZ + qu)[qs];
    const f32x2 x1 = {qx, qx}, y1 = {qy, qy}, z1 = {qz, qz};
    auto dist_pair = [&](int tp) {   // pcr_sqdist3(query, point) of the pair's two points (bits >= +0; +inf sorts last)
      const f32x4 pp = sP[tp * 64 + lane];
      const f32x2 pz = sZ[tp * 64 + lane];
      const f32x2 dx = f32x2{pp[0], pp[1]} - x1, dy = f32x2{pp[2], pp[3]} - y1, dz = pz - z1;   // (p - q, as pcr_sqdist3)
      const f32x2 a = dx * dx;
      const f32x2 bb = dy * dy;
      const f32x2 c = dz * dz;
      const f32x2 sab = a + bb;
      return sab + c;
    };
    auto dist_bits = [&](int t) {
      const f32x2 dd = dist_pair(t >> 1);
      return __float_as_uint((t & 1) ? dd[1] : dd[0]);
    };
    // (round 4: the T distances of pass 1 STAY in registers -- 64 VGPRs at T = 64, inside the 128 of four waves per SIMD --
    // instead of being recomputed from a second 64 KB sweep of the LDS copy: the kernel was bound by LDS bandwidth, 128 KB
    // per query; the rare overflow path below still recomputes)
    uint32_t dk[T];
    uint32_t m = 0xFFFFFFFFu;
#pragma unroll
    for (int tp = 0; tp < TP; tp++) {
      if ((tp & 3) == 0) __builtin_amdgcn_sched_barrier(0);   // (eight points in flight: all T reads at once would spill)
      const f32x2 dd = dist_pair(tp);
      dk[2 * tp] = __float_as_uint(dd[0]);
      dk[2 * tp + 1] = __float_as_uint(dd[1]);
      const uint32_t m01 = dk[2 * tp] < dk[2 * tp + 1] ? dk[2 * tp] : dk[2 * tp + 1];
      m = m01 < m ? m01 : m;
    }
    __builtin_amdgcn_sched_barrier(0);
    // (the float-bits network of the register kernel: two VALU instructions per step instead of the integer network's four;
    // a lane whose points are all padding holds +inf: clamped so that the tagged key is not a NaN pattern)
    const uint32_t mc = m < 0x7F7FFFFFu ? m : 0x7F7FFFFFu;
    const uint32_t mkey = (mc & ~63u) | (uint32_t)lane;
    const uint32_t ts = (uint32_t)__builtin_amdgcn_readlane((int)pcr_wave_sort_posf32(mkey, lane), K - 1);
    const uint32_t tau = ts >= 0x7F7FFFC0u ? 0x7F800000u : (ts | 63u);
    // ---- pass 2 (round 5): the candidates d <= tau WITHOUT a ballot per register.  Every lane shifts one compare per
    // register into private words (v_cmp + v_addc_co: w = 2 w + carry; bit t % 32 of word t / 32 = "my point t passed"),
    // a wave scan of the lanes' counts gives every lane the first slot of its own candidates, and a short divergent
    // loop over the set bits (about one per lane) writes their INDICES; the distances are then recomputed one candidate
    // per lane from the LDS copy -- the same operations on the same operands, so the same bits.  (Before: a compare, a
    // wave-uniform branch and, for the half of the registers with a hit, mbcnt + masked 8-byte store + scalar
    // bookkeeping: ~750 of the query's ~1250 issue slots at T = 64.  The ranking does not depend on the candidates'
    // order in the strip.)
    constexpr int NW = T / 32;
    uint32_t w[NW];
#pragma unroll
    for (int u = 0; u < NW; u++) w[u] = 0u;
#define PCR_KW1(a) "v_cmp_ge_u32 vcc, %[tau], %[" #a "]\n\tv_addc_co_u32 %[w], vcc, %[w], %[w], vcc\n\t"
#pragma unroll
    for (int u = 0; u < NW; u++)
#pragma unroll
      for (int g = 3; g >= 0; g--) {   // (registers in descending order: the last one shifted in is bit 0)
        const int t0 = 32 * u + 8 * g;
        asm(PCR_KW1(t7) PCR_KW1(t6) PCR_KW1(t5) PCR_KW1(t4) PCR_KW1(t3) PCR_KW1(t2) PCR_KW1(t1) PCR_KW1(t0)
            : [w] "+v"(w[u])
            : [tau] "s"(tau), [t0] "v"(dk[t0]), [t1] "v"(dk[t0 + 1]), [t2] "v"(dk[t0 + 2]), [t3] "v"(dk[t0 + 3]),
              [t4] "v"(dk[t0 + 4]), [t5] "v"(dk[t0 + 5]), [t6] "v"(dk[t0 + 6]), [t7] "v"(dk[t0 + 7])
            : "vcc");
      }
#undef PCR_KW1
    int mycnt = 0;
#pragma unroll
    for (int u = 0; u < NW; u++) mycnt += __popc(w[u]);
    const int incl = pcr_wave_incl_scan_i32(mycnt);
    const int total = __builtin_amdgcn_readlane(incl, 63);
    if (total <= kKnnCap) {
      uint32_t *cand32 = reinterpret_cast<uint32_t *>(cand);
      int pos = incl - mycnt;
#pragma unroll
      for (int u = 0; u < NW; u++) {
        uint32_t ww = w[u];
        while (ww) {
          const int t = 32 * u + (int)__builtin_ctz(ww);
          ww &= ww - 1u;
          cand32[2 * pos] = (uint32_t)(lane + 64 * t);
          pos++;
        }
      }
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      const float *sPf = reinterpret_cast<const float *>(sP), *sZf = reinterpret_cast<const float *>(sZ);
      for (int c = lane; c < total; c += 64) {
        const int i = (int)cand32[2 * c];
        const int t = i >> 6, unit = (t >> 1) * 64 + (i & 63), sl = t & 1;
        const float dx = sPf[4 * unit + sl] - qx, dy = sPf[4 * unit + 2 + sl] - qy, dz = sZf[2 * unit + sl] - qz;
        const float a = dx * dx;
        const float bb = dy * dy;
        const float cc = dz * dz;
        const float sab = a + bb;
        cand32[2 * c + 1] = __float_as_uint(sab + cc);
      }
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    int *out0 = idx + (b * S + q) * K0;
    int *out = two ? idx2 + (b * S2 + q) * K2 : out0;
    int *out_b = two ? out0 : nullptr;
    if (total <= kKnnCap) {
      knn_emit_from_candidates(cand, total, K, lane, out, out_b, K0);
    } else {
      unsigned long long last = 0ull;   // every real key has bit 63 set
      for (int k = 0; k < K; k++) {
        unsigned long long best = ~0ull;
        for (int t = 0; t < T; t++) {
          const unsigned long long key = ((unsigned long long)(dist_bits(t) | 0x80000000u) << 32) | (unsigned)(lane + 64 * t);
          best = (key > last && key < best) ? key : best;
        }
        best = pcr_wave_min_u64(best);
        if (lane == 0) {
          out[k] = (int)(best & 0xFFFFFFFFull);
          if (out_b && k < K0) out_b[k] = (int)(best & 0xFFFFFFFFull);
        }
        last = best;
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
}

}  // namespace

// ------------------------------------------------------------------------------ C ABI ----
PCR_EXPORT int pcr_abi_version(void) { return 17; }

PCR_EXPORT const char *pcr_status_string(int status) {
  switch (status) {
    case PCR_OK: return "ok";
    case PCR_ERR_INVALID: return "invalid argument or unsupported configuration";
    case PCR_ERR_LAUNCH: return "kernel launch failed";
    default: return "unknown status";
  }
}

PCR_EXPORT int pcr_fps_f32(const float *xyz, float *temp, int *idx, int B, int N, int M,
                           pcr_stream_t stream) {
  return fps_launch(false, xyz, temp, idx, B, N, M, pcr_s(stream));
}

PCR_EXPORT int pcr_fps_dist_f32(const float *dist, float *temp, int *idx, int B, int N, int M,
                                pcr_stream_t stream) {
  return fps_launch(true, dist, temp, idx, B, N, M, pcr_s(stream));
}

PCR_EXPORT int pcr_pairwise_sqdist_f32(const float *a, const float *b, float *out, int B, int N, int M, int C, int norm,
                                       pcr_stream_t stream) {
  if (!a || !b || !out || B < 0 || N < 0 || M < 0 || C < 1) return PCR_ERR_INVALID;
  if (B == 0 || N == 0 || M == 0) return PCR_OK;
  if (B > 65535 || N > 65535) return PCR_ERR_INVALID;
  hipLaunchKernelGGL(pairwise_sqdist_kernel, dim3((M + 255) / 256, N, B), dim3(256), 0, pcr_s(stream), a, b, out, N, M, C,
                     norm);
  PCR_CHECK_LAUNCH();
  return PCR_OK;
}

PCR_EXPORT int pcr_fps_py_f32(const float *xyz, float *temp, const int *start, int *idx, int B, int N, int M,
                              pcr_stream_t stream) {
  if (!xyz || !temp || !idx || B < 0 || N < 1 || N >= (1 << 22) || M < 0) return PCR_ERR_INVALID;
  if (B == 0 || M == 0) return PCR_OK;
  int logb = 0;
  while ((2 << logb) <= N && logb < 10) logb++;
  const int block = 1 << logb;
  const int threads = block < 64 ? 64 : block;
  const bool reg = N <= kFpsPpt * block;
  const int stage = N <= kFpsLdsPts ? 1 : 0;
  const size_t lds = 256 + (stage ? (size_t)3 * N * sizeof(float) : 0);
  if (reg) hipLaunchKernelGGL((fps_kernel<false, true>), dim3(B), dim3(threads), lds, pcr_s(stream), xyz, temp, idx, N, M,
                              block, logb, stage, start, 1);
  else hipLaunchKernelGGL((fps_kernel<false, false>), dim3(B), dim3(threads), lds, pcr_s(stream), xyz, temp, idx, N, M,
                          block, logb, stage, start, 1);
  PCR_CHECK_LAUNCH();
  return PCR_OK;
}

PCR_EXPORT int pcr_query_ball_point_f32(const float *centres, const float *xyz, int *idx, int B, int N, int M,
                                        float radius, int K, pcr_stream_t stream) {
  if (!centres || !xyz || !idx || B < 0 || N < 1 || M < 0 || K < 1) return PCR_ERR_INVALID;
  if (B == 0 || M == 0) return PCR_OK;
  if (B > 65535) return PCR_ERR_INVALID;
  hipLaunchKernelGGL(ball_query_kernel<true>, dim3((M + 255) / 256, B), dim3(256), 0, pcr_s(stream), centres, xyz, idx,
                     N, M, 0.f, radius * radius, K, (int *)nullptr);
  PCR_CHECK_LAUNCH();
  return PCR_OK;
}

PCR_EXPORT int pcr_ball_query_f32(const float *centres, const float *xyz, int *idx, int B, int N,
                                  int M, float min_r, float max_r, int K, pcr_stream_t stream) {
  if (!centres || !xyz || !idx || B < 0 || N < 1 || M < 0 || K < 1) return PCR_ERR_INVALID;
  if (B == 0 || M == 0) return PCR_OK;
  if (B > 65535) return PCR_ERR_INVALID;
  float max_r2 = max_r * max_r, min_r2 = min_r * min_r;
  ball_query_launch(centres, xyz, idx, nullptr, B, N, M, min_r2, max_r2, K, pcr_s(stream));
  PCR_CHECK_LAUNCH();
  return PCR_OK;
}

PCR_EXPORT int pcr_ball_query_cnt_f32(const float *centres, const float *xyz, int *idx, int *cnt, int B, int N,
                                      int M, float min_r, float max_r, int K, pcr_stream_t stream) {
  if (!centres || !xyz || !idx || !cnt || B < 0 || N < 1 || M < 0 || K < 1) return PCR_ERR_INVALID;
  if (B == 0 || M == 0) return PCR_OK;
  if (B > 65535) return PCR_ERR_INVALID;
  float max_r2 = max_r * max_r, min_r2 = min_r * min_r;
  ball_query_launch(centres, xyz, idx, cnt, B, N, M, min_r2, max_r2, K, pcr_s(stream));
  PCR_CHECK_LAUNCH();
  return PCR_OK;
}

PCR_EXPORT long pcr_ball_query_rows_floats(int B, int M, int K) {
  if (B < 1 || M < 1 || K < 1) return 0;
  return (long)B * ((M + 15) / 16) * 16 * K * 4;
}

PCR_EXPORT int pcr_ball_query_rows_ok(int N, int K, float min_r) { return N >= 1 && N <= 1024 && K >= 2 && !(K & 1) && min_r == 0.f; }

PCR_EXPORT int pcr_ball_query_rows_f32(const float *centres, const float *xyz, int *idx, int *cnt, float *rows, int B,
                                       int N, int M, float min_r, float max_r, int K, pcr_stream_t stream) {
  if (!centres || !xyz || !cnt || !rows || B < 0 || N < 1 || M < 0 || K < 1 || !(max_r > 0.f) ||
      !pcr_ball_query_rows_ok(N, K, min_r))
    return PCR_ERR_INVALID;
  if (B == 0 || M == 0) return PCR_OK;
  if (B > 65535) return PCR_ERR_INVALID;
  ball_query_launch(centres, xyz, idx, cnt, B, N, M, 0.f, max_r * max_r, K, pcr_s(stream), rows);
  PCR_CHECK_LAUNCH();
  return PCR_OK;
}

PCR_EXPORT int pcr_fps_ball_query_rows_ok(int N, int M, int K) {
  return N >= 1 && N <= 1024 && M >= 2 && M <= N && K >= 2 && K <= 64 * 16 && !(K & 1);
}

PCR_EXPORT int pcr_fps_ball_query_rows_f32(const float *xyz, float *temp, int *idx, float *new_xyz, int *cnt, float *rows,
                                           int B, int N, int M, float max_r, int K, pcr_stream_t stream) {
  if (!xyz || !temp || !idx || !new_xyz || !cnt || !rows || B < 0 || !(max_r > 0.f) || !pcr_fps_ball_query_rows_ok(N, M, K))
    return PCR_ERR_INVALID;
  if (B == 0) return PCR_OK;
  int logb = 0;
  while ((2 << logb) <= N && logb < 10) logb++;   // block = min(1024, 2^floor(log2 N)): fps_launch's tie-order block
  const int block = 1 << logb;
  const float max_r2 = max_r * max_r;
  f32x4 *r4 = reinterpret_cast<f32x4 *>(rows);
  hipStream_t st = pcr_s(stream);
#define PCR_FPS_BQ(PPv) \
  hipLaunchKernelGGL((fps_bq_wave_kernel<PPv>), dim3(B), dim3(64), 0, st, xyz, temp, idx, new_xyz, cnt, r4, N, M, block, logb, max_r2, K)
  if (N <= 128) PCR_FPS_BQ(1);
  else if (N <= 256) PCR_FPS_BQ(2);
  else if (N <= 512) PCR_FPS_BQ(4);
  else PCR_FPS_BQ(8);
#undef PCR_FPS_BQ
  PCR_CHECK_LAUNCH();
  return PCR_OK;
}

PCR_EXPORT int pcr_knn_f32(const float *xyz, const float *centres, int *idx, float *dist2, int B,
                           int N, int M, int K, pcr_stream_t stream) {
  if (!xyz || !centres || !idx || !dist2 || B < 0 || N < 1 || M < 0 || K < 1 || K > 100)
    return PCR_ERR_INVALID;
  if (B == 0 || M == 0) return PCR_OK;
  if (B > 65535) return PCR_ERR_INVALID;
  size_t lds = (size_t)2 * K * 64 * 4 + 3 * kKnnTile * 4;
  hipLaunchKernelGGL(knn_heap_kernel, dim3((M + 63) / 64, B), dim3(64), lds, pcr_s(stream), xyz,
                     centres, idx, dist2, N, M, K);
  PCR_CHECK_LAUNCH();
  return PCR_OK;
}

PCR_EXPORT int pcr_gather_fwd_f32(const float *feat, const int *idx, float *out, int B, int C,
                                  int N, int M, pcr_stream_t stream) {
  return gather_launch(false, feat, idx, out, B, C, N, M, pcr_s(stream));
}

PCR_EXPORT int pcr_gather_bwd_f32(const float *grad_out, const int *idx, float *grad_feat, int B,
                                  int C, int N, int M, pcr_stream_t stream) {
  return gather_launch(true, grad_out, idx, grad_feat, B, C, N, M, pcr_s(stream));
}

PCR_EXPORT int pcr_group_fwd_f32(const float *feat, const int *idx, float *out, int B, int C,
                                 int N, int S, int K, pcr_stream_t stream) {
  if (S < 0 || K < 0) return PCR_ERR_INVALID;
  return gather_launch(false, feat, idx, out, B, C, N, S * K, pcr_s(stream));
}

PCR_EXPORT int pcr_group_bwd_f32(const float *grad_out, const int *idx, float *grad_feat, int B,
                                 int C, int N, int S, int K, pcr_stream_t stream) {
  if (S < 0 || K < 0) return PCR_ERR_INVALID;
  return gather_launch(true, grad_out, idx, grad_feat, B, C, N, S * K, pcr_s(stream));
}

PCR_EXPORT int pcr_three_nn_f32(const float *unknown, const float *known, float *dist2, int *idx,
                                int B, int N, int M, pcr_stream_t stream) {
  if (!unknown || !known || !dist2 || !idx || B < 0 || N < 0 || M < 1) return PCR_ERR_INVALID;
  if (B == 0 || N == 0) return PCR_OK;
  if (B > 65535) return PCR_ERR_INVALID;
  hipLaunchKernelGGL(three_nn_kernel, dim3((N + 255) / 256, B), dim3(256), 0, pcr_s(stream),
                     unknown, known, dist2, idx, N, M);
  PCR_CHECK_LAUNCH();
  return PCR_OK;
}

PCR_EXPORT int pcr_three_interp_fwd_f32(const float *feat, const int *idx, const float *weight,
                                        float *out, int B, int C, int M, int N,
                                        pcr_stream_t stream) {
  if (!feat || !idx || !weight || !out || B < 0 || C < 0 || M < 1 || N < 0) return PCR_ERR_INVALID;
  size_t total = (size_t)B * C * N;
  if (total == 0) return PCR_OK;
  size_t blocks = (total + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(three_interp_fwd_kernel, dim3((unsigned)blocks), dim3(256), 0, pcr_s(stream),
                     feat, idx, weight, out, C, M, N, total);
  PCR_CHECK_LAUNCH();
  return PCR_OK;
}

PCR_EXPORT int pcr_three_interp_bwd_f32(const float *grad_out, const int *idx, const float *weight,
                                        float *grad_feat, int B, int C, int N, int M,
                                        pcr_stream_t stream) {
  if (!grad_out || !idx || !weight || !grad_feat || B < 0 || C < 0 || M < 1 || N < 0)
    return PCR_ERR_INVALID;
  size_t total = (size_t)B * C * N;
  if (total == 0) return PCR_OK;
  size_t blocks = (total + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  // entries e = 3 n + j scatter grad_out[n] * weight[n][j] to idx[n][j]: owner-computes (deterministic) when the
  // accumulators fit LDS, the atomic kernel otherwise
  if (!scatter_owner_launch(grad_out, idx, weight, grad_feat, B, C, M, 3 * N, 3, pcr_s(stream)))
    hipLaunchKernelGGL(three_interp_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, pcr_s(stream),
                       grad_out, idx, weight, grad_feat, C, N, M, total);
  PCR_CHECK_LAUNCH();
  return PCR_OK;
}

static int knn_prefix_launch(const float *xyz, int *idx, int B, int N, int S, int K, int *idx2, int S2, int K2,
                             pcr_stream_t stream) {
  if (!xyz || !idx || B < 0 || N < 1 || S < 0 || S > N || K < 1 || K > 64 || K > N || N > 4096)
    return PCR_ERR_INVALID;
  if (idx2 && (S2 < 0 || S2 > S || K2 < K || K2 > 64 || K2 > N)) return PCR_ERR_INVALID;
  if (!idx2) S2 = K2 = 0;
  if (B == 0 || S == 0) return PCR_OK;
  if (B > 65535) return PCR_ERR_INVALID;
  const int qpw = 64;   // queries per workgroup: the cloud is staged once per workgroup (32 / 128 / 256 measured: no better)
  dim3 g((S + qpw - 1) / qpw, B), blk(kKnnPThreads);
  size_t lds = (size_t)(3 * N + (N & 1)) * sizeof(float);
  hipStream_t st = pcr_s(stream);
#define PCR_KNN_REG(TP)                                                                                  \
  hipLaunchKernelGGL((knn_prefix_reg_kernel<TP>), g, blk,                                                \
                     lds + (size_t)4 * (kKnnCap + 64) * 8, st, xyz, idx, N, S, K, qpw, idx2, S2, K2)
  if (N <= 128) PCR_KNN_REG(1);
  else if (N <= 256) PCR_KNN_REG(2);
  else if (N <= 512) PCR_KNN_REG(4);
  else {
    // sixteen waves share one copy of the cloud in LDS (4 waves per SIMD already at one workgroup per CU)
    constexpr int NT = 1024;
    // queries per workgroup of the LDS form: 512 for clouds of more than 2048 points when that still leaves four workgroups
    // per CU (its 48 KB staging is then amortised over 32 queries per wave instead of 8: the 4096 x 4096 launch of pt4096
    // 3.95 -> 3.71 ms in one process, profiles/r06_knn_qpw_ab.txt; the 2048-point launch loses 3 % with it), else 128.
    // (May look at B: the output is an index list, the same whatever the split.)
    const int qpw_l = (N > 2048 && (long)((S + 511) / 512) * B >= 1024) ? 512 : 128;
    const dim3 gl((S + qpw_l - 1) / qpw_l, B);
    if (N <= 1024) {
      PCR_KNN_REG(8);   // (the two-pass LDS form measures 1.79 ms against 1.60 here: registers win while they fit)
    } else if (N <= 2048) {
      lds = (size_t)64 * 32 * 12 + (size_t)(NT / 64) * kKnnCap * 8;
      static bool big = hipFuncSetAttribute(reinterpret_cast<const void *>(knn_prefix_lds_kernel<32, NT>),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess;
      (void)big;
      hipLaunchKernelGGL((knn_prefix_lds_kernel<32, NT>), gl, dim3(NT), lds, st, xyz, idx, N, S, K, qpw_l, idx2, S2, K2);
    } else {
      lds = (size_t)64 * 64 * 12 + (size_t)(NT / 64) * kKnnCap * 8;
      static bool big = hipFuncSetAttribute(reinterpret_cast<const void *>(knn_prefix_lds_kernel<64, NT>),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess;
      (void)big;
      hipLaunchKernelGGL((knn_prefix_lds_kernel<64, NT>), gl, dim3(NT), lds, st, xyz, idx, N, S, K, qpw_l, idx2, S2, K2);
    }
  }
#undef PCR_KNN_REG
  PCR_CHECK_LAUNCH();
  return PCR_OK;
}

PCR_EXPORT int pcr_knn_prefix_f32(const float *xyz, int *idx, int B, int N, int S, int K, pcr_stream_t stream) {
  return knn_prefix_launch(xyz, idx, B, N, S, K, nullptr, 0, 0, stream);
}

PCR_EXPORT int pcr_knn_prefix2_f32(const float *xyz, int *idx, int *idx2, int B, int N, int S, int K, int S2, int K2,
                                   pcr_stream_t stream) {
  if (!idx2) return PCR_ERR_INVALID;
  return knn_prefix_launch(xyz, idx, B, N, S, K, idx2, S2, K2, stream);
}

