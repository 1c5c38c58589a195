// Pooling + match head, generic dense layer, host-side weight packing.
#include <stdlib.h>

#include <string.h>

#include "tile_dense.h"

namespace {
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

// the same reductions without LDS round trips: four DPP steps inside a row of 16 lanes (xor 1, xor 2, half-row mirror, row
// mirror: afterwards every lane holds its row's result), then the four rows' results by v_readlane.  (__shfl_xor is
// ds_bpermute: six dependent LDS round trips per reduction, and pool_head_kernel does 128 of them per pair.)
template <int CTRL>
__device__ __forceinline__ float dpp_f32(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float wave_sum_dpp(float v) {
  v += dpp_f32<0xB1>(v);
  v += dpp_f32<0x4E>(v);
  v += dpp_f32<0x141>(v);
  v += dpp_f32<0x140>(v);
  const float a = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0));
  const float b = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16));
  const float c = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 32));
  const float d = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 48));
  return (a + b) + (c + d);
}
__device__ __forceinline__ float wave_max_dpp(float v) {
  v = fmaxf(v, dpp_f32<0xB1>(v));
  v = fmaxf(v, dpp_f32<0x4E>(v));
  v = fmaxf(v, dpp_f32<0x141>(v));
  v = fmaxf(v, dpp_f32<0x140>(v));
  const float a = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0));
  const float b = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16));
  const float c = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 32));
  const float d = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 48));
  return fmaxf(fmaxf(a, b), fmaxf(c, d));
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
  constexpr int NW = kThreads / 64;
  if ((L & 3) == 0 && (reinterpret_cast<size_t>(p.o) & 15) == 0) {
    // round 4: 16-byte loads, BOTH clouds of a channel in one instruction (lanes 0-31: points 4 q .. + 3 of o1, lanes 32-63
    // of o2), eight channels of a wave in flight, DPP reductions.  The gallery's 36 k pairs read 2.1 GB here at 2.3 TB/s with
    // 4-byte loads, four channels in flight and bpermute reductions (0.9 ms).
    constexpr int CH = 8;
    const int q = lane & 31;
    const float *ob = lane < 32 ? o1 : o2;
    for (int c0 = wave * CH; c0 < C; c0 += NW * CH) {
      float mx[CH], sm[CH];
#pragma unroll
      for (int u = 0; u < CH; u++) {
        mx[u] = -INFINITY;
        sm[u] = 0.f;
      }
      for (int i0 = 0; i0 < L; i0 += 128) {
        const bool ok = i0 + 4 * q < L;
        f32x4 v[CH];
#pragma unroll
        for (int u = 0; u < CH; u++) {
          const int c = c0 + u < C ? c0 + u : c0;
          v[u] = *reinterpret_cast<const f32x4 *>(ob + (size_t)c * L + (ok ? i0 + 4 * q : 0));
        }
        if (ok) {
#pragma unroll
          for (int u = 0; u < CH; u++) {
            mx[u] = fmaxf(mx[u], fmaxf(fmaxf(v[u][0], v[u][1]), fmaxf(v[u][2], v[u][3])));
            sm[u] += (v[u][0] + v[u][1]) + (v[u][2] + v[u][3]);
          }
        }
      }
#pragma unroll
      for (int u = 0; u < CH; u++) {
        const int c = c0 + u;
        const float m1 = wave_max_dpp(mx[u]), s1 = wave_sum_dpp(sm[u]);
        if (lane == 0 && c < C) { x[c] = m1; x[C + c] = s1 / (float)(2 * L); }
      }
    }
  } else
  // four channels of a wave in flight at a time (one channel per trip left every load waiting on the previous trip's
  // reductions: 42 us per pair at 36 k pairs); same per-lane order of the sums
  for (int c0 = wave; c0 < C; c0 += 4 * NW) {
    float mx[4], sm[4];
#pragma unroll
    for (int u = 0; u < 4; u++) {
      mx[u] = -INFINITY;
      sm[u] = 0.f;
    }
    for (int i = lane; i < L; i += 64) {
      float av[4], bv[4];
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const int c = c0 + u * NW < C ? c0 + u * NW : c0;
        av[u] = o1[(size_t)c * L + i];
        bv[u] = o2[(size_t)c * L + i];
      }
#pragma unroll
      for (int u = 0; u < 4; u++) {
        mx[u] = fmaxf(mx[u], fmaxf(av[u], bv[u]));
        sm[u] += av[u] + bv[u];
      }
    }
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const int c = c0 + u * NW;
      const float m1 = wave_max(mx[u]), s1 = wave_sum(sm[u]);
      if (lane == 0 && c < C) { x[c] = m1; x[C + c] = s1 / (float)(2 * L); }
    }
  }
  __syncthreads();
  if (p.pooled && tid < n) p.pooled[pr * n + tid] = x[tid];
  // (the matvec rows in 16-byte pieces, the products added in the original order; with the transposed matrix the lanes of
  // a wave read consecutive floats -- one 256-byte run per step instead of 64 separate 16-byte pieces per load, which was
  // what bound a gallery launch: 8192 pieces per pair through one CU's address unit -- eight steps in flight, the same sums)
  auto matvec = [&](const float *wm, const float *wt, const float *v) {
    if (wt) {
      const float *w = wt + tid;
      float s = 0.f;
      int i = 0;
      for (; i + 8 <= n; i += 8) {
        float wv[8];
#pragma unroll
        for (int u = 0; u < 8; u++) wv[u] = w[(size_t)(i + u) * n];
#pragma unroll
        for (int u = 0; u < 8; u++) s += wv[u] * v[i + u];
      }
      for (; i < n; i++) s += w[(size_t)i * n] * v[i];
      return s;
    }
    const float *w = wm + (size_t)tid * n;
    float s = 0.f;
    if ((n & 3) == 0 && (reinterpret_cast<size_t>(wm) & 15) == 0) {
      for (int i = 0; i < n; i += 4) {
        const f32x4 w4 = *reinterpret_cast<const f32x4 *>(w + i);
        s += w4[0] * v[i];
        s += w4[1] * v[i + 1];
        s += w4[2] * v[i + 2];
        s += w4[3] * v[i + 3];
      }
    } else {
      for (int i = 0; i < n; i++) s += w[i] * v[i];
    }
    return s;
  };
  if (tid < n) y[tid] = matvec(p.w1, p.w1t, x);
  __syncthreads();
  vec_groupnorm(y, n, p.groups, p.gn1_g, p.gn1_b);
  if (tid < n) y[tid] = fmaxf(y[tid], 0.f);
  __syncthreads();
  if (tid < n) z[tid] = matvec(p.w2, p.w2t, y);
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

// get_pooled_feats with pool_type='max' (models/ReIDNet.py:145,526-528): nn.MaxPool1d(window) applied to the PERMUTED
// (B,L,C) tensor, i.e. a max over windows of `window` consecutive CHANNELS of every point (floor mode: a trailing
// partial window is dropped).  x (B,C,L) channel-major -> out (B,L,G), G = C / window.  Thread = point: the loads of a
// wave are 256 contiguous bytes per channel.
__global__ __launch_bounds__(kThreads) void channel_max_kernel(const float *__restrict__ x, float *__restrict__ out,
                                                               int C, int L, int window, int G) {
  const size_t b = blockIdx.y;
  const int l = blockIdx.x * kThreads + threadIdx.x;
  if (l >= L) return;
  const float *xp = x + b * C * L + l;
  for (int g = 0; g < G; g++) {
    float m = -INFINITY;
    for (int c = g * window; c < (g + 1) * window; c++) m = fmaxf(m, xp[(size_t)c * L]);
    out[(b * L + l) * G + g] = m;
  }
}

// ------------------------------------------------------------------- generic dense ----
// y (B,cout,L) = act(scale * (W x) + shift) for x (B,cin,L); the whole cin extent of a token tile
// sits in LDS, each workgroup produces one chunk of up to 256 output channels (grid.z) and stores
// it straight from the accumulators.  w_bstride != 0: per-cloud weights (PointNet's learned 3x3 /
// 64x64 transforms, i.e. torch.bmm), packed image of cloud b at wp + b * w_bstride.
struct DenseArgs {
  const float *x, *wp, *scale, *shift;
  float *y;
  int cin, cout, L, act;
  long w_bstride;
  int x_pm;   // x is (B,L,cin)
  int gn_gs;  // GN kernels only: channels per GroupNorm group (4, 8, 16 or 32); scale / shift are gamma / beta
  const float *res;   // GN kernels only: optional residual (B,cout,L) added after the normalisation
};

// GroupNorm of one 32-cout x 32-token accumulator tile over groups of GS consecutive channels of each token (nn.GroupNorm
// on (M,C) rows, lanegcn_nets.py:228-241): a lane holds couts 8g + 4h + {0..3} (g = 0..3) of token l31, so a group of
// 8 / 16 / 32 channels is 1 / 2 / 4 bands of this lane plus the same bands of lane ^ 32; groups of 4 are lane-local.
// Two-pass mean / biased variance, eps 1e-5.  Every lane of the wave must call it (cross-lane exchange).
template <int GS>
__device__ __forceinline__ void gn_tile(const f32x16 &acc, float (&y)[16]) {
  constexpr int NB = GS >= 8 ? GS / 8 : 1;
  constexpr bool kCross = GS >= 8;
#pragma unroll
  for (int g0 = 0; g0 < 4; g0 += NB) {
    float s = 0.f;
#pragma unroll
    for (int i = 4 * g0; i < 4 * (g0 + NB); i++) s += acc[i];
    if (kCross) s += __shfl_xor(s, 32, 64);
    const float mean = s * (1.0f / (float)GS);
    float v = 0.f;
#pragma unroll
    for (int i = 4 * g0; i < 4 * (g0 + NB); i++) {
      const float d = acc[i] - mean;
      v += d * d;
    }
    if (kCross) v += __shfl_xor(v, 32, 64);
    const float inv = 1.0f / sqrtf(v * (1.0f / (float)GS) + 1e-5f);
#pragma unroll
    for (int i = 4 * g0; i < 4 * (g0 + NB); i++) y[i] = (acc[i] - mean) * inv;
  }
}

// CH (chunked contraction, cin and cout multiples of 256): only kChunk input channels of the token tile sit in LDS at
// a time and the accumulators are carried across chunks, so a 1024-channel layer keeps a 64-token tile in 67 KB (two
// workgroups per CU) where the whole-extent form needs 135 KB for 32 tokens -- and every weight byte streamed from L2
// now feeds twice the tokens (at 32 tokens per tile the 1024 -> 512 layers asked L2 for ~10 TB/s of weights).
constexpr int kChunk = 256;

template <int TB, bool GN = false, bool CH = false>
__global__ __launch_bounds__(kThreads) void dense_kernel(DenseArgs a) {
  constexpr int T = 32 * TB, RP = T + 1;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int cinP = CH ? kChunk : ceil8(a.cin), coutP = ceil32(a.cout);
  float *X = smem;
  float *s_sc = X + cinP * RP;   // [256] scale and [256] shift of this workgroup's cout chunk (1 / 0 when absent)
  float *s_sh = s_sc + 256;
  const size_t b = blockIdx.y;
  const int t0 = blockIdx.x * T;
  const int chunk0 = blockIdx.z * 256;
  const int chunkP = coutP - chunk0 < 256 ? coutP - chunk0 : 256;
  {
    const int oc = chunk0 + threadIdx.x;   // kThreads == 256
    s_sc[threadIdx.x] = (a.scale && oc < a.cout) ? a.scale[oc] : 1.0f;
    s_sh[threadIdx.x] = (a.shift && oc < a.cout) ? a.shift[oc] : 0.0f;
  }
  if constexpr (!CH) {
    if (a.x_pm) load_tile_pm(X, RP, a.x + b * a.cin * a.L, a.cin, cinP, a.L, t0, T);
    else load_tile(X, RP, a.x + b * a.cin * a.L, a.cin, cinP, a.L, t0, T);
    __syncthreads();
  }
  const int cout = a.cout, act = a.act, L = a.L;
  float *out = a.y + b * a.cout * a.L;
  // packed image rows [chunk0, chunk0+chunkP) of every k-block: offset chunk0*8 floats, stride coutP.
  // Pipelined dense tile (weights through a register ring, B operands one k-block ahead); the epilogue takes a
  // whole 32x32 tile: the lane's 16 couts are four runs of four, so scale / shift are eight 16-byte LDS reads.
  const float *wp = a.wp + b * a.w_bstride + (size_t)chunk0 * 8;
  auto epi = [&](const f32x16 &acc, int cb, int tb, int l31, int h) {
    const int t = tb * 32 + l31;
    if constexpr (GN) {
      // y = act(GN(W x) * gamma + beta [+ res]): the Linear -> GroupNorm [-> + shortcut] [-> ReLU] steps of LinearRes
      float nrm[16];
      switch (a.gn_gs) {
        case 4: gn_tile<4>(acc, nrm); break;
        case 8: gn_tile<8>(acc, nrm); break;
        case 16: gn_tile<16>(acc, nrm); break;
        default: gn_tile<32>(acc, nrm); break;
      }
      if (t0 + t < L) {
        const float *res = a.res ? a.res + b * a.cout * a.L : nullptr;
#pragma unroll
        for (int g = 0; g < 4; g++) {
          const int o = cb * 32 + 8 * g + 4 * h;
          const f32x4 s4 = *reinterpret_cast<const f32x4 *>(s_sc + o), b4 = *reinterpret_cast<const f32x4 *>(s_sh + o);
#pragma unroll
          for (int q = 0; q < 4; q++) {
            const int oc = chunk0 + o + q;
            if (oc < cout) {
              float r = nrm[4 * g + q] * s4[q] + b4[q];
              if (res) r += res[(size_t)oc * L + t0 + t];
              out[(size_t)oc * L + t0 + t] = act == 1 ? fmaxf(r, 0.f) : r;
            }
          }
        }
      }
      return;
    }
    if (t0 + t < L) {
#pragma unroll
      for (int g = 0; g < 4; g++) {
        const int o = cb * 32 + 8 * g + 4 * h;
        const f32x4 s4 = *reinterpret_cast<const f32x4 *>(s_sc + o), b4 = *reinterpret_cast<const f32x4 *>(s_sh + o);
#pragma unroll
        for (int q = 0; q < 4; q++) {
          const int oc = chunk0 + o + q;
          if (oc < cout) {
            const float r = acc[4 * g + q] * s4[q] + b4[q];
            out[(size_t)oc * L + t0 + t] = act == 1 ? fmaxf(r, 0.f) : (act == 2 && r < 0.f) ? r * 0.2f : r;
          }
        }
      }
    }
  };
  if constexpr (CH) {
    // chunkP == 256: eight cout blocks, two rounds per wave; accumulators carried over the cin chunks
    f32x16 carry[2][TB];
#pragma unroll
    for (int nr = 0; nr < 2; nr++)
#pragma unroll
      for (int j = 0; j < TB; j++)
#pragma unroll
        for (int r = 0; r < 16; r++) carry[nr][j][r] = 0.f;
    const float *xb = a.x + b * a.cin * a.L;
    const int nch = a.cin / kChunk;
    const size_t wchunk = (size_t)(kChunk / 8) * coutP * 8;   // floats of kChunk input channels in the packed image
    const DenseNoHook nh;
    for (int c = 0; c < nch; c++) {
      if (c) __syncthreads();   // everyone is done with the previous chunk's operands
      load_tile(X, RP, xb + (size_t)c * kChunk * a.L, kChunk, kChunk, a.L, t0, T);
      __syncthreads();
      tile_dense_impl<TB, 2, 1, true, decltype(epi), PCR_PF, DenseNoHook, false, true, true>(
          X, kChunk, wp + c * wchunk, 256, false, epi, nullptr, nullptr, nh, coutP, carry);
    }
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
#pragma unroll
    for (int nr = 0; nr < 2; nr++)
#pragma unroll
      for (int j = 0; j < TB; j++) epi(carry[nr][j], wave + 4 * nr, j, lane & 31, lane >> 5);
  } else {
    tile_dense2<TB, 2, 0, true>(X, cinP, wp, chunkP, false, epi, nullptr, nullptr, DenseNoHook(), coutP);
  }
}

// The same layer followed by the max over the points (STN3d / STNkd: conv3 + BN + ReLU, then torch.max over N,
// models/pointnet.py:27-33, 67-73) WITHOUT the (B,cout,L) tensor: one workgroup owns a cloud's 256-cout chunk, walks the
// cloud's 64-token tiles and keeps the running maximum of every (cout, token lane) in registers; at the end the 32 token
// lanes of a cout meet by DPP and one lane stores out[c * B + b].  (Round 4: the dense launch wrote 2.1 GB per 2048
// clouds and pcr_max_over_l_f32 read them back, 0.64 ms, twice per forward.)  A maximum is order-independent: bit-equal
// to the two-launch form.
__global__ __launch_bounds__(kThreads) void dense_max_kernel(DenseArgs a, int B) {
  constexpr int TB = 2, T = 64, RP = T + 1;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int cinP = ceil8(a.cin), coutP = ceil32(a.cout);
  float *X = smem;
  float *s_sc = X + cinP * RP;
  float *s_sh = s_sc + 256;
  const size_t b = blockIdx.y;
  const int chunk0 = blockIdx.z * 256;
  const int chunkP = coutP - chunk0 < 256 ? coutP - chunk0 : 256;
  {
    const int oc = chunk0 + threadIdx.x;
    s_sc[threadIdx.x] = (a.scale && oc < a.cout) ? a.scale[oc] : 1.0f;
    s_sh[threadIdx.x] = (a.shift && oc < a.cout) ? a.shift[oc] : 0.0f;
  }
  const int act = a.act, L = a.L;
  const float *wp = a.wp + (size_t)chunk0 * 8;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  float vmax[2][16];
#pragma unroll
  for (int nr = 0; nr < 2; nr++)
#pragma unroll
    for (int r = 0; r < 16; r++) vmax[nr][r] = -INFINITY;
  for (int t0 = 0; t0 < L; t0 += T) {
    if (t0) __syncthreads();   // everyone is done with the previous tile's operands
    load_tile(X, RP, a.x + b * a.cin * a.L, a.cin, cinP, a.L, t0, T);
    __syncthreads();
    auto epi = [&](const f32x16 &acc, int cb, int tb, int l31, int h) {
      const int nr = cb >> 2;       // cout blocks wave, wave + 4 (chunkP <= 256: at most two rounds)
      if (t0 + tb * 32 + l31 < L) {
#pragma unroll
        for (int g = 0; g < 4; g++) {
          const int o = cb * 32 + 8 * g + 4 * h;
          const f32x4 s4 = *reinterpret_cast<const f32x4 *>(s_sc + o), b4 = *reinterpret_cast<const f32x4 *>(s_sh + o);
#pragma unroll
          for (int q = 0; q < 4; q++) {
            float r = acc[4 * g + q] * s4[q] + b4[q];
            r = act == 1 ? fmaxf(r, 0.f) : (act == 2 && r < 0.f) ? r * 0.2f : r;
            if (nr == 0) vmax[0][4 * g + q] = fmaxf(vmax[0][4 * g + q], r);
            else vmax[1][4 * g + q] = fmaxf(vmax[1][4 * g + q], r);
          }
        }
      }
    };
    tile_dense2<TB, 2, 0, true>(X, cinP, wp, chunkP, false, epi, nullptr, nullptr, DenseNoHook(), coutP);
  }
  // the 32 token lanes of a half (h) hold the same 16 couts: maximum across them, then lanes 0 and 32 store
#pragma unroll
  for (int nr = 0; nr < 2; nr++) {
    const int cb = wave + 4 * nr;
    if (cb < (chunkP >> 5)) {
#pragma unroll
      for (int r = 0; r < 16; r++) {
        float v = vmax[nr][r];
        v = fmaxf(v, __shfl_xor(v, 1, 64));
        v = fmaxf(v, __shfl_xor(v, 2, 64));
        v = fmaxf(v, __shfl_xor(v, 4, 64));
        v = fmaxf(v, __shfl_xor(v, 8, 64));
        v = fmaxf(v, __shfl_xor(v, 16, 64));
        const int oc = chunk0 + cb * 32 + 8 * (r >> 2) + 4 * (lane >> 5) + (r & 3);
        if ((lane & 31) == 0 && oc < a.cout) a.y[(size_t)oc * B + b] = v;
      }
    }
  }
}

// ---- f32 layers with a short contraction (cin <= 128: PointNet's conv2 / conv3 and the STN convs, which stay on the
// f32-input MFMA: DESIGN section 5) with the WEIGHTS RESIDENT (round 5).  The chunked / whole-extent kernels above stream the
// packed weights from L2 through a register ring for every 64-token tile and run load -> barrier -> MFMA -> epilogue as
// serial phases of a workgroup (dense_max[128 -> 1024]: 0.44 of the f32-MFMA peak).  Here a workgroup owns 128 couts of one
// cloud for ALL of the cloud's tiles: wave w keeps cout block w's whole A operand in registers (cin / 8 x 16 bytes per
// lane: 64 registers at cin = 128), so the tile loop contains no weight load at all -- which is what lets the NEXT tile's
// input be fetched into registers during the current tile's MFMAs (a wave's vector loads retire in order: behind a
// weight ring the prefetch would be waited for at the ring's first use) -- and ~150 registers leave three workgroups per CU.
// The k-steps run in the order of tile_dense_impl: the same bits as the kernels above.
//   MAXE: out[c * B + b] = max over the cloud's tokens (dense_max_kernel's contract), else y (B,cout,L).
template <int KB, bool MAXE>
__global__ __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(KB >= 16 ? 2 : 3, KB >= 16 ? 2 : 3))) void dense_rw_kernel(DenseArgs a, int B) {
  constexpr int TB = 2, T = 64, RP = T + 1, CP = 8 * KB, NQ = (CP * T / 4 + kThreads - 1) / kThreads;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float *X = smem;                    // [CP][RP]
  float *s_sc = X + CP * RP;          // [128] scale / shift of this workgroup's cout window
  float *s_sh = s_sc + 128;
  const int coutP = ceil32(a.cout);
  // (grid (window, cloud): the eight windows of a cloud sit on eight XCDs and each fetches the cloud's input itself -- 2.1 GB
  // of HBM reads for dense_max[128 -> 1024]'s 268 MB input, 1.4 TB/s, not the bound.  Measured and dropped: an XCD-aware
  // 1-D order that puts a cloud's windows on ONE XCD -- 1.42 -> 1.52 ms, the eight workgroups then ask one L2 for the same
  // lines at the same moment)
  const size_t b = blockIdx.y;
  const int chunk0 = blockIdx.x * 128;
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, h = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int cb = (chunk0 >> 5) + wave;                 // this wave's cout block (may lie beyond coutP: clamped, dropped)
  const bool cb_ok = cb * 32 < coutP;
  if (tid < 128) {
    const int oc = chunk0 + tid;
    s_sc[tid] = (a.scale && oc < a.cout) ? a.scale[oc] : 1.0f;
    s_sh[tid] = (a.shift && oc < a.cout) ? a.shift[oc] : 0.0f;
  }
  const int cin = a.cin, cout = a.cout, act = a.act, L = a.L;
  // the wave's A operand: k-block kb = four k-steps, f32x4 at [kb][cout][h]
  f32x4 wreg[KB];
  {
    const int cbc = cb_ok ? cb : (coutP >> 5) - 1;
    const f32x4 *wv = reinterpret_cast<const f32x4 *>(a.wp) + (size_t)(cbc * 32 + l31) * 2 + h;
    const size_t wstride = (size_t)coutP * 2;
    const int kbn = ceil8(cin) >> 3;             // k-blocks the packed image really has (KB is the template's bound)
#pragma unroll
    for (int kb = 0; kb < KB; kb++) {
      const f32x4 wv4 = wv[(size_t)(kb < kbn ? kb : kbn - 1) * wstride];
      wreg[kb] = kb < kbn ? wv4 : f32x4{0.f, 0.f, 0.f, 0.f};
    }
  }
  const float *xb = a.x + b * (size_t)cin * L;
  const bool vecL = (L & 3) == 0 && (reinterpret_cast<size_t>(xb) & 15) == 0;
  // the tile's 16-byte pieces of this thread: piece e = tid + u 256 -> (channel e / 16, token quad e % 16)
  f32x4 pre[NQ];
  auto request = [&](int t0) __attribute__((always_inline)) {
#pragma unroll
    for (int u = 0; u < NQ; u++) {
      const int e = tid + u * kThreads;
      const int c = e >> 4, qd = e & 15;
      const bool ok = e < CP * 16 && c < cin;
      const float *src = xb + (size_t)(ok ? c : 0) * L + t0 + 4 * qd;
      if (vecL && t0 + T <= L) {
        pre[u] = *reinterpret_cast<const f32x4 *>(src);
      } else {                                   // ragged / unaligned tiles: element loads from clamped addresses
#pragma unroll
        for (int k = 0; k < 4; k++) {
          const int t = t0 + 4 * qd + k;
          pre[u][k] = xb[(size_t)(ok ? c : 0) * L + (t < L ? t : L - 1)];
        }
      }
    }
  };
  auto commit = [&](int t0) __attribute__((always_inline)) {
#pragma unroll
    for (int u = 0; u < NQ; u++) {
      const int e = tid + u * kThreads;
      const int c = e >> 4, qd = e & 15;
      if (e < CP * 16) {
        float *d = X + c * RP + 4 * qd;
#pragma unroll
        for (int k = 0; k < 4; k++) d[k] = (c < cin && t0 + 4 * qd + k < L) ? pre[u][k] : 0.f;
      }
    }
  };
  float vmax[16];
#pragma unroll
  for (int r = 0; r < 16; r++) vmax[r] = -INFINITY;
  float *out = MAXE ? nullptr : a.y + b * (size_t)cout * L;
  request(0);
  for (int t0 = 0; t0 < L; t0 += T) {
    if (t0) __syncthreads();             // every wave is done with the previous tile's operands
    commit(t0);
    __syncthreads();
    if (t0 + T < L) request(t0 + T);     // (lands during this tile's MFMAs: nothing else of this wave loads from memory)
    f32x16 acc[TB];
#pragma unroll
    for (int j = 0; j < TB; j++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[j][r] = 0.f;
    const float *bp = X + h * RP + l31;
#pragma unroll
    for (int kb = 0; kb < KB; kb++) {
      float xq[TB][4];
#pragma unroll
      for (int j = 0; j < TB; j++)
#pragma unroll
        for (int qq = 0; qq < 4; qq++) xq[j][qq] = bp[(kb * 8 + 2 * qq) * RP + j * 32];
#pragma unroll
      for (int qq = 0; qq < 4; qq++)
#pragma unroll
        for (int j = 0; j < TB; j++) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(wreg[kb][qq], xq[j][qq], acc[j], 0, 0, 0);
    }
    if (cb_ok) {
#pragma unroll
      for (int j = 0; j < TB; j++) {
        const int t = t0 + j * 32 + l31;
        if (t < L) {
#pragma unroll
          for (int g = 0; g < 4; g++) {
            const int o = wave * 32 + 8 * g + 4 * h;       // row of the 128-cout window
            const f32x4 s4 = *reinterpret_cast<const f32x4 *>(s_sc + o), b4 = *reinterpret_cast<const f32x4 *>(s_sh + o);
#pragma unroll
            for (int qq = 0; qq < 4; qq++) {
              float r = acc[j][4 * g + qq] * s4[qq] + b4[qq];
              r = act == 1 ? fmaxf(r, 0.f) : (act == 2 && r < 0.f) ? r * 0.2f : r;
              if constexpr (MAXE) {
                vmax[4 * g + qq] = fmaxf(vmax[4 * g + qq], r);
              } else {
                const int oc = chunk0 + o + qq;
                if (oc < cout) out[(size_t)oc * L + t] = r;
              }
            }
          }
        }
      }
    }
  }
  if constexpr (MAXE) {
    if (cb_ok) {
#pragma unroll
      for (int r = 0; r < 16; r++) {
        float v = vmax[r];
        v = fmaxf(v, __shfl_xor(v, 1, 64));
        v = fmaxf(v, __shfl_xor(v, 2, 64));
        v = fmaxf(v, __shfl_xor(v, 4, 64));
        v = fmaxf(v, __shfl_xor(v, 8, 64));
        v = fmaxf(v, __shfl_xor(v, 16, 64));
        const int oc = chunk0 + wave * 32 + 8 * (r >> 2) + 4 * h + (r & 3);
        if (l31 == 0 && oc < cout) a.y[(size_t)oc * B + b] = v;
      }
    }
  }
}

// shapes the resident-weight kernel takes: cin <= 128 (a multiple of 8 after padding: the packed image's k-blocks)
static bool dense_rw_ok(int cin, int cout, int L) { return cin >= 16 && cin <= 128 && cout >= 128 && L >= 128; }

template <bool MAXE>
static int dense_rw_launch(const DenseArgs &a, int B, hipStream_t st) {
  const int KB = ceil8(a.cin) / 8;
  const dim3 g((ceil32(a.cout) + 127) / 128, B);
#define PCR_DRW(KBv)                                                                                       \
  do {                                                                                                     \
    const size_t lds = ((size_t)8 * KBv * 65 + 256) * sizeof(float);                                       \
    hipLaunchKernelGGL((dense_rw_kernel<KBv, MAXE>), g, dim3(kThreads), lds, st, a, B);                    \
  } while (0)
  if (KB <= 4) PCR_DRW(4);
  else if (KB <= 8) PCR_DRW(8);
  else PCR_DRW(16);
#undef PCR_DRW
  PCR_CHECK_LAUNCH();
  return PCR_OK;
}

// ---- the wide per-point layers on the bf16 matrix core (round 4): PointNet's 1x1 convs, DGCNN's conv5 and the LinearRes
// downsample rows were the last MFMA-bound launches in f32 (0.7 of a 157 TFLOP/s roof).  Same tiling as the chunked f32
// form -- 64 tokens x up to 256 couts per workgroup, the cin extent walked in chunks of KC channels with the accumulators
// carried -- but a chunk sits in LDS as a bf IMAGE (tile_dense.h): every thread loads the eight channels of a 16-byte piece
// for its token (coalesced along the tokens), splits them into bf16 hi / lo ONCE, and the four waves read the pieces as
// MFMA operands (two ds_read_b128 per three MFMAs per token block).  wp: pcr_pack_weight_bf16x2_f32 image.
// NS = 3: split bf16 (W x ~ W_hi x_hi + W_hi x_lo + W_lo x_hi, f32 accumulate), NS = 1: plain bf16.
template <bool GN, int NS, int NR>
__global__ __launch_bounds__(kThreads) void dense_bf_kernel(DenseArgs a, int KC) {
  constexpr int TB = 2, T = 64;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float *img = smem;                 // [KC / 8 pieces][hi, lo][T] 16-byte units
  float *s_sc = img + KC * T;
  float *s_sh = s_sc + 256;
  const int coutP = ceil32(a.cout);
  const size_t b = blockIdx.y;
  const int t0 = blockIdx.x * T;
  const int chunk0 = blockIdx.z * 256;
  const int chunkP = coutP - chunk0 < 256 ? coutP - chunk0 : 256;
  {
    const int oc = chunk0 + threadIdx.x;   // kThreads == 256
    s_sc[threadIdx.x] = (a.scale && oc < a.cout) ? a.scale[oc] : 1.0f;
    s_sh[threadIdx.x] = (a.shift && oc < a.cout) ? a.shift[oc] : 0.0f;
  }
  const int cout = a.cout, act = a.act, L = a.L;
  float *out = a.y + b * a.cout * a.L;
  auto epi = [&](const f32x16 &acc, int cb, int tb, int l31, int h) {
    const int t = tb * 32 + l31;
    float v[16];
    if constexpr (GN) {
      switch (a.gn_gs) {
        case 4: gn_tile<4>(acc, v); break;
        case 8: gn_tile<8>(acc, v); break;
        case 16: gn_tile<16>(acc, v); break;
        default: gn_tile<32>(acc, v); break;
      }
    } else {
#pragma unroll
      for (int r = 0; r < 16; r++) v[r] = acc[r];
    }
    if (t0 + t < L) {
      const float *res = (GN && a.res) ? a.res + b * a.cout * a.L : nullptr;
      // the residual's sixteen values as ONE batch of loads from clamped addresses (a load inside the store loop below
      // is a load + s_waitcnt vmcnt(0) per element)
      float rs[16];
#pragma unroll
      for (int r = 0; r < 16; r++) {
        const int oc = chunk0 + cb * 32 + 8 * (r >> 2) + 4 * h + (r & 3);
        rs[r] = res ? res[(size_t)(oc < cout ? oc : cout - 1) * L + t0 + t] : 0.f;
      }
#pragma unroll
      for (int g = 0; g < 4; g++) {
        const int o = cb * 32 + 8 * g + 4 * h;
        const f32x4 s4 = *reinterpret_cast<const f32x4 *>(s_sc + o), b4 = *reinterpret_cast<const f32x4 *>(s_sh + o);
#pragma unroll
        for (int q = 0; q < 4; q++) {
          const int oc = chunk0 + o + q;
          if (oc < cout) {
            const float r = v[4 * g + q] * s4[q] + b4[q] + rs[4 * g + q];
            out[(size_t)oc * L + t0 + t] = act == 1 ? fmaxf(r, 0.f) : (act == 2 && r < 0.f) ? r * 0.2f : r;
          }
        }
      }
    }
  };
  f32x16 carry[NR][TB];
#pragma unroll
  for (int nr = 0; nr < NR; nr++)
#pragma unroll
    for (int j = 0; j < TB; j++)
#pragma unroll
      for (int r = 0; r < 16; r++) carry[nr][j][r] = 0.f;
  const float *xb = a.x + b * a.cin * a.L;
  const int nch = a.cin / KC, npiece = KC >> 3;
  const size_t wstep = (size_t)(coutP >> 5) * 128 * 4;               // floats of one 16-channel step of the image
  const float *wp = a.wp + (size_t)(chunk0 >> 5) * 128 * 4;
  const int tt = threadIdx.x & 63, pg = threadIdx.x >> 6;
  const bool tok_ok = t0 + tt < L;
  const float *xt = xb + (tok_ok ? t0 + tt : 0);
  bf16x8 *u = reinterpret_cast<bf16x8 *>(img);
  const DenseNoHook nh;
  const bool vec = (L & 3) == 0 && t0 + T <= L && (reinterpret_cast<size_t>(xb) & 15) == 0;
  for (int c = 0; c < nch; c++) {
    if (c) __syncthreads();   // everyone is done with the previous chunk's operands
    if (vec) {
      // 16-byte loads: lane (u, pgrp) takes tokens 4 u .. + 3 of the eight channels of piece pgrp, pgrp + 16, ...
      // (16 lanes x 16 bytes = one 256-byte run per channel) and forms the four tokens' pieces in registers
      const int uq = threadIdx.x & 15, pgrp = threadIdx.x >> 4;
      for (int P = pgrp; P < npiece; P += 16) {
        const int cbase = c * KC + 16 * (P >> 1);
        f32x4 v[8];
#pragma unroll
        for (int j = 0; j < 8; j++)
          v[j] = *reinterpret_cast<const f32x4 *>(xb + (size_t)(cbase + bf_kpos(P & 1, j)) * L + t0 + 4 * uq);
#pragma unroll
        for (int k = 0; k < 4; k++) {
          float x8[8];
#pragma unroll
          for (int j = 0; j < 8; j++) x8[j] = v[j][k];
          bf16x8 hi, lo;
          bf_split8(x8, hi, lo, NS == 3);
          u[(2 * P) * T + 4 * uq + k] = hi;
          if constexpr (NS == 3) u[(2 * P + 1) * T + 4 * uq + k] = lo;
        }
      }
    } else
    // pieces pg, pg + 4, ...: piece P = channels 16 (P >> 1) + bf_kpos(P & 1, j) of the chunk, two at a time in flight
    for (int P0 = pg; P0 < npiece; P0 += 8) {
      float x[2][8];
#pragma unroll
      for (int k = 0; k < 2; k++) {
        const int P = P0 + 4 * k;
        const int cbase = c * KC + 16 * (P >> 1);
#pragma unroll
        for (int j = 0; j < 8; j++) {
          const int ch = cbase + bf_kpos(P & 1, j);
          x[k][j] = P < npiece ? xt[(size_t)ch * L] : 0.f;
        }
      }
#pragma unroll
      for (int k = 0; k < 2; k++) {
        const int P = P0 + 4 * k;
        if (P < npiece) {
          if (!tok_ok) {
#pragma unroll
            for (int j = 0; j < 8; j++) x[k][j] = 0.f;
          }
          bf16x8 hi, lo;
          bf_split8(x[k], hi, lo, NS == 3);
          u[(2 * P) * T + tt] = hi;
          if constexpr (NS == 3) u[(2 * P + 1) * T + tt] = lo;
        }
      }
    }
    __syncthreads();
    tile_dense_bf_impl<TB, NR, 1, true, NS, decltype(epi), bf_pf(NR), DenseNoHook, true, false, true, true>(
        img, KC, wp + (size_t)c * (KC >> 4) * wstep, chunkP, false, epi, nullptr, nh, coutP, nullptr, carry);
  }
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
#pragma unroll
  for (int nr = 0; nr < NR; nr++)
    if (wave + 4 * nr < (chunkP >> 5)) {
#pragma unroll
      for (int j = 0; j < TB; j++) epi(carry[nr][j], wave + 4 * nr, j, lane & 31, lane >> 5);
    }
}

// ---- the same layer with the two halves of the work on DIFFERENT waves (round 5): 128 tokens x 256 couts per workgroup of
// eight waves.  Waves 4-7 ("producers") fetch a chunk of 128 input channels, split it into bf16 hi / lo once and write the
// bf image; waves 0-3 ("consumers") run the MFMAs of the previous chunk out of the other image buffer with their weights
// streamed from L2 through the register ring -- one workgroup barrier per chunk, nothing else shared.  Why two roles and not
// a register prefetch inside one wave: a wave's vector-memory loads retire in order, so an HBM fetch of the NEXT chunk
// (~2 us) that is older than a weight load of the CURRENT one (L2, needed two k-steps later) stalls the first s_waitcnt of
// the ring for the whole HBM latency; in separate waves the two streams have separate counters.  Against the one-role
// kernel above (64-token tiles, load -> split -> barrier -> MFMA in every wave, two workgroups per CU: matrix pipe 0.36
// busy, every weight byte fetched from L2 per 64 tokens = ~13 TB/s of L2 reads at the 1024 -> 512 layers) a weight
// byte now feeds 128 tokens and the consumers never wait for HBM.  Shapes: cin % 128 == 0, L % 128 == 0, 16-byte aligned x;
// everything else keeps dense_bf_kernel.  Same arithmetic, same order of the k-steps: bit-identical outputs.
constexpr int kPcT = 128, kPcKC = 128, kPcThreads = 512;
constexpr int kPcOP = 136;           // floats per cout row of the output staging tile (4 rows apart = 32 banks apart)
constexpr size_t kPcLds = ((size_t)256 * kPcOP + 512) * sizeof(float);   // >= the two images (2 x 64 KB)

template <bool GN, int NS, int NCH>
__global__ __launch_bounds__(kPcThreads) void dense_bf_pc_kernel(DenseArgs a) {
  constexpr int TB = 4, T = kPcT, KC = kPcKC, NR = 2, OP = kPcOP;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float *img0 = smem;                // two images of [KC / 8 pieces][hi, lo][T] 16-byte units; then the output staging tile
  float *s_sc = smem + 256 * OP;
  float *s_sh = s_sc + 256;
  const int coutP = ceil32(a.cout);
  // XCD-aware order of a 1-D grid: workgroup n runs on XCD n % 8 (round-robin dispatch); the nz cout windows of one
  // token tile take consecutive slots of ONE XCD, so they are resident together and the tile's input comes from HBM once
  // and from that XCD's L2 for the other windows (as a 3-D grid the windows of a tile were a whole grid apart: the
  // 1024 -> 512 layers read their 2.1 GB input twice, 6.3 GB per launch against ~5.5 TB/s)
  const int nz = (coutP + 255) >> 8, tpc = a.L / T;
  const int m = blockIdx.x >> 3;
  const long tile = (long)(m / nz) * 8 + (blockIdx.x & 7);
  if (tile >= a.w_bstride) return;             // (w_bstride = B x tiles per cloud here; the grid is rounded up to 8 nz)
  const size_t b = (size_t)(tile / tpc);
  const int t0 = (int)(tile - (long)b * tpc) * T;
  const int chunk0 = (m % nz) * 256;
  const int chunkP = coutP - chunk0 < 256 ? coutP - chunk0 : 256;
  const int tid = threadIdx.x;
  const bool producer = tid >= 256;            // wave-uniform: waves 4..7
  if (tid < 256) {
    const int oc = chunk0 + tid;
    s_sc[tid] = (a.scale && oc < a.cout) ? a.scale[oc] : 1.0f;
    s_sh[tid] = (a.shift && oc < a.cout) ? a.shift[oc] : 0.0f;
  }
  const int cout = a.cout, act = a.act, L = a.L;
  const float *xb = a.x + b * a.cin * a.L;
  // NCH = cin / 128 is a template argument and the chunk loop is unrolled completely: with a loop the compiler's s_waitcnt
  // bookkeeping gives up at the back edge and makes a commit wait for the YOUNGER request too (vmcnt(14) .. vmcnt(0) with 32
  // loads outstanding) -- the producers then wait out the HBM latency every chunk with the consumers at the barrier
  constexpr int nch = NCH;
  const size_t wstep = (size_t)(coutP >> 5) * 128 * 4;               // floats of one 16-channel step of the image
  const float *wp = a.wp + (size_t)(chunk0 >> 5) * 128 * 4;
  const DenseNoHook nh;
  auto noepi = [](const f32x16 &, int, int, int, int) {};
  f32x16 carry[NR][TB];
  if (!producer) {
#pragma unroll
    for (int nr = 0; nr < NR; nr++)
#pragma unroll
      for (int j = 0; j < TB; j++)
#pragma unroll
        for (int r = 0; r < 16; r++) carry[nr][j][r] = 0.f;
  }
  // producer lane (uq, pgrp): tokens 4 uq .. + 3 of the eight channels of pieces pgrp and pgrp + 8 of a chunk: 32 lanes x 16
  // bytes = one 512-byte run per channel row, sixteen loads per chunk and lane.  TWO chunks are in flight per lane (a chunk
  // requested during chunk c is committed during chunk c + 2: with one, every barrier waited ~1 us for the fetch), and
  // their 2 x 16 pieces live IN the consumers' accumulator array -- piece (i, j) of set s = registers 4 (j & 3) .. + 3 of
  // carry[s][2 i + (j >> 2)]: the register file is allocated per kernel, not per role, and as arrays of their own the
  // pieces pushed the kernel 90-200 registers into scratch
  const int q = tid & 255, uq = q & 31, pgrp = q >> 5;
#define PCR_PV(s, i, j, k) carry[s][2 * (i) + ((j) >> 2)][4 * ((j) & 3) + (k)]
  // (buffer loads: one lane offset register + a scalar offset per piece; sixteen 64-bit lane addresses per request and set
  // were another 80 registers of scratch)
  const int dbg = a.x_pm;              // (PCR_DPC_DBG of a tuning build: 1 = no MFMAs, 2 = no fetches, 4 = no split / image stores,
                                       //  8 = every fetch from the first tile of the first cloud: L2 hits)
  // (the descriptor's base is the same in every lane; said explicitly, or each load sits in a waterfall loop)
  const size_t xbase = reinterpret_cast<size_t>((PCR_TUNING != 0 && (dbg & 8)) ? a.x : xb);
  const size_t xbase_u = (size_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)xbase) |
                         ((size_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(xbase >> 32)) << 32);
  const __amdgpu_buffer_rsrc_t rx =
      __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<float *>(xbase_u), 0, a.cin * L * 4, 0x00020000);
  const int vo_x = ((16 * (pgrp >> 1) + 4 * (pgrp & 1)) * L + ((PCR_TUNING != 0 && (dbg & 8)) ? 0 : t0) + 4 * uq) * 4;
  auto request = [&](auto stag, int c) __attribute__((always_inline)) {
    constexpr int S = decltype(stag)::value;
#pragma unroll
    for (int i = 0; i < 2; i++) {
#pragma unroll
      for (int j = 0; j < 8; j++) {
        // piece P = pgrp + 8 i: channel c KC + 16 (P >> 1) + bf_kpos(P & 1, j) = [lane part in vo_x] + c KC + 64 i + bf_kpos(0, j)
        const int so = (((PCR_TUNING != 0 && (dbg & 8)) ? 0 : c) * KC + 64 * i + bf_kpos(0, j)) * L * 4;
        const f32x4 t = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, vo_x, so, 0));
        PCR_PV(S, i, j, 0) = t[0];
        PCR_PV(S, i, j, 1) = t[1];
        PCR_PV(S, i, j, 2) = t[2];
        PCR_PV(S, i, j, 3) = t[3];
      }
    }
  };
  auto commit = [&](auto stag, float *img) __attribute__((always_inline)) {
    constexpr int S = decltype(stag)::value;
    bf16x8 *u = reinterpret_cast<bf16x8 *>(img);
#pragma unroll
    for (int i = 0; i < 2; i++) {
      const int P = pgrp + 8 * i;
#pragma unroll
      for (int k = 0; k < 4; k++) {
        float x8[8];
#pragma unroll
        for (int j = 0; j < 8; j++) x8[j] = PCR_PV(S, i, j, k);
        bf16x8 hi, lo;
        bf_split8(x8, hi, lo, NS == 3);
        u[(2 * P) * T + bf_tok_swz(4 * uq + k)] = hi;          // (swizzled token positions: tile_dense.h, BSWZ)
        if constexpr (NS == 3) u[(2 * P + 1) * T + bf_tok_swz(4 * uq + k)] = lo;
      }
    }
  };
  const std::integral_constant<int, 0> set0;
  const std::integral_constant<int, 1> set1;
  // The two roles are two SEPARATE straight-line instruction streams that meet the same number of s_barrier instructions
  // (the hardware counts arrivals, not program counters).  Written as one loop with `if (producer)` blocks the compiler's
  // s_waitcnt bookkeeping merges the consumers' pending weight loads with the producers' pending fetches at every join and
  // makes a commit wait for the YOUNGER request as well (vmcnt(14) .. vmcnt(0) with 32 loads outstanding; as separate
  // streams vmcnt(31) .. vmcnt(16): only the older set).  The barrier orders LDS traffic only.
#define PCR_LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
  // (the PCR_DPC_DBG ablation switches exist in tuning builds only: as run-time branches they are joins again, and the
  // production stream must be straight-line)
  auto off = [&](int bit) __attribute__((always_inline)) { return PCR_TUNING != 0 && (dbg & bit) != 0; };
  if (producer) {
    if (!off(2)) {
      request(set0, 0);
      request(set1, 1);
    }
#pragma unroll
    for (int c = 0; c < nch; c += 2) {
      // (chunk c - 2's readers passed the previous barrier: the buffer is free)
      if (!off(4)) commit(set0, img0);
      if (c + 2 < nch && !off(2)) request(set0, c + 2);
      PCR_LDS_BARRIER();               // chunk c is in LDS
      if (!off(4)) commit(set1, img0 + KC * T);
      if (c + 3 < nch && !off(2)) request(set1, c + 3);
      PCR_LDS_BARRIER();               // chunk c + 1 is in LDS
    }
  } else {
#pragma unroll
    for (int c = 0; c < nch; c++) {
      PCR_LDS_BARRIER();               // chunk c is in LDS; every consumer is done with chunk c - 1
      if (!off(1))
        tile_dense_bf_impl<TB, NR, 1, true, NS, decltype(noepi), bf_pf(NR), DenseNoHook, true, false, true, true, true>(
            img0 + (c & 1) * (KC * T), KC, wp + (size_t)c * (KC >> 4) * wstep, chunkP, false, noepi, nullptr, nh, coutP, nullptr,
            carry);
    }
  }
#undef PCR_LDS_BARRIER
  // ---- epilogue by ALL eight waves: the consumers normalise their tiles in registers (GN) and put them into a [cout][token]
  // staging tile (the images are dead), then every thread takes sixteen 16-byte pieces of it -- scale, shift, residual,
  // activation -- with all its residual loads in flight at once and whole 512-byte rows per 32 lanes.  (From the accumulator
  // layout the residual came as sixteen 4-byte loads per tile and lane, eight dependent HBM round trips per workgroup with
  // nothing else resident on the CU: 0.97 of the launch's 2.4 ms.)
  __syncthreads();                     // every consumer is done with the last image
  float *stage = smem;
  if (!producer) {
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, l31 = lane & 31, h = lane >> 5;
#pragma unroll
    for (int nr = 0; nr < NR; nr++) {
      const int cb = wave + 4 * nr;
      if (cb < (chunkP >> 5)) {
#pragma unroll
        for (int j = 0; j < TB; j++) {
          float v[16];
          if constexpr (GN) {
            switch (a.gn_gs) {
              case 4: gn_tile<4>(carry[nr][j], v); break;
              case 8: gn_tile<8>(carry[nr][j], v); break;
              case 16: gn_tile<16>(carry[nr][j], v); break;
              default: gn_tile<32>(carry[nr][j], v); break;
            }
          } else {
#pragma unroll
            for (int r = 0; r < 16; r++) v[r] = carry[nr][j][r];
          }
#pragma unroll
          for (int r = 0; r < 16; r++)
            stage[(cb * 32 + 8 * (r >> 2) + 4 * h + (r & 3)) * OP + j * 32 + l31] = v[r];
        }
      }
    }
  }
  __syncthreads();
  {
    float *out = a.y + b * a.cout * a.L;
    const float *res = (GN && a.res) ? a.res + b * a.cout * a.L : nullptr;
    const int qd = tid & 31, r0 = tid >> 5;          // rows r0 + 16 i, tokens 4 qd .. + 3; two rounds of eight rows
#pragma unroll 1
    for (int half = 0; half < 2; half++) {
      f32x4 rs[8];
#pragma unroll
      for (int i = 0; i < 8; i++) {
        const int row = r0 + 16 * (8 * half + i), oc = chunk0 + row;
        rs[i] = (res && oc < cout) ? *reinterpret_cast<const f32x4 *>(res + (size_t)oc * L + t0 + 4 * qd) : f32x4{0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int i = 0; i < 8; i++) {
        const int row = r0 + 16 * (8 * half + i), oc = chunk0 + row;
        if (row < chunkP && oc < cout) {
          const f32x4 v4 = *reinterpret_cast<const f32x4 *>(stage + row * OP + 4 * qd);
          const float sc = s_sc[row], sh = s_sh[row];
          f32x4 o4;
#pragma unroll
          for (int k = 0; k < 4; k++) {
            const float r = v4[k] * sc + sh + rs[i][k];
            o4[k] = act == 1 ? fmaxf(r, 0.f) : (act == 2 && r < 0.f) ? r * 0.2f : r;
          }
          *reinterpret_cast<f32x4 *>(out + (size_t)oc * L + t0 + 4 * qd) = o4;
        }
      }
    }
  }
#undef PCR_PV
}

// (B,C,L) -> out[c * B + b] = max over L   (channel-major with the clouds as tokens: (1,C,B))
__global__ __launch_bounds__(kThreads) void max_over_l_kernel(const float *__restrict__ x,
                                                              float *__restrict__ out, int B, int C, int L) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const size_t row = (size_t)blockIdx.x * (kThreads / 64) + wave;  // (b, c) row
  if (row >= (size_t)B * C) return;
  const float *p = x + row * L;
  float m = -INFINITY;
  for (int i = lane; i < L; i += 64) m = fmaxf(m, p[i]);
  m = wave_max(m);
  if (lane == 0) out[(row % C) * B + row / C] = m;
}

// GroupNorm over channel groups of every token of x (B,C,L), optional residual add and ReLU:
//   y = [relu]( GN(x) [+ res] ).  One thread per (group, token); two passes over C/groups channels.
struct GnArgs {
  const float *x, *gamma, *beta, *res;
  float *y;
  int C, L, groups, relu;
};

__global__ __launch_bounds__(kThreads) void groupnorm_kernel(GnArgs a) {
  const size_t b = blockIdx.z;
  const int g = blockIdx.y;
  const int t = blockIdx.x * kThreads + threadIdx.x;
  if (t >= a.L) return;
  const int gs = a.C / a.groups;
  const size_t base = (b * a.C + (size_t)g * gs) * a.L + t;
  float mean = 0.f;
  for (int i = 0; i < gs; i++) mean += a.x[base + (size_t)i * a.L];
  mean /= (float)gs;
  float var = 0.f;
  for (int i = 0; i < gs; i++) {
    const float d = a.x[base + (size_t)i * a.L] - mean;
    var += d * d;
  }
  var /= (float)gs;
  const float inv = 1.0f / sqrtf(var + 1e-5f);
  for (int i = 0; i < gs; i++) {
    const int c = g * gs + i;
    float v = (a.x[base + (size_t)i * a.L] - mean) * inv * a.gamma[c] + a.beta[c];
    if (a.res) v += a.res[base + (size_t)i * a.L];
    a.y[base + (size_t)i * a.L] = a.relu ? fmaxf(v, 0.f) : v;
  }
}

// t (1, k*k, B) (entry [c*k + c2][b] = T_b[c][c2], the fc3 output of an STN) -> per-cloud PACKED
// weight images of W_b[c2][c] = T_b[c][c2], so that dense(x_b, W_b) = (x_b^T T_b)^T = torch.bmm.
__global__ void pack_bmm_kernel(const float *__restrict__ t, float *__restrict__ wp, int B, int k) {
  const int CP = ceil8(k), OP = ceil32(k);
  const size_t per = (size_t)CP * OP;
  const size_t total = per * B;
  for (size_t e = blockIdx.x * (size_t)blockDim.x + threadIdx.x; e < total;
       e += (size_t)gridDim.x * blockDim.x) {
    const size_t b = e / per;
    const int r = (int)(e - b * per);
    const int j = r & 3, h = (r >> 2) & 1, o = (r >> 3) % OP, kb = (r >> 3) / OP;
    const int c = kb * 8 + j * 2 + h;  // input channel (contraction index)
    wp[e] = (o < k && c < k) ? t[((size_t)c * k + o) * B + b] : 0.f;
  }
}


// ---- point-major input, split bf16, wave-autonomous (round 4): y (B,cout,L) = act(scale (W x) + shift) for x (B,L,cin),
// cout = 32 NCB <= 128, cin a multiple of 64 (SSG's Conv1d 256 -> 64 after the last set-abstraction layer, whose output is
// point-major).  The f32 tile kernel spent 128 f32 MFMAs of 64 cycles per cout block and 32 tokens on it (as long as the
// tensor takes to stream from HBM) behind LDS tiles and barriers: 0.27 ms where the bytes take 0.13.  Here a token's row
// IS the B operand: lane (t, h) reads the two 16-byte pieces 16 s + 4 h, 16 s + 8 + 4 h of its token's row per 16-channel
// step -- the k order of the pcr_pack_weight_bf16x2_f32 image (bf_kpos) -- splits them into bf16 hi / lo and feeds three
// v_mfma_f32_32x32x16_bf16 per cout block; the weight image sits in LDS (staged once by a persistent workgroup of eight
// waves); accumulators leave through the usual scale / shift / activation as coalesced 128-byte channel rows.  No LDS
// tile, no barrier after the staging; the rows of the next four steps are requested before the current four are used.
constexpr int kDpsWaves = 8;
template <int NCB, bool LO>
__global__ __launch_bounds__(64 * kDpsWaves) __attribute__((amdgpu_waves_per_eu(2, 4)))
void dense_pm_stream_kernel(DenseArgs a, int B) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int KS = a.cin >> 4, WU = KS * NCB * 128;               // 16-channel steps; 16-byte units of the weight image
  bf16x8 *s_w = reinterpret_cast<bf16x8 *>(smem);
  float *s_sc = smem + 4 * WU, *s_sh = s_sc + 32 * NCB;
  const int tid = threadIdx.x, lane = tid & 63, j = lane & 31, h = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  {
    const f32x4 *src = reinterpret_cast<const f32x4 *>(a.wp);
    f32x4 *dst = reinterpret_cast<f32x4 *>(smem);
    for (int e = tid; e < WU; e += 64 * kDpsWaves) dst[e] = src[e];
    if (tid < 32 * NCB) {
      s_sc[tid] = (a.scale && tid < a.cout) ? a.scale[tid] : 1.0f;
      s_sh[tid] = (a.shift && tid < a.cout) ? a.shift[tid] : 0.0f;
    }
  }
  __syncthreads();
  const int L = a.L, cin = a.cin, cout = a.cout, act = a.act;
  const int nblk = (L + 31) >> 5, nitem = B * nblk;
  for (int it = blockIdx.x * kDpsWaves + wave; it < nitem; it += gridDim.x * kDpsWaves) {
    asm volatile("" ::: "memory");   // (the weight reads stay inside the item loop)
    const int b = it / nblk, blk = it - b * nblk;
    const int t = blk * 32 + j;
    const float *row = a.x + ((size_t)b * L + (t < L ? t : L - 1)) * cin + 4 * h;
    f32x16 acc[NCB];
#pragma unroll
    for (int cb = 0; cb < NCB; cb++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[cb][r] = 0.f;
    f32x4 xa[4][2], xn[4][2];
#pragma unroll
    for (int s2 = 0; s2 < 4; s2++) {
      xa[s2][0] = *reinterpret_cast<const f32x4 *>(row + 16 * s2);
      xa[s2][1] = *reinterpret_cast<const f32x4 *>(row + 16 * s2 + 8);
    }
    const bf16x8 *wb = s_w + lane;
    for (int q4 = 0; q4 < KS; q4 += 4) {
      if (q4 + 4 < KS) {
#pragma unroll
        for (int s2 = 0; s2 < 4; s2++) {
          xn[s2][0] = *reinterpret_cast<const f32x4 *>(row + 16 * (q4 + 4 + s2));
          xn[s2][1] = *reinterpret_cast<const f32x4 *>(row + 16 * (q4 + 4 + s2) + 8);
        }
      }
#pragma unroll
      for (int s2 = 0; s2 < 4; s2++) {
        float v[8];
#pragma unroll
        for (int e = 0; e < 4; e++) {
          v[e] = xa[s2][0][e];
          v[4 + e] = xa[s2][1][e];
        }
        bf16x8 bh, bl;
        bf_split8(v, bh, bl, LO);
        bf16x8 wh[NCB], wl[NCB];
#pragma unroll
        for (int cb = 0; cb < NCB; cb++) {
          wh[cb] = wb[(((q4 + s2) * NCB + cb) * 2) * 64];
          if constexpr (LO) wl[cb] = wb[(((q4 + s2) * NCB + cb) * 2 + 1) * 64];
        }
        if constexpr (LO) {
#pragma unroll
          for (int cb = 0; cb < NCB; cb++) acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh[cb], bl, acc[cb], 0, 0, 0);
#pragma unroll
          for (int cb = 0; cb < NCB; cb++) acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wl[cb], bh, acc[cb], 0, 0, 0);
        }
#pragma unroll
        for (int cb = 0; cb < NCB; cb++) acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh[cb], bh, acc[cb], 0, 0, 0);
      }
#pragma unroll
      for (int s2 = 0; s2 < 4; s2++) {
        xa[s2][0] = xn[s2][0];
        xa[s2][1] = xn[s2][1];
      }
    }
    if (t < L) {
      float *out = a.y + (size_t)b * cout * L + t;
#pragma unroll
      for (int cb = 0; cb < NCB; cb++)
#pragma unroll
        for (int g = 0; g < 4; g++) {
          const int o = cb * 32 + 8 * g + 4 * h;
          const f32x4 s4 = *reinterpret_cast<const f32x4 *>(s_sc + o), b4 = *reinterpret_cast<const f32x4 *>(s_sh + o);
#pragma unroll
          for (int q = 0; q < 4; q++) {
            if (o + q < cout) {
              const float r = acc[cb][4 * g + q] * s4[q] + b4[q];
              out[(size_t)(o + q) * L] = act == 1 ? fmaxf(r, 0.f) : (act == 2 && r < 0.f) ? r * 0.2f : r;
            }
          }
        }
    }
  }
}
}  // namespace

// ------------------------------------------------------------------------------ C ABI ----
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

// ---- bf16 images (tile_dense.h, tile_dense_bf_impl): [S = ceil16(cin)/16][OP/32][hi, lo][64 lanes][8 bf16] ----
static inline unsigned short pcr_bf16_rn(float f) {   // round to nearest even; weights are finite
  uint32_t u;
  memcpy(&u, &f, 4);
  if ((u & 0x7f800000u) == 0x7f800000u) return (unsigned short)(u >> 16);   // inf / nan: truncate
  u += 0x7fffu + ((u >> 16) & 1u);
  return (unsigned short)(u >> 16);
}
static inline float pcr_bf16_to_f32(unsigned short b) {
  const uint32_t u = (uint32_t)b << 16;
  float f;
  memcpy(&f, &u, 4);
  return f;
}

PCR_EXPORT long pcr_packed_weight_bf16_floats(int cout, int cin) {
  if (cout < 1 || cin < 1) return 0;
  return (long)((cin + 15) / 16) * (ceil32(cout) / 32) * 2 * 64 * 4;
}

PCR_EXPORT int pcr_pack_weight_bf16x2_f32(const float *w, int cout, int cin, float *packed) {
  if (!w || !packed || cout < 1 || cin < 1) return PCR_ERR_INVALID;
  const int S = (cin + 15) / 16, nCB = ceil32(cout) / 32;
  unsigned short *img = reinterpret_cast<unsigned short *>(packed);
  for (int s = 0; s < S; s++)
    for (int cb = 0; cb < nCB; cb++)
      for (int lane = 0; lane < 64; lane++)
        for (int j = 0; j < 8; j++) {
          const int o = cb * 32 + (lane & 31), k = 16 * s + bf_kpos(lane >> 5, j);   // (accumulator order: tile_dense.h)
          const float v = (o < cout && k < cin) ? w[(size_t)o * cin + k] : 0.f;
          const unsigned short hi = pcr_bf16_rn(v);
          const unsigned short lo = pcr_bf16_rn(v - pcr_bf16_to_f32(hi));
          const size_t base = (((size_t)s * nCB + cb) * 2) * 64;
          img[(base + lane) * 8 + j] = hi;
          img[(base + 64 + lane) * 8 + j] = lo;
        }
  return PCR_OK;
}

PCR_EXPORT long pcr_attn_kv_floats(int d) { return (long)d * d + d; }

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

static int dense_launch(const float *x, const float *wp, long w_bstride, const float *scale, const float *shift,
                        float *y, int B, int cin, int cout, int L, int act, pcr_stream_t stream, int x_pm = 0,
                        int gn_gs = 0, const float *res = nullptr) {
  if (!x || !wp || !y || B < 0 || cin < 1 || cout < 1 || L < 1) return PCR_ERR_INVALID;
  if (B == 0) return PCR_OK;
  if (B > 65535) return PCR_ERR_INVALID;
  DenseArgs a{x, wp, scale, shift, y, cin, cout, L, act, w_bstride, x_pm, gn_gs, res};
  if (!w_bstride && !x_pm && !gn_gs && dense_rw_ok(cin, cout, L) && !pcr_tune_str("PCR_DENSE_NO_RW"))
    return dense_rw_launch<false>(a, B, pcr_s(stream));
  const int cinP = ceil8(cin);
  if (cin >= 2 * kChunk && cin % kChunk == 0 && cout % 256 == 0 && !x_pm && !w_bstride && L > 32 &&
      !pcr_tune_str("PCR_DENSE_NO_CHUNK")) {
    const size_t ldsc = ((size_t)kChunk * 65 + 512) * sizeof(float);
    static bool okc = allow_big_lds(dense_kernel<2, true, true>) && allow_big_lds(dense_kernel<2, false, true>);
    (void)okc;
    const dim3 gc((L + 63) / 64, B, cout / 256);
    if (gn_gs) hipLaunchKernelGGL((dense_kernel<2, true, true>), gc, dim3(kThreads), ldsc, pcr_s(stream), a);
    else hipLaunchKernelGGL((dense_kernel<2, false, true>), gc, dim3(kThreads), ldsc, pcr_s(stream), a);
    PCR_CHECK_LAUNCH();
    return PCR_OK;
  }
  const int tb = ((size_t)cinP * 65 * 4 <= 72 * 1024 && L > 32) ? 2 : 1;
  size_t lds = ((size_t)cinP * (32 * tb + 1) + 512) * sizeof(float);
  if (lds > (size_t)kMaxDynLds) return PCR_ERR_INVALID;
  static bool ok = allow_big_lds(dense_kernel<1, false>) && allow_big_lds(dense_kernel<2, false>);
  (void)ok;
  dim3 g((L + 32 * tb - 1) / (32 * tb), B, (ceil32(cout) + 255) / 256);
  if (gn_gs) {
    static bool okg = allow_big_lds(dense_kernel<1, true>) && allow_big_lds(dense_kernel<2, true>);
    (void)okg;
    if (tb == 2) hipLaunchKernelGGL((dense_kernel<2, true>), g, dim3(kThreads), lds, pcr_s(stream), a);
    else hipLaunchKernelGGL((dense_kernel<1, true>), g, dim3(kThreads), lds, pcr_s(stream), a);
  } else if (tb == 2) hipLaunchKernelGGL((dense_kernel<2, false>), g, dim3(kThreads), lds, pcr_s(stream), a);
  else hipLaunchKernelGGL((dense_kernel<1, false>), g, dim3(kThreads), lds, pcr_s(stream), a);
  PCR_CHECK_LAUNCH();
  return PCR_OK;
}

// bf16 forms: cin a multiple of 64 (chunks of 256 / 128 / 64 channels), cout a multiple of 32, channel-major x, shared
// weights.  pcr_dense_prec_ok tells a caller whether a shape is covered (else it keeps the f32 launch).
PCR_EXPORT int pcr_dense_prec_ok(int cin, int cout, int L) {
  return cin >= 64 && cin % 64 == 0 && cout >= 32 && cout % 32 == 0 && L >= 1;
}

static int dense_bf_launch(const float *x, const float *wp_bf, const float *scale, const float *shift, float *y, int B,
                           int cin, int cout, int L, int act, int precision, pcr_stream_t stream, int gn_gs = 0,
                           const float *res = nullptr) {
  if (!x || !wp_bf || !y || B < 0 || (precision != PCR_PREC_BF16X3 && precision != PCR_PREC_BF16) ||
      !pcr_dense_prec_ok(cin, cout, L))
    return PCR_ERR_INVALID;
  if (B == 0) return PCR_OK;
  if (B > 65535) return PCR_ERR_INVALID;
  DenseArgs a{x, wp_bf, scale, shift, y, cin, cout, L, act, 0, 0, gn_gs, res};
  if ((cin == 256 || cin == 512 || cin == 1024) && L % kPcT == 0 && (reinterpret_cast<size_t>(x) & 15) == 0 && cout > 128 &&
      (size_t)cin * L * 4 < 0x7FFFFFFFull &&
      !pcr_tune_str("PCR_DENSE_NO_PC")) {
    const size_t ldsp = kPcLds;
    a.x_pm = pcr_tune_int("PCR_DPC_DBG");
    const long tiles = (long)B * (L / kPcT), nzw = (ceil32(cout) + 255) / 256;
    if (((tiles + 7) / 8) * 8 * nzw > 0x7FFFFFFFl) return PCR_ERR_INVALID;
    a.w_bstride = tiles;
#define PCR_DPC1(GNv, NSv, NCHv)                                                                  \
  do {                                                                                            \
    static bool ok = allow_big_lds(dense_bf_pc_kernel<GNv, NSv, NCHv>);                           \
    (void)ok;                                                                                     \
    hipLaunchKernelGGL((dense_bf_pc_kernel<GNv, NSv, NCHv>), dim3((unsigned)(((tiles + 7) / 8) * 8 * nzw)), dim3(kPcThreads), ldsp, pcr_s(stream), a); \
  } while (0)
#define PCR_DPC(GNv, NSv)                  \
  do {                                     \
    if (cin == 1024) PCR_DPC1(GNv, NSv, 8); \
    else if (cin == 512) PCR_DPC1(GNv, NSv, 4); \
    else PCR_DPC1(GNv, NSv, 2);            \
  } while (0)
    if (precision == PCR_PREC_BF16X3) { if (gn_gs) PCR_DPC(true, 3); else PCR_DPC(false, 3); }
    else { if (gn_gs) PCR_DPC(true, 1); else PCR_DPC(false, 1); }
#undef PCR_DPC1
#undef PCR_DPC
    PCR_CHECK_LAUNCH();
    return PCR_OK;
  }
  const int KC = cin % 256 == 0 ? 256 : (cin % 128 == 0 ? 128 : 64);
  const size_t lds = ((size_t)KC * 64 + 512) * sizeof(float);
  const dim3 g((L + 63) / 64, B, (cout + 255) / 256);
  const bool wide = cout > 128;      // more than four cout blocks in a workgroup's window: two rounds per wave
#define PCR_DBF(GNv, NSv, NRv)                                                                    \
  do {                                                                                            \
    static bool ok = allow_big_lds(dense_bf_kernel<GNv, NSv, NRv>);                               \
    (void)ok;                                                                                     \
    hipLaunchKernelGGL((dense_bf_kernel<GNv, NSv, NRv>), g, dim3(kThreads), lds, pcr_s(stream), a, KC); \
  } while (0)
  if (precision == PCR_PREC_BF16X3) {
    if (gn_gs) { if (wide) PCR_DBF(true, 3, 2); else PCR_DBF(true, 3, 1); }
    else { if (wide) PCR_DBF(false, 3, 2); else PCR_DBF(false, 3, 1); }
  } else {
    if (gn_gs) { if (wide) PCR_DBF(true, 1, 2); else PCR_DBF(true, 1, 1); }
    else { if (wide) PCR_DBF(false, 1, 2); else PCR_DBF(false, 1, 1); }
  }
#undef PCR_DBF
  PCR_CHECK_LAUNCH();
  return PCR_OK;
}

PCR_EXPORT int pcr_dense_prec_f32(const float *x, const float *wp_bf, const float *scale, const float *shift, float *y,
                                  int B, int cin, int cout, int L, int act, int precision, pcr_stream_t stream) {
  return dense_bf_launch(x, wp_bf, scale, shift, y, B, cin, cout, L, act, precision, stream);
}

// the point-major-input form of pcr_dense_prec_f32 (x is (B,L,cin)): shapes pcr_dense_xpm_prec_ok accepts
PCR_EXPORT int pcr_dense_xpm_prec_ok(int cin, int cout, int L) {
  if (!(cin >= 64 && cin % 64 == 0 && cout >= 1 && cout <= 128 && L >= 1)) return 0;
  const int ncb = (cout + 31) / 32;
  return (size_t)(cin / 16) * ncb * 2048 + 1024 <= (size_t)144 * 1024;   // the weight image must fit LDS
}

PCR_EXPORT int pcr_dense_xpm_prec_f32(const float *x, const float *wp_bf, const float *scale, const float *shift, float *y,
                                      int B, int cin, int cout, int L, int act, int precision, pcr_stream_t stream) {
  if (!x || !wp_bf || !y || B < 0 || (precision != PCR_PREC_BF16X3 && precision != PCR_PREC_BF16) ||
      !pcr_dense_xpm_prec_ok(cin, cout, L))
    return PCR_ERR_INVALID;
  if (B == 0) return PCR_OK;
  const long items = (long)B * ((L + 31) / 32);
  if (B > 65535 || items >= 0x7FFFFFFFl) return PCR_ERR_INVALID;
  pcr_note_arith(precision);
  DenseArgs a{x, wp_bf, scale, shift, y, cin, cout, L, act, 0, 1, 0, nullptr};
  const int ncb = (cout + 31) / 32;
  const size_t lds = (size_t)(cin / 16) * ncb * 2048 + (size_t)64 * ncb * sizeof(float);
  static const int ncu = [] {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess ||
        n < 1)
      n = 256;
    return n;
  }();
  long wgs = (items + kDpsWaves - 1) / kDpsWaves;
  const long resident = (long)ncu * (lds <= (size_t)78 * 1024 ? 2 : 1);
  if (wgs > resident) wgs = resident;
#define PCR_DPS(NCBv, LOv)                                                                                  \
  do {                                                                                                      \
    static bool ok = allow_big_lds(dense_pm_stream_kernel<NCBv, LOv>);                                      \
    (void)ok;                                                                                               \
    hipLaunchKernelGGL((dense_pm_stream_kernel<NCBv, LOv>), dim3((unsigned)wgs), dim3(64 * kDpsWaves), lds, \
                       pcr_s(stream), a, B);                                                                \
  } while (0)
  if (precision == PCR_PREC_BF16X3) {
    if (ncb == 1) PCR_DPS(1, true); else if (ncb == 2) PCR_DPS(2, true); else if (ncb == 3) PCR_DPS(3, true); else PCR_DPS(4, true);
  } else {
    if (ncb == 1) PCR_DPS(1, false); else if (ncb == 2) PCR_DPS(2, false); else if (ncb == 3) PCR_DPS(3, false); else PCR_DPS(4, false);
  }
#undef PCR_DPS
  PCR_CHECK_LAUNCH();
  return PCR_OK;
}

PCR_EXPORT int pcr_dense_gn_prec_f32(const float *x, const float *wp_bf, const float *gamma, const float *beta,
                                     const float *res, float *y, int B, int cin, int cout, int L, int groups, int relu,
                                     int precision, pcr_stream_t stream) {
  if (!gamma || !beta || groups < 1 || cout % groups) return PCR_ERR_INVALID;
  const int gs = cout / groups;
  if (gs != 4 && gs != 8 && gs != 16 && gs != 32) return PCR_ERR_INVALID;
  return dense_bf_launch(x, wp_bf, gamma, beta, y, B, cin, cout, L, relu ? 1 : 0, precision, stream, gs, res);
}

PCR_EXPORT int pcr_dense_gn_f32(const float *x, const float *wp, const float *gamma, const float *beta,
                                const float *res, float *y, int B, int cin, int cout, int L, int groups, int relu,
                                pcr_stream_t stream) {
  if (!gamma || !beta || groups < 1 || cout % groups) return PCR_ERR_INVALID;
  const int gs = cout / groups;
  if (gs != 4 && gs != 8 && gs != 16 && gs != 32) return PCR_ERR_INVALID;
  return dense_launch(x, wp, 0, gamma, beta, y, B, cin, cout, L, relu ? 1 : 0, stream, 0, gs, res);
}

PCR_EXPORT int pcr_dense_f32(const float *x, const float *wp, const float *scale, const float *shift,
                             float *y, int B, int cin, int cout, int L, int act, pcr_stream_t stream) {
  return dense_launch(x, wp, 0, scale, shift, y, B, cin, cout, L, act, stream);
}

PCR_EXPORT int pcr_dense_xpm_f32(const float *x, const float *wp, const float *scale, const float *shift,
                                 float *y, int B, int cin, int cout, int L, int act, pcr_stream_t stream) {
  return dense_launch(x, wp, 0, scale, shift, y, B, cin, cout, L, act, stream, 1);
}

PCR_EXPORT int pcr_dense_bmm_f32(const float *x, const float *wp_per_cloud, float *y, int B, int cin, int cout,
                                 int L, pcr_stream_t stream) {
  return dense_launch(x, wp_per_cloud, (long)ceil8(cin) * ceil32(cout), nullptr, nullptr, y, B, cin, cout, L, 0,
                      stream);
}

PCR_EXPORT int pcr_pack_bmm_f32(const float *t, float *wp_per_cloud, int B, int k, pcr_stream_t stream) {
  if (!t || !wp_per_cloud || B < 0 || k < 1) return PCR_ERR_INVALID;
  if (B == 0) return PCR_OK;
  size_t total = (size_t)ceil8(k) * ceil32(k) * B;
  size_t blocks = (total + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(pack_bmm_kernel, dim3((unsigned)blocks), dim3(256), 0, pcr_s(stream), t, wp_per_cloud, B, k);
  PCR_CHECK_LAUNCH();
  return PCR_OK;
}

PCR_EXPORT int pcr_channel_max_f32(const float *x, float *out, int B, int C, int L, int window, pcr_stream_t stream) {
  if (!x || !out || B < 0 || C < 1 || L < 1 || window < 1 || window > C) return PCR_ERR_INVALID;
  if (B == 0) return PCR_OK;
  if (B > 65535) return PCR_ERR_INVALID;
  hipLaunchKernelGGL(channel_max_kernel, dim3((L + kThreads - 1) / kThreads, B), dim3(kThreads), 0, pcr_s(stream), x, out,
                     C, L, window, C / window);
  PCR_CHECK_LAUNCH();
  return PCR_OK;
}

PCR_EXPORT int pcr_max_over_l_f32(const float *x, float *out, int B, int C, int L, pcr_stream_t stream) {
  if (!x || !out || B < 0 || C < 1 || L < 1) return PCR_ERR_INVALID;
  if (B == 0) return PCR_OK;
  size_t rows = (size_t)B * C;
  hipLaunchKernelGGL(max_over_l_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(kThreads), 0, pcr_s(stream), x, out,
                     B, C, L);
  PCR_CHECK_LAUNCH();
  return PCR_OK;
}

PCR_EXPORT int pcr_dense_max_ok(int cin, int cout, int L) {
  // (whole 256-cout chunks: eight cout blocks per workgroup, every wave owns ALL token blocks of its two)
  return cin >= 1 && cout >= 256 && cout % 256 == 0 && L >= 1 && (size_t)ceil8(cin) * 65 * 4 + 2048 <= (size_t)64 * 1024;
}

PCR_EXPORT int pcr_dense_max_f32(const float *x, const float *wp, const float *scale, const float *shift, float *out,
                                 int B, int cin, int cout, int L, int act, pcr_stream_t stream) {
  if (!x || !wp || !out || B < 0 || !pcr_dense_max_ok(cin, cout, L)) return PCR_ERR_INVALID;
  if (B == 0) return PCR_OK;
  if (B > 65535) return PCR_ERR_INVALID;
  DenseArgs a{x, wp, scale, shift, out, cin, cout, L, act, 0, 0, 0, nullptr};
  if (dense_rw_ok(cin, cout, L) && !pcr_tune_str("PCR_DENSE_NO_RW")) return dense_rw_launch<true>(a, B, pcr_s(stream));
  const size_t lds = ((size_t)ceil8(cin) * 65 + 512) * sizeof(float);
  static bool ok = allow_big_lds(dense_max_kernel);
  (void)ok;
  hipLaunchKernelGGL(dense_max_kernel, dim3(1, B, (ceil32(cout) + 255) / 256), dim3(kThreads), lds, pcr_s(stream), a, B);
  PCR_CHECK_LAUNCH();
  return PCR_OK;
}

PCR_EXPORT int pcr_groupnorm_f32(const float *x, const float *gamma, const float *beta, const float *res, float *y,
                                 int B, int C, int L, int groups, int relu, pcr_stream_t stream) {
  if (!x || !gamma || !beta || !y || B < 0 || C < 1 || L < 1 || groups < 1 || C % groups || groups > 65535)
    return PCR_ERR_INVALID;
  if (B == 0) return PCR_OK;
  if (B > 65535) return PCR_ERR_INVALID;
  GnArgs a{x, gamma, beta, res, y, C, L, groups, relu};
  hipLaunchKernelGGL(groupnorm_kernel, dim3((L + kThreads - 1) / kThreads, groups, B), dim3(kThreads), 0,
                     pcr_s(stream), a);
  PCR_CHECK_LAUNCH();
  return PCR_OK;
}
