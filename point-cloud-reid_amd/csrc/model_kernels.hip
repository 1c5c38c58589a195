// Pooling + match head, generic dense layer, host-side weight packing.
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

