// Training-mode kernels of the attention blocks and the match head that are not dense layers: the linear-attention
// core (forward + backward), LayerNorm / GroupNorm over the channels of every token (forward + backward) and the
// pair pooling of the match head (forward + backward).  The dense layers around them are pcr_tdense_{fwd,bwd}_f32.
//
// Reference graph: LinearAttention (models/pointnet2_utils.py:14-47), Self_Attention (:90-114), FP_SA (:407-437),
// corss_attention (models/attention.py:192-219), LinearRes (models/lanegcn_nets.py:228-241), get_pooled_feats
// 'both' (models/ReIDNet.py:529-532).  All reductions have a fixed order (no float atomics).
#include <type_traits>

#include "tile_dense.h"

namespace {

template <int CTRL>
__device__ __forceinline__ float dpp_f32(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
// sum over the 64 lanes of a wave (the same value in every lane): four DPP adds inside the 16-lane rows, then the four
// row totals through v_readlane -- no LDS traffic (a __shfl_xor butterfly is six ds_bpermute round trips)
__device__ __forceinline__ float wsum(float v) {
  v += dpp_f32<0xB1>(v);     // quad_perm [1,0,3,2]
  v += dpp_f32<0x4E>(v);     // quad_perm [2,3,0,1]
  v += dpp_f32<0x141>(v);    // row_half_mirror
  v += dpp_f32<0x140>(v);    // row_mirror
  const int b = __float_as_int(v);     // (v_readlane moves 32-bit patterns: the builtin is typed int)
  return (__int_as_float(__builtin_amdgcn_readlane(b, 0)) + __int_as_float(__builtin_amdgcn_readlane(b, 16))) +
         (__int_as_float(__builtin_amdgcn_readlane(b, 32)) + __int_as_float(__builtin_amdgcn_readlane(b, 48)));
}

// ------------------------------------------------------------------ token norm (LayerNorm / GroupNorm) ----
// x (B,C,L): every token is normalised over each of its G groups of C/G consecutive channels (LayerNorm: G = 1),
// y = (x - mean) rstd gamma + beta [+ res].  One thread per (token, group); mean / rstd are kept for the backward.
struct TNorm {
  const float *x, *gamma, *beta, *res;
  float *y, *mean, *rstd;     // mean / rstd (B,G,L)
  int C, L, G;
  float eps;
  int relu;                   // y = relu(...) (LinearRes, lanegcn_nets.py:233,240)
};

// (tokens are indexed flat over all clouds: ft = b * L + t, so a block is 256 real tokens whatever L is)
__global__ __launch_bounds__(64) void tnorm_fwd_kernel(TNorm a, int B) {
  const int g = blockIdx.y;
  const size_t ft = (size_t)blockIdx.x * 64 + threadIdx.x;   // one wave per block: small tensors need the workgroup count
  if (ft >= (size_t)B * a.L) return;
  const size_t b = ft / a.L;
  const int t = (int)(ft - b * a.L);
  const int gs = a.C / a.G;
  const size_t base = (b * a.C + (size_t)g * gs) * a.L + t;
  float m = 0.f;
#pragma unroll 8
  for (int i = 0; i < gs; i++) m += a.x[base + (size_t)i * a.L];
  m /= (float)gs;
  float v = 0.f;
#pragma unroll 8
  for (int i = 0; i < gs; i++) {
    const float d = a.x[base + (size_t)i * a.L] - m;
    v += d * d;
  }
  const float r = 1.0f / sqrtf(v / (float)gs + a.eps);
#pragma unroll 8
  for (int i = 0; i < gs; i++) {
    const int c = g * gs + i;
    float o = (a.x[base + (size_t)i * a.L] - m) * r * a.gamma[c] + a.beta[c];
    if (a.res) o += a.res[base + (size_t)i * a.L];
    a.y[base + (size_t)i * a.L] = a.relu ? fmaxf(o, 0.f) : o;
  }
  a.mean[(b * a.G + g) * a.L + t] = m;
  a.rstd[(b * a.G + g) * a.L + t] = r;
}

struct TNormBwd {
  const float *g, *x, *gamma, *mean, *rstd;
  float *dx;
  float *part;     // partials [gridDim.z * gridDim.x][2][C]: d gamma, d beta   (grid.y = groups)
  int C, L, G;
  const float *y;  // forward output when it went through a ReLU: the gradient is masked by y > 0 first
  float *dres;     // optional: the masked gradient (= gradient of the residual input)
};

__global__ __launch_bounds__(64) void tnorm_bwd_kernel(TNormBwd a, int B) {
  const int g = blockIdx.y;
  const size_t ft = (size_t)blockIdx.x * 64 + threadIdx.x;
  const int lane = threadIdx.x & 63;
  const bool ok = ft < (size_t)B * a.L;
  const size_t b = ok ? ft / a.L : 0;
  const int t = ok ? (int)(ft - b * a.L) : 0;
  const int gs = a.C / a.G;
  const size_t base = (b * a.C + (size_t)g * gs) * a.L + t;
  const float m = ok ? a.mean[(b * a.G + g) * a.L + t] : 0.f, r = ok ? a.rstd[(b * a.G + g) * a.L + t] : 0.f;
  float s1 = 0.f, s2 = 0.f;
  auto grad = [&](int i) {   // incoming gradient of channel i of this group, through the optional ReLU
    float go = ok ? a.g[base + (size_t)i * a.L] : 0.f;
    if (a.y && ok && !(a.y[base + (size_t)i * a.L] > 0.f)) go = 0.f;
    return go;
  };
#pragma unroll 8
  for (int i = 0; i < gs; i++) {
    const float gv = grad(i) * a.gamma[g * gs + i];
    const float xh = ok ? (a.x[base + (size_t)i * a.L] - m) * r : 0.f;
    s1 += gv;
    s2 += gv * xh;
  }
  const float inv = 1.0f / (float)gs;
  // d gamma / d beta: one partial row per wave = block (64 tokens), no barriers; rows [blockIdx.x][2][C]
  float *part = a.part + (size_t)blockIdx.x * 2 * a.C;
#pragma unroll 4
  for (int i = 0; i < gs; i++) {
    const int c = g * gs + i;
    const float go = grad(i);
    const float xh = ok ? (a.x[base + (size_t)i * a.L] - m) * r : 0.f;
    if (ok) a.dx[base + (size_t)i * a.L] = r * (go * a.gamma[c] - s1 * inv - xh * s2 * inv);
    if (ok && a.dres) a.dres[base + (size_t)i * a.L] = go;
    const float w1 = wsum(go * xh), w2 = wsum(go);
    if (lane == 0) {
      part[c] = w1;
      part[a.C + c] = w2;
    }
  }
}

// The same two kernels for wide groups (LayerNorm over 32 .. 256 channels): FOUR waves share the 64 tokens of a block,
// each holding a quarter of the group's channels in registers (NC per thread), and exchange their partial sums through
// LDS.  One pass over HBM instead of three (two in the backward), and four times the waves in flight: with one wave
// per 64 tokens a (512, 64, 128) tensor is 1024 waves on 1024 SIMDs, each waiting out its own load latency.
template <int NC>
__global__ __launch_bounds__(256) void tnorm_fwd4_kernel(TNorm a, int B) {
  __shared__ float ex[2][4][64];
  // gamma / beta of the group through LDS: read from global inside the store loops, every load waits behind the
  // previous store (the compiler cannot tell that y does not alias them)
  __shared__ float sgam[4 * NC], sbet[4 * NC];
  const int g = blockIdx.y, lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (threadIdx.x < 4 * NC) {
    sgam[threadIdx.x] = a.gamma[g * 4 * NC + threadIdx.x];
    sbet[threadIdx.x] = a.beta[g * 4 * NC + threadIdx.x];
  }
  const size_t ft = (size_t)blockIdx.x * 64 + lane;
  const bool ok = ft < (size_t)B * a.L;
  const size_t b = ok ? ft / a.L : 0;
  const int t = ok ? (int)(ft - b * a.L) : 0;
  const int gs = 4 * NC, c0 = g * gs + w * NC;
  const size_t base = (b * a.C + c0) * a.L + t;
  float xv[NC];
#pragma unroll
  for (int i = 0; i < NC; i++) xv[i] = ok ? a.x[base + (size_t)i * a.L] : 0.f;
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NC; i++) s += xv[i];
  ex[0][w][lane] = s;
  __syncthreads();
  const float m = ((ex[0][0][lane] + ex[0][1][lane]) + (ex[0][2][lane] + ex[0][3][lane])) / (float)gs;
  float v = 0.f;
#pragma unroll
  for (int i = 0; i < NC; i++) {
    const float d = xv[i] - m;
    v += d * d;
  }
  ex[1][w][lane] = v;
  __syncthreads();
  v = (ex[1][0][lane] + ex[1][1][lane]) + (ex[1][2][lane] + ex[1][3][lane]);
  const float r = 1.0f / sqrtf(v / (float)gs + a.eps);
  if (!ok) return;
  float rv[NC];
#pragma unroll
  for (int i = 0; i < NC; i++) rv[i] = a.res ? a.res[base + (size_t)i * a.L] : 0.f;
#pragma unroll
  for (int i = 0; i < NC; i++) {
    const float o = (xv[i] - m) * r * sgam[w * NC + i] + sbet[w * NC + i] + rv[i];
    a.y[base + (size_t)i * a.L] = a.relu ? fmaxf(o, 0.f) : o;
  }
  if (w == 0) {
    a.mean[(b * a.G + g) * a.L + t] = m;
    a.rstd[(b * a.G + g) * a.L + t] = r;
  }
}

template <int NC>
__global__ __launch_bounds__(256) void tnorm_bwd4_kernel(TNormBwd a, int B) {
  __shared__ float ex[2][4][64];
  // gamma / beta of the group through LDS: read from global inside the store loops, every load waits behind the
  // previous store (the compiler cannot tell that y does not alias them)
  __shared__ float sgam[4 * NC];
  const int g = blockIdx.y, lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (threadIdx.x < 4 * NC) {
    sgam[threadIdx.x] = a.gamma[g * 4 * NC + threadIdx.x];
  }
  const size_t ft = (size_t)blockIdx.x * 64 + lane;
  const bool ok = ft < (size_t)B * a.L;
  const size_t b = ok ? ft / a.L : 0;
  const int t = ok ? (int)(ft - b * a.L) : 0;
  const int gs = 4 * NC, c0 = g * gs + w * NC;
  const size_t base = (b * a.C + c0) * a.L + t;
  const float m = ok ? a.mean[(b * a.G + g) * a.L + t] : 0.f, r = ok ? a.rstd[(b * a.G + g) * a.L + t] : 0.f;
  float go[NC], xh[NC];
  const size_t base_c = ok ? base : 0;       // (clamped: the loads below are unconditional, so they all go out at once)
  if (a.y) {
#pragma unroll
    for (int i = 0; i < NC; i++) {
      const float gv = a.g[base_c + (size_t)i * a.L], yv = a.y[base_c + (size_t)i * a.L];
      xh[i] = a.x[base_c + (size_t)i * a.L];
      go[i] = (ok && yv > 0.f) ? gv : 0.f;
    }
  } else {
#pragma unroll
    for (int i = 0; i < NC; i++) {
      const float gv = a.g[base_c + (size_t)i * a.L];
      xh[i] = a.x[base_c + (size_t)i * a.L];
      go[i] = ok ? gv : 0.f;
    }
  }
#pragma unroll
  for (int i = 0; i < NC; i++) xh[i] = ok ? xh[i] : m;
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int i = 0; i < NC; i++) {
    xh[i] = (xh[i] - m) * r;
    const float gv = go[i] * sgam[w * NC + i];
    s1 += gv;
    s2 += gv * xh[i];
  }
  ex[0][w][lane] = s1;
  ex[1][w][lane] = s2;
  __syncthreads();
  s1 = (ex[0][0][lane] + ex[0][1][lane]) + (ex[0][2][lane] + ex[0][3][lane]);
  s2 = (ex[1][0][lane] + ex[1][1][lane]) + (ex[1][2][lane] + ex[1][3][lane]);
  const float inv = 1.0f / (float)gs;
  float *part = a.part + (size_t)blockIdx.x * 2 * a.C;
#pragma unroll
  for (int i = 0; i < NC; i++) {
    const int c = c0 + i;
    if (ok) a.dx[base + (size_t)i * a.L] = r * (go[i] * sgam[w * NC + i] - s1 * inv - xh[i] * s2 * inv);
    if (ok && a.dres) a.dres[base + (size_t)i * a.L] = go[i];
    const float w1 = wsum(go[i] * xh[i]), w2 = wsum(go[i]);
    if (lane == 0) {
      part[c] = w1;
      part[a.C + c] = w2;
    }
  }
}

// ------------------------------------------------------------------------- linear attention core ----
// Per (cloud, head): Q' = elu(q) + 1, K' = elu(k) + 1, V' = v / S, A = sum_s K'_s V'_s^T (dh x dh), ks = sum_s K'_s,
// out_l = (Q'_l^T A) S / (Q'_l . ks + eps)   (LinearAttention.forward, pointnet2_utils.py:26-47).
// q / k / v / out are (B, *, L) channel-major blocks addressed by a batch stride and a channel offset, so a fused
// (B,3d,L) projection needs no slicing copies.
struct LinAttn {
  const float *q, *k, *v;
  long q_bs, k_bs, v_bs;       // floats between clouds
  int Lq, Sk, d, H;
  float eps;
  float *out;                  // (B,d,Lq)
  float *A, *ks;               // saved for the backward: (B,H,dh,dh), (B,H,dh)
  // backward
  const float *dout;
  float *dq, *dk, *dv;
  long dq_bs, dk_bs, dv_bs;
  int B, kv_roll;              // keys / values (and their gradients) of cloud b live at cloud (b + kv_roll) % B
};

// the cloud whose keys / values query cloud b reads (a permutation of the batch: every dk / dv block has one writer)
__device__ __forceinline__ size_t linattn_kv_cloud(const LinAttn &a, size_t b) {
  return a.kv_roll ? (b + (size_t)a.kv_roll) % (size_t)a.B : b;
}

constexpr int kAT = 64;        // tokens per staged chunk

__device__ __forceinline__ float elu1f(float x) { return x > 0.f ? x + 1.0f : __expf(x); }

// KAT: tokens per staged chunk (64; 16 for the 128-wide heads of the mul = 2 configs' SA3 attention, whose dh x dh
// matrices leave that much LDS).  LDS is dynamic: the launch sites compute the byte counts.
template <int DH, int KAT = kAT>
__global__ __launch_bounds__(256) void linattn_fwd_kernel(LinAttn a) {
  constexpr int RP = KAT + 1, NP = (DH * DH + 255) / 256;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float *Kt = smem, *Vt = Kt + DH * RP, *Al = Vt + DH * RP, *ksl = Al + DH * (DH + 1), *zl = ksl + DH;
  const int tid = threadIdx.x;
  const int h = blockIdx.x;
  const size_t b = blockIdx.y;
  const float *q = a.q + b * a.q_bs + (size_t)h * DH * a.Lq;
  const size_t bk = linattn_kv_cloud(a, b);
  const float *k = a.k + bk * a.k_bs + (size_t)h * DH * a.Sk;
  const float *v = a.v + bk * a.v_bs + (size_t)h * DH * a.Sk;
  const float sk = (float)a.Sk;
  float acc[NP];
#pragma unroll
  for (int p = 0; p < NP; p++) acc[p] = 0.f;
  float ksum = 0.f;
  for (int s0 = 0; s0 < a.Sk; s0 += KAT) {
    const int ns = a.Sk - s0 < KAT ? a.Sk - s0 : KAT;
    if (s0) __syncthreads();
    for (int e = tid; e < DH * KAT; e += 256) {
      const int i = e / KAT, s = e - i * KAT;
      const bool ok = s < ns;
      Kt[i * RP + s] = ok ? elu1f(k[(size_t)i * a.Sk + s0 + s]) : 0.f;
      Vt[i * RP + s] = ok ? v[(size_t)i * a.Sk + s0 + s] / sk : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int p = 0; p < NP; p++) {
      const int e = tid + p * 256;
      if (e < DH * DH) {
        const int i = e / DH, j = e - i * DH;
        const float *kr = Kt + i * RP, *vr = Vt + j * RP;
        float s = 0.f;
#pragma unroll 8
        for (int t = 0; t < KAT; t++) s += kr[t] * vr[t];
        acc[p] += s;
      }
    }
    if (tid < DH) {
      const float *kr = Kt + tid * RP;
      float s = 0.f;
      for (int t = 0; t < KAT; t++) s += kr[t];
      ksum += s;
    }
  }
  float *Ag = a.A + (b * a.H + h) * DH * DH;
#pragma unroll
  for (int p = 0; p < NP; p++) {
    const int e = tid + p * 256;
    if (e < DH * DH) {
      Al[(e / DH) * (DH + 1) + e % DH] = acc[p];
      Ag[e] = acc[p];
    }
  }
  if (tid < DH) {
    ksl[tid] = ksum;
    a.ks[(b * a.H + h) * DH + tid] = ksum;
  }
  float *out = a.out + (b * a.d + (size_t)h * DH) * a.Lq;
  float *Qt = Kt;     // the key tiles are dead
  for (int l0 = 0; l0 < a.Lq; l0 += KAT) {
    const int nl = a.Lq - l0 < KAT ? a.Lq - l0 : KAT;
    __syncthreads();
    for (int e = tid; e < DH * KAT; e += 256) {
      const int i = e / KAT, l = e - i * KAT;
      Qt[i * RP + l] = l < nl ? elu1f(q[(size_t)i * a.Lq + l0 + l]) : 0.f;
    }
    __syncthreads();
    if (tid < KAT) {
      float z = 0.f;
      for (int i = 0; i < DH; i++) z += Qt[i * RP + tid] * ksl[i];
      zl[tid] = 1.0f / (z + a.eps);
    }
    __syncthreads();
    for (int e = tid; e < DH * KAT; e += 256) {
      const int vv = e / KAT, l = e - vv * KAT;
      float s = 0.f;
#pragma unroll 8
      for (int i = 0; i < DH; i++) s += Qt[i * RP + l] * Al[i * (DH + 1) + vv];
      if (l < nl) out[(size_t)vv * a.Lq + l0 + l] = s * zl[l] * sk;
    }
  }
}

template <int DH, int KAT = kAT>
__global__ __launch_bounds__(256) void linattn_bwd_kernel(LinAttn a) {
  constexpr int RP = KAT + 1, NP = (DH * DH + 255) / 256;
  extern __shared__ __attribute__((aligned(16))) float smem[];    // 83 KB at DH = 64: dynamic
  float *Qt = smem, *Gt = Qt + DH * RP, *Dn = Gt + DH * RP, *Al = Dn + DH * RP, *dAl = Al + DH * (DH + 1);
  float *ksl = dAl + DH * (DH + 1), *dksl = ksl + DH, *zl = dksl + DH, *ddl = zl + KAT;
  const int tid = threadIdx.x;
  const int h = blockIdx.x;
  const size_t b = blockIdx.y;
  const float *q = a.q + b * a.q_bs + (size_t)h * DH * a.Lq;
  const size_t bk = linattn_kv_cloud(a, b);
  const float *k = a.k + bk * a.k_bs + (size_t)h * DH * a.Sk;
  const float *v = a.v + bk * a.v_bs + (size_t)h * DH * a.Sk;
  const float *go = a.dout + (b * a.d + (size_t)h * DH) * a.Lq;
  float *dq = a.dq + b * a.dq_bs + (size_t)h * DH * a.Lq;
  float *dk = a.dk + bk * a.dk_bs + (size_t)h * DH * a.Sk;
  float *dv = a.dv + bk * a.dv_bs + (size_t)h * DH * a.Sk;
  const float sk = (float)a.Sk;
  const float *Ag = a.A + (b * a.H + h) * DH * DH;
  for (int e = tid; e < DH * DH; e += 256) Al[(e / DH) * (DH + 1) + e % DH] = Ag[e];
  if (tid < DH) ksl[tid] = a.ks[(b * a.H + h) * DH + tid];
  float dA[NP];
#pragma unroll
  for (int p = 0; p < NP; p++) dA[p] = 0.f;
  float dks = 0.f;
  // ---- pass 1 over the query tokens: dq, and the sums dA = sum_l Q'_l dnum_l^T, dks = sum_l dden_l Q'_l
  for (int l0 = 0; l0 < a.Lq; l0 += KAT) {
    const int nl = a.Lq - l0 < KAT ? a.Lq - l0 : KAT;
    __syncthreads();
    for (int e = tid; e < DH * KAT; e += 256) {
      const int i = e / KAT, l = e - i * KAT;
      const bool ok = l < nl;
      Qt[i * RP + l] = ok ? elu1f(q[(size_t)i * a.Lq + l0 + l]) : 0.f;
      Gt[i * RP + l] = ok ? go[(size_t)i * a.Lq + l0 + l] : 0.f;
    }
    __syncthreads();
    if (tid < KAT) {
      float z = 0.f;
      for (int i = 0; i < DH; i++) z += Qt[i * RP + tid] * ksl[i];
      zl[tid] = 1.0f / (z + a.eps);
    }
    __syncthreads();
    // num[v][l] = Q'_l . A[:,v];  dnum = dout z S;  dz_l = sum_v dout num S  (accumulated per token below)
    for (int e = tid; e < DH * KAT; e += 256) {
      const int vv = e / KAT, l = e - vv * KAT;
      float s = 0.f;
#pragma unroll 8
      for (int i = 0; i < DH; i++) s += Qt[i * RP + l] * Al[i * (DH + 1) + vv];
      const float gv = Gt[vv * RP + l];
      Dn[vv * RP + l] = gv * zl[l] * sk;
      Gt[vv * RP + l] = gv * s * sk;          // dout * num * S (summed over v next)
    }
    __syncthreads();
    if (tid < KAT) {
      float dz = 0.f;
      for (int vv = 0; vv < DH; vv++) dz += Gt[vv * RP + tid];
      ddl[tid] = -zl[tid] * zl[tid] * dz;      // gradient of the denominator Q'.ks + eps
    }
    __syncthreads();
    for (int e = tid; e < DH * KAT; e += 256) {
      const int i = e / KAT, l = e - i * KAT;
      float s = ddl[l] * ksl[i];
#pragma unroll 8
      for (int vv = 0; vv < DH; vv++) s += Dn[vv * RP + l] * Al[i * (DH + 1) + vv];
      const float qp = Qt[i * RP + l];
      if (l < nl) dq[(size_t)i * a.Lq + l0 + l] = s * (qp > 1.0f ? 1.0f : qp);   // elu'(q) = 1 (q > 0) | exp(q) = Q'
    }
#pragma unroll
    for (int p = 0; p < NP; p++) {
      const int e = tid + p * 256;
      if (e < DH * DH) {
        const int i = e / DH, vv = e - i * DH;
        const float *qr = Qt + i * RP, *dr = Dn + vv * RP;
        float s = 0.f;
#pragma unroll 8
        for (int t = 0; t < KAT; t++) s += qr[t] * dr[t];
        dA[p] += s;
      }
    }
    if (tid < DH) {
      const float *qr = Qt + tid * RP;
      float s = 0.f;
      for (int t = 0; t < KAT; t++) s += ddl[t] * qr[t];
      dks += s;
    }
  }
  __syncthreads();
#pragma unroll
  for (int p = 0; p < NP; p++) {
    const int e = tid + p * 256;
    if (e < DH * DH) dAl[(e / DH) * (DH + 1) + e % DH] = dA[p];
  }
  if (tid < DH) dksl[tid] = dks;
  // ---- pass 2 over the key tokens: dK' = dA V' + dks, dV' = dA^T K'
  float *Kt = Qt, *Vt = Gt;
  for (int s0 = 0; s0 < a.Sk; s0 += KAT) {
    const int ns = a.Sk - s0 < KAT ? a.Sk - s0 : KAT;
    __syncthreads();
    for (int e = tid; e < DH * KAT; e += 256) {
      const int i = e / KAT, s = e - i * KAT;
      const bool ok = s < ns;
      Kt[i * RP + s] = ok ? elu1f(k[(size_t)i * a.Sk + s0 + s]) : 0.f;
      Vt[i * RP + s] = ok ? v[(size_t)i * a.Sk + s0 + s] / sk : 0.f;
    }
    __syncthreads();
    for (int e = tid; e < DH * KAT; e += 256) {
      const int i = e / KAT, s = e - i * KAT;
      float dkp = dksl[i], dvp = 0.f;
#pragma unroll 8
      for (int j = 0; j < DH; j++) {
        dkp += dAl[i * (DH + 1) + j] * Vt[j * RP + s];
        dvp += Kt[j * RP + s] * dAl[j * (DH + 1) + i];
      }
      if (s < ns) {
        const float kp = Kt[i * RP + s];
        dk[(size_t)i * a.Sk + s0 + s] = dkp * (kp > 1.0f ? 1.0f : kp);
        dv[(size_t)i * a.Sk + s0 + s] = dvp / sk;
      }
    }
  }
}

// ---- the same two kernels with the small matrix products on the matrix cores (head widths 32 and 64) ----
// Every product of the attention core is a (DH x DH) by (DH x 64 tokens) GEMM or a contraction over the 64 tokens of a
// staged chunk.  As scalar loops each multiply-add costs two LDS reads (no register reuse): 0.40 ms for the ONE
// dh = 64 launch of a training step (B = 512, two heads, L = 32).  v_mfma_f32_32x32x2_f32 reads the same LDS tiles
// once per 32 x 32 output tile: acc[m][n] += sum_k A(m,k) B(n,k), one tile per wave.
template <class AF, class BF>
__device__ __forceinline__ f32x16 mm_tile(int K, AF af, BF bf, f32x16 acc) {
  const int l31 = threadIdx.x & 31, h = (threadIdx.x >> 5) & 1;
#pragma unroll 8
  for (int ks = 0; ks < K / 2; ks++)
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af(l31, 2 * ks + h), bf(l31, 2 * ks + h), acc, 0, 0, 0);
  return acc;
}
template <class F>
__device__ __forceinline__ void mm_each(const f32x16 &acc, F f) {   // f(m, n, value) over the lane's 16 results
  const int l31 = threadIdx.x & 31, h = (threadIdx.x >> 5) & 1;
#pragma unroll
  for (int r = 0; r < 16; r++) f((r & 3) + 8 * (r >> 2) + 4 * h, l31, acc[r]);
}
__device__ __forceinline__ f32x16 zero16() {
  f32x16 z;
#pragma unroll
  for (int r = 0; r < 16; r++) z[r] = 0.f;
  return z;
}

// Two (DH x 64-token) tiles into LDS, every global load of a thread in flight before its first LDS write (a load ->
// store loop serialises on the load latency: 8 .. 16 round trips per tile and tensor).
template <int DH, class FA, class FB>
__device__ __forceinline__ void stage_pair(float *dA, float *dB, const float *sA, const float *sB, int Lrow, int l0,
                                           int nl, FA fa, FB fb) {
  constexpr int RP = kAT + 1, NE = DH * kAT / 256;
  float va[NE], vb[NE];
#pragma unroll
  for (int u = 0; u < NE; u++) {
    const int e = threadIdx.x + u * 256, i = e / kAT, l = e % kAT;
    const bool ok = l < nl;
    const size_t o = (size_t)i * Lrow + l0 + (ok ? l : 0);
    va[u] = sA[o];
    vb[u] = sB ? sB[o] : 0.f;
  }
#pragma unroll
  for (int u = 0; u < NE; u++) {
    const int e = threadIdx.x + u * 256, i = e / kAT, l = e % kAT;
    const bool ok = l < nl;
    dA[i * RP + l] = ok ? fa(va[u]) : 0.f;
    if (dB) dB[i * RP + l] = ok ? fb(vb[u]) : 0.f;
  }
}

template <int DH>
__global__ __launch_bounds__(256) void linattn_fwd_mfma_kernel(LinAttn a) {
  constexpr int RP = kAT + 1, AP = DH + 1, MT = DH / 32, NT = kAT / 32;
  __shared__ float Kt[DH * RP], Vt[DH * RP], Al[DH * AP], ksl[DH], zl[kAT];
  const int tid = threadIdx.x, wave = tid >> 6;
  const int h = blockIdx.x;
  const size_t b = blockIdx.y;
  const float *q = a.q + b * a.q_bs + (size_t)h * DH * a.Lq;
  const size_t bk = linattn_kv_cloud(a, b);
  const float *k = a.k + bk * a.k_bs + (size_t)h * DH * a.Sk;
  const float *v = a.v + bk * a.v_bs + (size_t)h * DH * a.Sk;
  const float sk = (float)a.Sk;
  f32x16 accA = zero16();              // tile `wave` of A (MT x MT tiles; DH = 32: wave 0 only)
  const bool a_owner = wave < MT * MT;
  const int amt = wave / MT, ant = wave % MT;
  float ksum = 0.f;
  for (int s0 = 0; s0 < a.Sk; s0 += kAT) {
    const int ns = a.Sk - s0 < kAT ? a.Sk - s0 : kAT;
    if (s0) __syncthreads();
    stage_pair<DH>(Kt, Vt, k, v, a.Sk, s0, ns, [](float x) { return elu1f(x); }, [&](float x) { return x / sk; });
    __syncthreads();
    const int KT = ns > 32 ? 64 : 32;      // (the padding tokens are zeros)
    if (a_owner)
      accA = mm_tile(KT, [&](int m, int kk) { return Kt[(amt * 32 + m) * RP + kk]; },
                     [&](int n, int kk) { return Vt[(ant * 32 + n) * RP + kk]; }, accA);
    if (tid < DH) {
      const float *kr = Kt + tid * RP;
      float s = 0.f;
      for (int t = 0; t < kAT; t++) s += kr[t];
      ksum += s;
    }
  }
  float *Ag = a.A + (b * a.H + h) * DH * DH;
  if (a_owner)
    mm_each(accA, [&](int m, int n, float val) {
      const int i = amt * 32 + m, j = ant * 32 + n;
      Al[i * AP + j] = val;
      Ag[i * DH + j] = val;
    });
  if (tid < DH) {
    ksl[tid] = ksum;
    a.ks[(b * a.H + h) * DH + tid] = ksum;
  }
  float *out = a.out + (b * a.d + (size_t)h * DH) * a.Lq;
  float *Qt = Kt;     // the key tiles are dead
  for (int l0 = 0; l0 < a.Lq; l0 += kAT) {
    const int nl = a.Lq - l0 < kAT ? a.Lq - l0 : kAT;
    __syncthreads();
    stage_pair<DH>(Qt, (float *)nullptr, q, (const float *)nullptr, a.Lq, l0, nl, [](float x) { return elu1f(x); },
                   [](float x) { return x; });
    __syncthreads();
    if (tid < kAT) {
      float z = 0.f;
      for (int i = 0; i < DH; i++) z += Qt[i * RP + tid] * ksl[i];
      zl[tid] = 1.0f / (z + a.eps);
    }
    __syncthreads();
    for (int tile = wave; tile < MT * NT; tile += 4) {      // out[v][l] = sum_i A[i][v] Q'[i][l]
      const int mt = tile / NT, nt = tile % NT;
      if (nt * 32 >= nl) continue;
      const f32x16 acc = mm_tile(DH, [&](int m, int kk) { return Al[kk * AP + mt * 32 + m]; },
                                 [&](int n, int kk) { return Qt[kk * RP + nt * 32 + n]; }, zero16());
      mm_each(acc, [&](int m, int n, float val) {
        const int vv = mt * 32 + m, l = nt * 32 + n;
        if (l < nl) out[(size_t)vv * a.Lq + l0 + l] = val * zl[l] * sk;
      });
    }
  }
}

template <int DH>
__global__ __launch_bounds__(256) void linattn_bwd_mfma_kernel(LinAttn a) {
  constexpr int RP = kAT + 1, AP = DH + 1, MT = DH / 32, NT = kAT / 32;
  extern __shared__ __attribute__((aligned(16))) float smem[];    // 83 KB at DH = 64: dynamic
  float *Qt = smem, *Gt = Qt + DH * RP, *Dn = Gt + DH * RP, *Al = Dn + DH * RP, *dAl = Al + DH * AP;
  float *ksl = dAl + DH * AP, *dksl = ksl + DH, *zl = dksl + DH, *ddl = zl + kAT;
  const int tid = threadIdx.x, wave = tid >> 6;
  const int h = blockIdx.x;
  const size_t b = blockIdx.y;
  const float *q = a.q + b * a.q_bs + (size_t)h * DH * a.Lq;
  const size_t bk = linattn_kv_cloud(a, b);
  const float *k = a.k + bk * a.k_bs + (size_t)h * DH * a.Sk;
  const float *v = a.v + bk * a.v_bs + (size_t)h * DH * a.Sk;
  const float *go = a.dout + (b * a.d + (size_t)h * DH) * a.Lq;
  float *dq = a.dq + b * a.dq_bs + (size_t)h * DH * a.Lq;
  float *dk = a.dk + bk * a.dk_bs + (size_t)h * DH * a.Sk;
  float *dv = a.dv + bk * a.dv_bs + (size_t)h * DH * a.Sk;
  const float sk = (float)a.Sk;
  const float *Ag = a.A + (b * a.H + h) * DH * DH;
  {
    constexpr int NA = DH * DH / 256;
    float va[NA];
#pragma unroll
    for (int u = 0; u < NA; u++) va[u] = Ag[tid + u * 256];
#pragma unroll
    for (int u = 0; u < NA; u++) {
      const int e = tid + u * 256;
      Al[(e / DH) * AP + e % DH] = va[u];
    }
  }
  if (tid < DH) ksl[tid] = a.ks[(b * a.H + h) * DH + tid];
  f32x16 accdA = zero16();             // tile `wave` of dA = sum_l Q'_l dnum_l^T
  const bool a_owner = wave < MT * MT;
  const int amt = wave / MT, ant = wave % MT;
  float dks = 0.f;
  // ---- pass 1 over the query tokens: dq, and the sums dA, dks = sum_l dden_l Q'_l
  for (int l0 = 0; l0 < a.Lq; l0 += kAT) {
    const int nl = a.Lq - l0 < kAT ? a.Lq - l0 : kAT;
    const int KT = nl > 32 ? 64 : 32;
    __syncthreads();
    stage_pair<DH>(Qt, Gt, q, go, a.Lq, l0, nl, [](float x) { return elu1f(x); }, [](float x) { return x; });
    __syncthreads();
    if (tid < kAT) {
      float z = 0.f;
      for (int i = 0; i < DH; i++) z += Qt[i * RP + tid] * ksl[i];
      zl[tid] = 1.0f / (z + a.eps);
    }
    __syncthreads();
    // num[v][l] = Q'_l . A[:,v];  dnum = dout z S;  dout * num * S is summed over v into dz below
    for (int tile = wave; tile < MT * NT; tile += 4) {
      const int mt = tile / NT, nt = tile % NT;
      if (nt * 32 >= KT) continue;
      const f32x16 acc = mm_tile(DH, [&](int m, int kk) { return Al[kk * AP + mt * 32 + m]; },
                                 [&](int n, int kk) { return Qt[kk * RP + nt * 32 + n]; }, zero16());
      mm_each(acc, [&](int m, int n, float num) {
        const int vv = mt * 32 + m, l = nt * 32 + n;
        const float gv = Gt[vv * RP + l];
        Dn[vv * RP + l] = gv * zl[l] * sk;
        Gt[vv * RP + l] = gv * num * sk;
      });
    }
    __syncthreads();
    if (tid < kAT) {
      float dz = 0.f;
      if (tid < KT)
        for (int vv = 0; vv < DH; vv++) dz += Gt[vv * RP + tid];
      ddl[tid] = -zl[tid] * zl[tid] * dz;      // gradient of the denominator Q'.ks + eps (zero on padding tokens)
    }
    __syncthreads();
    for (int tile = wave; tile < MT * NT; tile += 4) {      // dq[i][l] = ddl[l] ks[i] + sum_v A[i][v] dnum[v][l]
      const int mt = tile / NT, nt = tile % NT;
      if (nt * 32 >= nl) continue;
      const f32x16 acc = mm_tile(DH, [&](int m, int kk) { return Al[(mt * 32 + m) * AP + kk]; },
                                 [&](int n, int kk) { return Dn[kk * RP + nt * 32 + n]; }, zero16());
      mm_each(acc, [&](int m, int n, float val) {
        const int i = mt * 32 + m, l = nt * 32 + n;
        const float qp = Qt[i * RP + l];
        if (l < nl) dq[(size_t)i * a.Lq + l0 + l] = (val + ddl[l] * ksl[i]) * (qp > 1.0f ? 1.0f : qp);   // elu' = 1 | Q'
      });
    }
    if (a_owner)
      accdA = mm_tile(KT, [&](int m, int kk) { return Qt[(amt * 32 + m) * RP + kk]; },
                      [&](int n, int kk) { return Dn[(ant * 32 + n) * RP + kk]; }, accdA);
    if (tid < DH) {
      const float *qr = Qt + tid * RP;
      float s = 0.f;
      for (int t = 0; t < KT; t++) s += ddl[t] * qr[t];
      dks += s;
    }
  }
  __syncthreads();
  if (a_owner)
    mm_each(accdA, [&](int m, int n, float val) { dAl[(amt * 32 + m) * AP + ant * 32 + n] = val; });
  if (tid < DH) dksl[tid] = dks;
  // ---- pass 2 over the key tokens: dK' = dA V' + dks, dV' = dA^T K'
  float *Kt = Qt, *Vt = Gt;
  for (int s0 = 0; s0 < a.Sk; s0 += kAT) {
    const int ns = a.Sk - s0 < kAT ? a.Sk - s0 : kAT;
    __syncthreads();
    stage_pair<DH>(Kt, Vt, k, v, a.Sk, s0, ns, [](float x) { return elu1f(x); }, [&](float x) { return x / sk; });
    __syncthreads();
    for (int tile = wave; tile < MT * NT; tile += 4) {
      const int mt = tile / NT, nt = tile % NT;
      if (nt * 32 >= ns) continue;
      const f32x16 ak = mm_tile(DH, [&](int m, int kk) { return dAl[(mt * 32 + m) * AP + kk]; },
                                [&](int n, int kk) { return Vt[kk * RP + nt * 32 + n]; }, zero16());
      const f32x16 av = mm_tile(DH, [&](int m, int kk) { return dAl[kk * AP + mt * 32 + m]; },
                                [&](int n, int kk) { return Kt[kk * RP + nt * 32 + n]; }, zero16());
      const int l31 = threadIdx.x & 31, hh = (threadIdx.x >> 5) & 1, s = nt * 32 + l31;
#pragma unroll
      for (int r = 0; r < 16; r++) {
        const int i = mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
        if (s < ns) {
          const float kp = Kt[i * RP + s];
          dk[(size_t)i * a.Sk + s0 + s] = (ak[r] + dksl[i]) * (kp > 1.0f ? 1.0f : kp);
          dv[(size_t)i * a.Sk + s0 + s] = av[r] / sk;
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------ pair pooling ----
// o (2P,C,L): clouds p and p + P are pair p.  pooled (P,2C) = [max over the 2L points, mean over the 2L points];
// arg (P,C) = position of the maximum in the point-concatenated pair (0 .. 2L-1, first maximum).
__global__ __launch_bounds__(256) void pool_pair_fwd_kernel(const float *__restrict__ o, float *__restrict__ pooled,
                                                            int *__restrict__ arg, int P, int C, int L) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const size_t p = blockIdx.x;
  for (int c = wave; c < C; c += 4) {
    const float *o1 = o + (p * C + c) * L, *o2 = o + ((p + P) * C + c) * L;
    float mx = -INFINITY, sm = 0.f;
    int am = 0;
    for (int i = lane; i < 2 * L; i += 64) {
      const float x = i < L ? o1[i] : o2[i - L];
      if (x > mx) {
        mx = x;
        am = i;
      }
      sm += x;
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
      const float ox = __shfl_xor(mx, m, 64);
      const int oa = __shfl_xor(am, m, 64);
      if (ox > mx || (ox == mx && oa < am)) {
        mx = ox;
        am = oa;
      }
    }
    sm = wsum(sm);
    if (lane == 0) {
      pooled[p * 2 * C + c] = mx;
      pooled[p * 2 * C + C + c] = sm / (float)(2 * L);
      arg[p * C + c] = am;
    }
  }
}

__global__ __launch_bounds__(256) void pool_pair_bwd_kernel(const float *__restrict__ g, const int *__restrict__ arg,
                                                            float *__restrict__ dout, int P, int C, int L) {
  const size_t bc = blockIdx.x;                 // (cloud, channel) row of the (2P,C,L) gradient
  const size_t b = bc / C;
  const int c = (int)(bc - b * C);
  const size_t p = b < (size_t)P ? b : b - P;
  const int half = b < (size_t)P ? 0 : 1;
  const float gm = g[p * 2 * C + c], ga = g[p * 2 * C + C + c] / (float)(2 * L);
  const int am = arg[p * C + c] - half * L;
  for (int l = threadIdx.x; l < L; l += 256) dout[bc * L + l] = ga + (l == am ? gm : 0.f);
}

// ---- the same pooling for ONE tensor (match types that pool a single branch: xcorr-baseline, ReIDNet.py:258-264 with
// get_pooled_feats 'both' :529-532), and the channel-window max of pool_type 'max' (:145, 526-528) ----
// o (P,C,L) -> pooled (P,2C) = [max over L, mean over L]; arg (P,C)
__global__ __launch_bounds__(256) void pool_both_fwd_kernel(const float *__restrict__ o, float *__restrict__ pooled,
                                                            int *__restrict__ arg, int C, int L) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const size_t p = blockIdx.x;
  for (int c = wave; c < C; c += 4) {
    const float *row = o + (p * C + c) * L;
    float mx = -INFINITY, sm = 0.f;
    int am = 0;
    for (int i = lane; i < L; i += 64) {
      const float x = row[i];
      if (x > mx) {
        mx = x;
        am = i;
      }
      sm += x;
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
      const float ox = __shfl_xor(mx, m, 64);
      const int oa = __shfl_xor(am, m, 64);
      if (ox > mx || (ox == mx && oa < am)) {
        mx = ox;
        am = oa;
      }
    }
    sm = wsum(sm);
    if (lane == 0) {
      pooled[p * 2 * C + c] = mx;
      pooled[p * 2 * C + C + c] = sm / (float)L;
      arg[p * C + c] = am;
    }
  }
}

__global__ __launch_bounds__(256) void pool_both_bwd_kernel(const float *__restrict__ g, const int *__restrict__ arg,
                                                            float *__restrict__ dout, int C, int L) {
  const size_t pc = blockIdx.x;                 // (cloud, channel) row
  const size_t p = pc / C;
  const int c = (int)(pc - p * C);
  const float gm = g[p * 2 * C + c], ga = g[p * 2 * C + C + c] / (float)L;
  const int am = arg[pc];
  for (int l = threadIdx.x; l < L; l += 256) dout[pc * L + l] = ga + (l == am ? gm : 0.f);
}

// x (B,C,L) -> y (B,C/W,L) = max over windows of W consecutive channels of every point (nn.MaxPool1d(W) on the
// permuted (B,L,C) tensor); arg (B,C/W,L) = the winning channel (first maximum)
__global__ __launch_bounds__(256) void channel_max_idx_kernel(const float *__restrict__ x, float *__restrict__ y,
                                                              int *__restrict__ arg, int C, int L, int W, size_t total) {
  const int G = C / W;
  for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
    const int l = (int)(e % L);
    const size_t bg = e / L;
    const int g = (int)(bg % G);
    const size_t b = bg / G;
    const float *src = x + (b * C + (size_t)g * W) * L + l;
    float mx = src[0];
    int am = 0;
    for (int w = 1; w < W; w++) {
      const float v = src[(size_t)w * L];
      if (v > mx) {
        mx = v;
        am = w;
      }
    }
    y[e] = mx;
    arg[e] = g * W + am;
  }
}

__global__ __launch_bounds__(256) void channel_max_bwd_kernel(const float *__restrict__ g, const int *__restrict__ arg,
                                                              float *__restrict__ dx, int C, int L, int W, size_t total) {
  const int G = C / W;
  for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {   // over (B,C,L)
    const int l = (int)(e % L);
    const size_t bc = e / L;
    const int c = (int)(bc % C);
    const size_t b = bc / C;
    const size_t o = (b * G + c / W) * L + l;
    dx[e] = arg[o] == c ? g[o] : 0.f;
  }
}

template <class F>
void dh_dispatch(int dh, F f) {
  switch (dh) {
    case 16: f(std::integral_constant<int, 16>()); break;
    case 32: f(std::integral_constant<int, 32>()); break;
    case 128: f(std::integral_constant<int, 128>()); break;
    default: f(std::integral_constant<int, 64>()); break;
  }
}

}  // namespace

PCR_EXPORT int pcr_tnorm_fwd_f32(const float *x, const float *gamma, const float *beta, const float *res, float *y,
                                 float *mean, float *rstd, int B, int C, int L, int G, float eps, int relu,
                                 pcr_stream_t stream) {
  if (!x || !gamma || !beta || !y || !mean || !rstd || B < 0 || C < 1 || L < 1 || G < 1 || C % G) return PCR_ERR_INVALID;
  if (B == 0) return PCR_OK;
  if (G > 65535) return PCR_ERR_INVALID;
  TNorm a{x, gamma, beta, res, y, mean, rstd, C, L, G, eps, relu};
  const unsigned nb = (unsigned)(((size_t)B * L + 63) / 64);
  const int gs = C / G;
  switch ((gs & 3) == 0 ? gs / 4 : 0) {
    case 8: hipLaunchKernelGGL(tnorm_fwd4_kernel<8>, dim3(nb, G), dim3(256), 0, pcr_s(stream), a, B); break;
    case 16: hipLaunchKernelGGL(tnorm_fwd4_kernel<16>, dim3(nb, G), dim3(256), 0, pcr_s(stream), a, B); break;
    case 32: hipLaunchKernelGGL(tnorm_fwd4_kernel<32>, dim3(nb, G), dim3(256), 0, pcr_s(stream), a, B); break;
    case 64: hipLaunchKernelGGL(tnorm_fwd4_kernel<64>, dim3(nb, G), dim3(256), 0, pcr_s(stream), a, B); break;
    default: hipLaunchKernelGGL(tnorm_fwd_kernel, dim3(nb, G), dim3(64), 0, pcr_s(stream), a, B);
  }
  PCR_CHECK_LAUNCH();
  return PCR_OK;
}

PCR_EXPORT int pcr_tnorm_bwd_f32(const float *g, const float *x, const float *gamma, const float *mean, const float *rstd,
                                 const float *y_relu, float *dx, float *dres, float *part, int B, int C, int L, int G,
                                 pcr_stream_t stream) {
  if (!g || !x || !gamma || !mean || !rstd || !dx || !part || B < 0 || C < 1 || L < 1 || G < 1 || C % G)
    return PCR_ERR_INVALID;
  if (B == 0) return PCR_OK;
  if (G > 65535) return PCR_ERR_INVALID;
  TNormBwd a{g, x, gamma, mean, rstd, dx, part, C, L, G, y_relu, dres};
  const unsigned nb = (unsigned)(((size_t)B * L + 63) / 64);
  const int gs = C / G;
  switch ((gs & 3) == 0 ? gs / 4 : 0) {
    case 8: hipLaunchKernelGGL(tnorm_bwd4_kernel<8>, dim3(nb, G), dim3(256), 0, pcr_s(stream), a, B); break;
    case 16: hipLaunchKernelGGL(tnorm_bwd4_kernel<16>, dim3(nb, G), dim3(256), 0, pcr_s(stream), a, B); break;
    case 32: hipLaunchKernelGGL(tnorm_bwd4_kernel<32>, dim3(nb, G), dim3(256), 0, pcr_s(stream), a, B); break;
    case 64: hipLaunchKernelGGL(tnorm_bwd4_kernel<64>, dim3(nb, G), dim3(256), 0, pcr_s(stream), a, B); break;
    default: hipLaunchKernelGGL(tnorm_bwd_kernel, dim3(nb, G), dim3(64), 0, pcr_s(stream), a, B);
  }
  PCR_CHECK_LAUNCH();
  return PCR_OK;
}

static int linattn_check(const pcr_linattn *p) {
  if (!p || !p->q || !p->k || !p->v || !p->A || !p->ks || p->B < 0 || p->Lq < 1 || p->Sk < 1 || p->H < 1 || p->d % p->H)
    return 1;
  const int dh = p->d / p->H;
  return !(dh == 16 || dh == 32 || dh == 64 || dh == 128);
}

static LinAttn linattn_args(const pcr_linattn *p) {
  LinAttn a;
  a.q = p->q; a.k = p->k; a.v = p->v; a.q_bs = p->q_bs; a.k_bs = p->k_bs; a.v_bs = p->v_bs;
  a.Lq = p->Lq; a.Sk = p->Sk; a.d = p->d; a.H = p->H; a.eps = p->eps;
  a.out = p->out; a.A = p->A; a.ks = p->ks; a.dout = p->dout;
  a.dq = p->dq; a.dk = p->dk; a.dv = p->dv; a.dq_bs = p->dq_bs; a.dk_bs = p->dk_bs; a.dv_bs = p->dv_bs;
  a.B = p->B; a.kv_roll = p->B > 0 ? ((p->kv_roll % p->B) + p->B) % p->B : 0;
  return a;
}

PCR_EXPORT int pcr_linattn_fwd_f32(const pcr_linattn *p, pcr_stream_t stream) {
  if (linattn_check(p) || !p->out) return PCR_ERR_INVALID;
  if (p->B == 0) return PCR_OK;
  if (p->B > 65535) return PCR_ERR_INVALID;
  const LinAttn a = linattn_args(p);
  dh_dispatch(p->d / p->H, [&](auto tag) {
    constexpr int DH = decltype(tag)::value;
    if constexpr (DH == 32 || DH == 64) {
      hipLaunchKernelGGL(linattn_fwd_mfma_kernel<DH>, dim3(p->H, p->B), dim3(256), 0, pcr_s(stream), a);
    } else {
      // scalar form: 16-wide heads, and the 128-wide ones (d_model 256: the mul = 2 configs) with 16-token chunks
      constexpr int KAT = DH == 128 ? 16 : kAT;
      const size_t lds = (2 * (size_t)DH * (KAT + 1) + (size_t)DH * (DH + 1) + DH + KAT) * sizeof(float);
      static bool ok = allow_big_lds(linattn_fwd_kernel<DH, KAT>);
      (void)ok;
      hipLaunchKernelGGL((linattn_fwd_kernel<DH, KAT>), dim3(p->H, p->B), dim3(256), lds, pcr_s(stream), a);
    }
  });
  PCR_CHECK_LAUNCH();
  return PCR_OK;
}

PCR_EXPORT int pcr_linattn_bwd_f32(const pcr_linattn *p, pcr_stream_t stream) {
  if (linattn_check(p) || !p->dout || !p->dq || !p->dk || !p->dv) return PCR_ERR_INVALID;
  if (p->B == 0) return PCR_OK;
  if (p->B > 65535) return PCR_ERR_INVALID;
  const LinAttn a = linattn_args(p);
  dh_dispatch(p->d / p->H, [&](auto tag) {
    constexpr int DH = decltype(tag)::value;
    constexpr int KAT = DH == 128 ? 16 : kAT;
    const size_t lds = (3 * (size_t)DH * (KAT + 1) + 2 * (size_t)DH * (DH + 1) + 2 * DH + 2 * KAT) * sizeof(float);
    if constexpr (DH == 32 || DH == 64) {
      static bool ok = allow_big_lds(linattn_bwd_mfma_kernel<DH>);
      (void)ok;
      hipLaunchKernelGGL(linattn_bwd_mfma_kernel<DH>, dim3(p->H, p->B), dim3(256), lds, pcr_s(stream), a);
    } else {
      static bool ok = allow_big_lds(linattn_bwd_kernel<DH, KAT>);
      (void)ok;
      hipLaunchKernelGGL((linattn_bwd_kernel<DH, KAT>), dim3(p->H, p->B), dim3(256), lds, pcr_s(stream), a);
    }
  });
  PCR_CHECK_LAUNCH();
  return PCR_OK;
}

PCR_EXPORT int pcr_pool_pair_fwd_f32(const float *o, float *pooled, int *arg, int P, int C, int L, pcr_stream_t stream) {
  if (!o || !pooled || !arg || P < 0 || C < 1 || L < 1) return PCR_ERR_INVALID;
  if (P == 0) return PCR_OK;
  hipLaunchKernelGGL(pool_pair_fwd_kernel, dim3(P), dim3(256), 0, pcr_s(stream), o, pooled, arg, P, C, L);
  PCR_CHECK_LAUNCH();
  return PCR_OK;
}

PCR_EXPORT int pcr_pool_pair_bwd_f32(const float *g, const int *arg, float *dout, int P, int C, int L,
                                     pcr_stream_t stream) {
  if (!g || !arg || !dout || P < 0 || C < 1 || L < 1) return PCR_ERR_INVALID;
  if (P == 0) return PCR_OK;
  hipLaunchKernelGGL(pool_pair_bwd_kernel, dim3(2 * P * C), dim3(256), 0, pcr_s(stream), g, arg, dout, P, C, L);
  PCR_CHECK_LAUNCH();
  return PCR_OK;
}

PCR_EXPORT int pcr_pool_both_fwd_f32(const float *o, float *pooled, int *arg, int P, int C, int L, pcr_stream_t stream) {
  if (!o || !pooled || !arg || P < 0 || C < 1 || L < 1) return PCR_ERR_INVALID;
  if (P == 0) return PCR_OK;
  hipLaunchKernelGGL(pool_both_fwd_kernel, dim3(P), dim3(256), 0, pcr_s(stream), o, pooled, arg, C, L);
  PCR_CHECK_LAUNCH();
  return PCR_OK;
}

PCR_EXPORT int pcr_pool_both_bwd_f32(const float *g, const int *arg, float *dout, int P, int C, int L, pcr_stream_t stream) {
  if (!g || !arg || !dout || P < 0 || C < 1 || L < 1) return PCR_ERR_INVALID;
  if (P == 0) return PCR_OK;
  hipLaunchKernelGGL(pool_both_bwd_kernel, dim3(P * C), dim3(256), 0, pcr_s(stream), g, arg, dout, C, L);
  PCR_CHECK_LAUNCH();
  return PCR_OK;
}

PCR_EXPORT int pcr_channel_max_fwd_f32(const float *x, float *y, int *arg, int B, int C, int L, int W, pcr_stream_t stream) {
  if (!x || !y || !arg || B < 0 || C < 1 || L < 1 || W < 1 || C % W) return PCR_ERR_INVALID;
  if (B == 0) return PCR_OK;
  const size_t total = (size_t)B * (C / W) * L;
  size_t blocks = (total + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(channel_max_idx_kernel, dim3((unsigned)blocks), dim3(256), 0, pcr_s(stream), x, y, arg, C, L, W, total);
  PCR_CHECK_LAUNCH();
  return PCR_OK;
}

PCR_EXPORT int pcr_channel_max_bwd_f32(const float *g, const int *arg, float *dx, int B, int C, int L, int W,
                                       pcr_stream_t stream) {
  if (!g || !arg || !dx || B < 0 || C < 1 || L < 1 || W < 1 || C % W) return PCR_ERR_INVALID;
  if (B == 0) return PCR_OK;
  const size_t total = (size_t)B * C * L;
  size_t blocks = (total + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(channel_max_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, pcr_s(stream), g, arg, dx, C, L, W, total);
  PCR_CHECK_LAUNCH();
  return PCR_OK;
}
