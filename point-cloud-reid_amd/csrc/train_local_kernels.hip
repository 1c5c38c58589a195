// Training-mode core of local_self_attention (reference mmdet3d/models/attention.py:262-296 with
// LinearAttention, pointnet2_utils.py:26-47, on one query token per point over its K feature-space neighbours):
//   msg_i = sum_j a_ij v_j / (sum_j a_ij + eps),   a_ij = < elu(q_i) + 1, elu(k_j) + 1 >  per head, j in N(i).
// The reference projects the gathered (B N, K, C) neighbour tensor; key and value of an edge depend on the neighbour
// POINT only (fea_knn + pos(xyz_knn) = the neighbour's own feature + position code), so q | k | v are per-point rows of
// ONE fused projection qkv (B, 3C, N) -- channel-major, as every tensor of the training graph -- and the K-fold gather
// happens here.  Forward: one wave per point, lane = channel.  Backward: the same wave recomputes a_ij and writes
//   dq_i                      (dense, (B,C,N)),
//   the per-EDGE gradients    e[0:C]  = d k_j contribution  = da_ij Q_i elu'(k_j),
//                             e[C:2C] = d v_j contribution  = a_ij / z_i g_i                  ((B,2C,N,K)),
// which pcr_group_bwd_f32 (owner-computes scatter, no float atomics) folds back onto the points: the gradient is
// bit-reproducible.  C <= 64, head width a power of two.
#include "pcr_common.h"

namespace {

constexpr int kLT = 256;   // four waves = four points per workgroup

template <int CTRL>
__device__ __forceinline__ float la_dpp(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
// sum over the dh lanes of a head (dh a power of two <= 64; heads are aligned lane groups)
__device__ __forceinline__ float la_head_sum(float a, int dh) {
  if (dh > 1) a += la_dpp<0xB1>(a);
  if (dh > 2) a += la_dpp<0x4E>(a);
  if (dh > 4) a += la_dpp<0x141>(a);
  if (dh > 8) a += la_dpp<0x140>(a);
  if (dh > 16) a += __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(a), 0x401F));   // lane ^ 16
  if (dh > 32) a += __shfl_xor(a, 32, 64);
  return a;
}
__device__ __forceinline__ float la_elu1(float x) { return x > 0.f ? x + 1.0f : __expf(x); }

struct LAArgs {
  const float *qkv;   // (B,3C,N)
  const int *idx;     // (B,N,K)
  const float *g;     // (B,C,N) gradient of msg (backward)
  float *msg;         // (B,C,N) (forward)
  float *dq;          // (B,C,N) with batch stride dq_bs (backward)
  float *edge;        // (B,2C,N,K) (backward)
  long dq_bs;
  int N, C, K, dh;
  float eps;
};

template <bool BWD>
__global__ __launch_bounds__(kLT) void local_attn_cm_kernel(LAArgs a) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int N = a.N, C = a.C, K = a.K, dh = a.dh;
  const size_t b = blockIdx.y;
  const int i = blockIdx.x * (kLT / 64) + wave;
  if (i >= N) return;
  const bool live = lane < C;
  const int c = live ? lane : C - 1;
  const float *q = a.qkv + (b * 3 * C + c) * N, *kk = q + (size_t)C * N, *vv = q + (size_t)2 * C * N;
  const int *nb = a.idx + (b * N + i) * K;
  const float qraw = q[i], Q = la_elu1(qraw);
  float num = 0.f, den = 0.f;
  for (int k = 0; k < K; k++) {
    const int j = nb[k];
    float av = Q * la_elu1(kk[j]);
    av = la_head_sum(live ? av : 0.f, dh);
    den += av;
    num = fmaf(av, vv[j], num);
  }
  const float z = den + a.eps, m = num / z;
  if constexpr (!BWD) {
    if (live) a.msg[(b * C + c) * N + i] = m;
    return;
  } else {
    const float gc = live ? a.g[(b * C + c) * N + i] : 0.f;
    const float gm = la_head_sum(gc * m, dh);
    float dQ = 0.f;
    float *ek = a.edge + ((b * 2 * C + c) * N + i) * K, *ev = ek + (size_t)C * N * K;
    for (int k = 0; k < K; k++) {
      const int j = nb[k];
      const float kraw = kk[j], Kf = la_elu1(kraw), v = vv[j];
      const float av = la_head_sum(live ? Q * Kf : 0.f, dh);
      const float gv = la_head_sum(gc * v, dh);
      const float da = (gv - gm) / z;
      dQ = fmaf(da, Kf, dQ);
      if (live) {
        ek[k] = da * Q * (kraw > 0.f ? 1.f : Kf);
        ev[k] = av / z * gc;
      }
    }
    if (live) a.dq[b * a.dq_bs + (size_t)c * N + i] = dQ * (qraw > 0.f ? 1.f : Q);
  }
}

int la_check(const float *qkv, const int *idx, int B, int N, int C, int K, int nhead) {
  if (!qkv || !idx || B < 0 || N < 1 || K < 1 || C < 1 || C > 64 || nhead < 1 || C % nhead || B > 65535) return 1;
  const int dh = C / nhead;
  return (dh & (dh - 1)) != 0;
}

}  // namespace

PCR_EXPORT int pcr_local_attn_train_fwd_f32(const float *qkv, const int *idx, float *msg, int B, int N, int C, int K,
                                            int nhead, float eps, pcr_stream_t stream) {
  if (la_check(qkv, idx, B, N, C, K, nhead) || !msg) return PCR_ERR_INVALID;
  if (B == 0) return PCR_OK;
  LAArgs a{qkv, idx, nullptr, msg, nullptr, nullptr, 0, N, C, K, C / nhead, eps};
  hipLaunchKernelGGL(local_attn_cm_kernel<false>, dim3((N + kLT / 64 - 1) / (kLT / 64), B), dim3(kLT), 0, pcr_s(stream), a);
  PCR_CHECK_LAUNCH();
  return PCR_OK;
}

PCR_EXPORT int pcr_local_attn_train_bwd_f32(const float *qkv, const int *idx, const float *g, float *dq, long dq_bstride,
                                            float *edge, int B, int N, int C, int K, int nhead, float eps,
                                            pcr_stream_t stream) {
  if (la_check(qkv, idx, B, N, C, K, nhead) || !g || !dq || !edge || dq_bstride < (long)C * N) return PCR_ERR_INVALID;
  if (B == 0) return PCR_OK;
  LAArgs a{qkv, idx, g, nullptr, dq, edge, dq_bstride, N, C, K, C / nhead, eps};
  hipLaunchKernelGGL(local_attn_cm_kernel<true>, dim3((N + kLT / 64 - 1) / (kLT / 64), B), dim3(kLT), 0, pcr_s(stream), a);
  PCR_CHECK_LAUNCH();
  return PCR_OK;
}
