// Training-mode pieces of the PointNet encoder that are not dense layers (reference: mmdet3d/models/pointnet.py:27-45,
// 67-85, 103-127): BatchNorm1d in batch-statistics mode as a stand-alone layer (forward + backward) and the per-cloud
// input / feature transforms (torch.bmm with a learned k x k matrix per cloud, forward + backward).  The Point-Transformer
// path folds its BatchNorms into the neighbouring train-dense launches (train_kernels.hip); PointNet's sit between a
// 1024-row layer, a max over the points and fully connected layers on (clouds, channels) rows, so here they are plain
// HBM-bound passes over (B, C, L) channel-major tensors.  Every reduction is two-stage in a fixed order (no atomics).
#include "pcr_common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
  return v;
}

// part [G][2][CP]: slice g of the clouds, channel c: (sum y, sum y^2)  -- or, with g2 given, (sum g', sum g' y) where
// g' = g2 act'(scale y + shift), act' = 1 (z > 0) | slope (ReLU: slope 0, LeakyReLU(0.2): 0.2; `relu` = 0: g' = g2):
// the two sums of the BatchNorm backward
__global__ __launch_bounds__(256) void bn_sums_kernel(const float *__restrict__ y, const float *__restrict__ g2,
                                                      const float *__restrict__ scale, const float *__restrict__ shift,
                                                      int relu, float slope, const float *__restrict__ centre,
                                                      int centre_first, float *__restrict__ part, int B, int C, int CP,
                                                      int L) {
  __shared__ float red[2][4];
  const int c = blockIdx.x, g = blockIdx.y, G = gridDim.y;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float sc = g2 && relu ? scale[c] : 0.f, sh = g2 && relu ? shift[c] : 0.f;
  // sums of y - m0 (m0 near the mean): E[y^2] - mean^2 and sum g' y - mean sum g' lose digits to cancellation otherwise
  const float m0 = centre ? centre[c] : (centre_first ? y[(size_t)c * L] : 0.f);
  float s0 = 0.f, s1 = 0.f;
  for (int b = g; b < B; b += G) {
    const float *row = y + ((size_t)b * C + c) * L;
    const float *grow = g2 ? g2 + ((size_t)b * C + c) * L : nullptr;
    for (int l = threadIdx.x; l < L; l += 256) {
      const float v = row[l] - m0;
      if (grow) {
        float gv = grow[l];
        if (relu && !(sc * row[l] + sh > 0.f)) gv *= slope;
        s0 += gv;
        s1 += gv * v;
      } else {
        s0 += v;
        s1 += v * v;
      }
    }
  }
  s0 = wave_sum(s0);
  s1 = wave_sum(s1);
  if (lane == 0) {
    red[0][wave] = s0;
    red[1][wave] = s1;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    part[((size_t)g * 2) * CP + c] = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
    part[((size_t)g * 2 + 1) * CP + c] = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
  }
}

// forward: z = [relu](scale y + shift);  backward (g given): dy = a0 g [scale y + shift > 0] + a1 (y - centre) + a2
// (centre = the batch mean, a1 = kb, a2 = -ka dbeta / R: the centred form of ka g + kb y + kc, without its cancellation)
__global__ __launch_bounds__(256) void bn_affine_kernel(const float *__restrict__ y, const float *__restrict__ g,
                                                        const float *__restrict__ a0, const float *__restrict__ a1,
                                                        const float *__restrict__ a2, const float *__restrict__ scale,
                                                        const float *__restrict__ shift, const float *__restrict__ centre,
                                                        int relu, float slope, float *__restrict__ out, int C, int L,
                                                        size_t total) {
  for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
    const int c = (int)((e / L) % C);
    const float v = y[e];
    if (!g) {
      const float z = a0[c] * v + a1[c];
      out[e] = (relu && !(z > 0.f)) ? slope * z : z;
    } else {
      float gv = g[e];
      if (relu && !(scale[c] * v + shift[c] > 0.f)) gv *= slope;
      out[e] = a0[c] * gv + a1[c] * (v - (centre ? centre[c] : 0.f)) + a2[c];
    }
  }
}

// y[b][j][n] = sum_i T[b][i][j] x[b][i][n]   (torch.bmm(x^T, T)^T, pointnet.py:109-111, 117-119); TR: T used transposed
// (the backward's dx[b][i][n] = sum_j T[b][i][j] dy[b][j][n])
__global__ __launch_bounds__(256) void bmm_apply_kernel(const float *__restrict__ x, const float *__restrict__ T,
                                                        float *__restrict__ y, int k, int N, int TR) {
  extern __shared__ float sT[];   // [k][k + 1]
  const size_t b = blockIdx.y;
  const float *Tb = T + b * k * k;
  for (int e = threadIdx.x; e < k * k; e += 256) {
    const int i = e / k, j = e - i * k;
    sT[(TR ? j : i) * (k + 1) + (TR ? i : j)] = Tb[e];     // sT[i'][j'] = coefficient of input row i' in output row j'
  }
  __syncthreads();
  const int n = blockIdx.x * 64 + (threadIdx.x & 63);
  const float *xb = x + b * k * N;
  float *yb = y + b * k * N;
  if (n < N)
    for (int j = threadIdx.x >> 6; j < k; j += 4) {
      float s = 0.f;
      for (int i = 0; i < k; i++) s += sT[i * (k + 1) + j] * xb[(size_t)i * N + n];
      yb[(size_t)j * N + n] = s;
    }
}

// dT[b][i][j] = sum_n x[b][i][n] dy[b][j][n]: one wave per (i, j) element, lanes over n
__global__ __launch_bounds__(256) void bmm_dt_kernel(const float *__restrict__ x, const float *__restrict__ dy,
                                                     float *__restrict__ dT, int k, int N) {
  const size_t b = blockIdx.y;
  const int lane = threadIdx.x & 63;
  const int e = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (e >= k * k) return;
  const int i = e / k, j = e - i * k;
  const float *xr = x + (b * k + i) * N, *dr = dy + (b * k + j) * N;
  float s = 0.f;
  for (int n = lane; n < N; n += 64) s += xr[n] * dr[n];
  s = wave_sum(s);
  if (lane == 0) dT[b * k * k + e] = s;
}

// EdgeConv tail in training mode (dgcnn_orig.py:127-143: BatchNorm2d over all edges, LeakyReLU, max over the k
// neighbours) on the materialised pre-activation y (B,C,N k): pooled = act(max_k (scale y + shift)) -- the activation
// is increasing, so it commutes with the max -- arg = the first k attaining it, yraw = y there.
__global__ __launch_bounds__(256) void edge_pool_fwd_kernel(const float *__restrict__ y, const float *__restrict__ scale,
                                                            const float *__restrict__ shift, float slope,
                                                            float *__restrict__ pooled, int *__restrict__ arg,
                                                            float *__restrict__ yraw, int C, int N, int K, size_t total) {
  for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {   // over (B,C,N)
    const int c = (int)((e / N) % C);
    const float *row = y + e * K;
    const float sc = scale[c], sh = shift[c];
    float best = sc * row[0] + sh, raw = row[0];
    int bk = 0;
    for (int k = 1; k < K; k++) {
      const float v = sc * row[k] + sh;
      if (v > best) {
        best = v;
        bk = k;
        raw = row[k];
      }
    }
    pooled[e] = best > 0.f ? best : slope * best;
    arg[e] = bk;
    yraw[e] = raw;
  }
}

// the pooled gradient routed back to its edge: g[b][c][n K + k] = gp[b][c][n] act'(pooled) if k == arg else 0
__global__ __launch_bounds__(256) void edge_pool_route_kernel(const float *__restrict__ gp, const float *__restrict__ pooled,
                                                              const int *__restrict__ arg, float slope,
                                                              float *__restrict__ g, int K, size_t total) {
  for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {   // over (B,C,N K)
    const size_t o = e / K;
    const int k = (int)(e - o * K);
    g[e] = arg[o] == k ? gp[o] * (pooled[o] > 0.f ? 1.f : slope) : 0.f;
  }
}

}  // namespace

PCR_EXPORT int pcr_edge_pool_fwd_f32(const float *y, const float *scale, const float *shift, float slope, float *pooled,
                                     int *arg, float *yraw, int B, int C, int N, int K, pcr_stream_t stream) {
  if (!y || !scale || !shift || !pooled || !arg || !yraw || B < 1 || C < 1 || N < 1 || K < 1) return PCR_ERR_INVALID;
  const size_t total = (size_t)B * C * N;
  size_t blocks = (total + 255) / 256;
  if (blocks > 16384) blocks = 16384;
  hipLaunchKernelGGL(edge_pool_fwd_kernel, dim3((unsigned)blocks), dim3(256), 0, pcr_s(stream), y, scale, shift, slope, pooled,
                     arg, yraw, C, N, K, total);
  PCR_CHECK_LAUNCH();
  return PCR_OK;
}

PCR_EXPORT int pcr_edge_pool_route_f32(const float *gp, const float *pooled, const int *arg, float slope, float *g, int B,
                                       int C, int N, int K, pcr_stream_t stream) {
  if (!gp || !pooled || !arg || !g || B < 1 || C < 1 || N < 1 || K < 1) return PCR_ERR_INVALID;
  const size_t total = (size_t)B * C * N * K;
  size_t blocks = (total + 255) / 256;
  if (blocks > 16384) blocks = 16384;
  hipLaunchKernelGGL(edge_pool_route_kernel, dim3((unsigned)blocks), dim3(256), 0, pcr_s(stream), gp, pooled, arg, slope, g, K,
                     total);
  PCR_CHECK_LAUNCH();
  return PCR_OK;
}

PCR_EXPORT int pcr_bn_sums_f32(const float *y, const float *g, const float *scale, const float *shift, int relu,
                               float slope, const float *centre, int centre_first, float *part, int nparts, int B, int C,
                               int L, pcr_stream_t stream) {
  if (!y || !part || nparts < 1 || B < 1 || C < 1 || L < 1 || nparts > 65535 || (g && relu && (!scale || !shift)))
    return PCR_ERR_INVALID;
  const int CP = (C + 31) & ~31;
  hipLaunchKernelGGL(bn_sums_kernel, dim3(C, nparts), dim3(256), 0, pcr_s(stream), y, g, scale, shift, relu, slope,
                     centre, centre_first, part, B, C, CP, L);
  PCR_CHECK_LAUNCH();
  return PCR_OK;
}

PCR_EXPORT int pcr_bn_affine_f32(const float *y, const float *g, const float *a0, const float *a1, const float *a2,
                                 const float *scale, const float *shift, const float *centre, int relu, float slope,
                                 float *out, int B, int C, int L, pcr_stream_t stream) {
  if (!y || !a0 || !a1 || !out || B < 1 || C < 1 || L < 1 || (g && (!a2 || (relu && (!scale || !shift)))))
    return PCR_ERR_INVALID;
  const size_t total = (size_t)B * C * L;
  size_t blocks = (total + 255) / 256;
  if (blocks > 16384) blocks = 16384;
  hipLaunchKernelGGL(bn_affine_kernel, dim3((unsigned)blocks), dim3(256), 0, pcr_s(stream), y, g, a0, a1, a2, scale, shift,
                     centre, relu, slope, out, C, L, total);
  PCR_CHECK_LAUNCH();
  return PCR_OK;
}

PCR_EXPORT int pcr_bmm_apply_f32(const float *x, const float *T, float *y, int B, int k, int N, int transposed,
                                 pcr_stream_t stream) {
  if (!x || !T || !y || B < 0 || k < 1 || k > 128 || N < 1) return PCR_ERR_INVALID;
  if (B == 0) return PCR_OK;
  if (B > 65535) return PCR_ERR_INVALID;
  const size_t lds = (size_t)k * (k + 1) * sizeof(float);
  hipLaunchKernelGGL(bmm_apply_kernel, dim3((N + 63) / 64, B), dim3(256), lds, pcr_s(stream), x, T, y, k, N, transposed);
  PCR_CHECK_LAUNCH();
  return PCR_OK;
}

PCR_EXPORT int pcr_bmm_dt_f32(const float *x, const float *dy, float *dT, int B, int k, int N, pcr_stream_t stream) {
  if (!x || !dy || !dT || B < 0 || k < 1 || k > 128 || N < 1) return PCR_ERR_INVALID;
  if (B == 0) return PCR_OK;
  if (B > 65535) return PCR_ERR_INVALID;
  hipLaunchKernelGGL(bmm_dt_kernel, dim3((k * k + 3) / 4, B), dim3(256), 0, pcr_s(stream), x, dy, dT, k, N);
  PCR_CHECK_LAUNCH();
  return PCR_OK;
}
