// The parameter update of one training iteration in two launches: the global gradient norm (mmcv OptimizerHook
// grad_clip, configs_reid/_base_/schedules/cyclic_200e_lr3e-4.py:9 -> torch.nn.utils.clip_grad_norm_) and AdamW
// (cyclic_200e_lr3e-4.py:7) over EVERY parameter tensor of the model at once.  The tensors stay where the module holds them;
// a device table lists their pointers and the launch is cut into fixed chunks of kChunk elements listed by
// (tensor, first element), so that ~150 small tensors cost two launches instead of a few multi-tensor launches per
// arithmetic step.  Every sum runs in a fixed order (no atomics): the update is reproducible bit for bit.
#include "pcr_common.h"

namespace {

constexpr int kThreads = 256;
constexpr int kChunk = 2048;   // elements per workgroup (8 per thread, strided by the block: coalesced 1 KB rows)

struct OptTensor {             // = pcr_opt_tensor
  float *p, *g, *m, *v;
  long n;
  float step_size, bc2_sqrt, decay, one_m_beta1, beta2, one_m_beta2, eps, pad_;
};
static_assert(sizeof(OptTensor) == sizeof(pcr_opt_tensor), "pcr_opt_tensor layout");

__device__ __forceinline__ double block_sum(double v, double *red) {
  // lanes: xor butterflies visit the same pairs on every run; waves: summed in index order by every thread
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
  const int tid = threadIdx.x;
  __syncthreads();
  if ((tid & 63) == 0) red[tid >> 6] = v;
  __syncthreads();
  double s = 0.0;
#pragma unroll
  for (int w = 0; w < kThreads / 64; w++) s += red[w];
  return s;
}

__global__ __launch_bounds__(kThreads) void grad_sumsq_kernel(const OptTensor *__restrict__ tab,
                                                              const int *__restrict__ chunk_tensor,
                                                              const int *__restrict__ chunk_first,
                                                              double *__restrict__ part) {
  __shared__ double red[kThreads / 64];
  const int c = blockIdx.x;
  const OptTensor t = tab[chunk_tensor[c]];
  const long first = chunk_first[c];
  double s = 0.0;
  if (t.g) {
#pragma unroll
    for (int u = 0; u < kChunk / kThreads; u++) {
      const long e = first + threadIdx.x + u * kThreads;
      if (e < t.n) {
        const float g = t.g[e];
        s += (double)g * (double)g;
      }
    }
  }
  s = block_sum(s, red);
  if (threadIdx.x == 0) part[c] = s;
}

__global__ __launch_bounds__(kThreads) void adamw_step_kernel(const OptTensor *__restrict__ tab,
                                                              const int *__restrict__ chunk_tensor,
                                                              const int *__restrict__ chunk_first, int n_chunks,
                                                              const double *__restrict__ part, float max_norm,
                                                              float *__restrict__ norm_out) {
  __shared__ double red[kThreads / 64];
  const int c = blockIdx.x;
  float coef = 1.f;
  if (part) {
    // every workgroup adds the chunk sums in the same order: one norm, no second launch, no host round trip
    double s = 0.0;
    for (int i = threadIdx.x; i < n_chunks; i += kThreads) s += part[i];
    s = block_sum(s, red);
    const float norm = (float)sqrt(s);
    if (c == 0 && threadIdx.x == 0 && norm_out) *norm_out = norm;
    if (max_norm > 0.f) {
      coef = max_norm / (norm + 1e-6f);          // clip_grad_norm_: clamp(max_norm / (total + 1e-6), max = 1)
      coef = coef > 1.f ? 1.f : coef;
    }
  }
  const OptTensor t = tab[chunk_tensor[c]];
  if (!t.g) return;                                // a parameter without a gradient is left alone, as in torch
  const long first = chunk_first[c];
  const float one_m_b1 = t.one_m_beta1, one_m_b2 = t.one_m_beta2;   // (rounded from double on the host, as torch's are)
#pragma unroll
  for (int u = 0; u < kChunk / kThreads; u++) {
    const long e = first + threadIdx.x + u * kThreads;
    if (e < t.n) {
      const float g = t.g[e] * coef;
      float p = t.p[e] * t.decay;                  // decoupled weight decay: p <- p (1 - lr wd)
      float m = t.m[e], v = t.v[e];
      m = m + one_m_b1 * (g - m);                  // exp_avg.lerp_(grad, 1 - beta1)
      v = t.beta2 * v + one_m_b2 * g * g;
      const float denom = sqrtf(v) / t.bc2_sqrt + t.eps;
      p = p - t.step_size * (m / denom);
      if (part) t.g[e] = g;                        // the clipped gradient stays visible in .grad, as after the hook
      t.p[e] = p;
      t.m[e] = m;
      t.v[e] = v;
    }
  }
}

}  // namespace

PCR_EXPORT int pcr_opt_chunk(void) { return kChunk; }

PCR_EXPORT int pcr_grad_sumsq_f32(const pcr_opt_tensor *tab, const int *chunk_tensor, const int *chunk_first,
                                  int n_chunks, double *part, pcr_stream_t stream) {
  if (!tab || !chunk_tensor || !chunk_first || !part || n_chunks < 0) return PCR_ERR_INVALID;
  if (n_chunks == 0) return PCR_OK;
  hipLaunchKernelGGL(grad_sumsq_kernel, dim3(n_chunks), dim3(kThreads), 0, pcr_s(stream),
                     reinterpret_cast<const OptTensor *>(tab), chunk_tensor, chunk_first, part);
  PCR_CHECK_LAUNCH();
  return PCR_OK;
}

PCR_EXPORT int pcr_adamw_step_f32(const pcr_opt_tensor *tab, const int *chunk_tensor, const int *chunk_first,
                                  int n_chunks, const double *part, float max_norm, float *grad_norm,
                                  pcr_stream_t stream) {
  if (!tab || !chunk_tensor || !chunk_first || n_chunks < 0) return PCR_ERR_INVALID;
  if (n_chunks == 0) return PCR_OK;
  hipLaunchKernelGGL(adamw_step_kernel, dim3(n_chunks), dim3(kThreads), 0, pcr_s(stream),
                     reinterpret_cast<const OptTensor *>(tab), chunk_tensor, chunk_first, n_chunks, part, max_norm,
                     grad_norm);
  PCR_CHECK_LAUNCH();
  return PCR_OK;
}
