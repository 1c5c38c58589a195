// Measurement probe (not part of libpcr_hip.so): how fast does v_mfma_f32_32x32x2_f32 issue on gfx950 with
//   mode 0: operands in registers only            mode 1: B operand from LDS (ds_read_b32 per MFMA)
//   mode 2: + A operand streamed from global (16-byte loads, ring of 4)   mode 3: + __syncthreads every 16 k-blocks
// for 1..3 workgroups of 4 waves per CU.   hipcc --offload-arch=gfx950 -O3 tools/probe_mfma.hip -o /tmp/probe && /tmp/probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE, int NACC>
__global__ __launch_bounds__(256) void probe(const float *__restrict__ w, float *__restrict__ out, int iters, int lds_pad) {
  extern __shared__ float lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 128 * 65; i += 256) lds[i] = (float)(i & 7) * 0.125f;
  __syncthreads();
  f32x16 acc[NACC];
  for (int a = 0; a < NACC; a++)
    for (int r = 0; r < 16; r++) acc[a][r] = 0.f;
  const f32x4 *wp = reinterpret_cast<const f32x4 *>(w) + wave * 64 + lane;
  f32x4 ring[4];
  for (int i = 0; i < 4; i++) ring[i] = wp[i * 512];
  const float *bp = lds + (lane >> 5) * 65 + (lane & 31);
  float a0 = 1.0f + lane, b0 = 0.5f;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < 4; i++) {
      f32x4 av = MODE >= 2 ? ring[i] : f32x4{a0, a0, a0, a0};
#pragma unroll
      for (int q = 0; q < 4; q++) {
#pragma unroll
        for (int a = 0; a < NACC; a++) {
          const float bv = MODE >= 1 ? bp[((it * 4 + i) & 15) * 8 * 65 + q * 2 * 65 + (a & 1) * 32] : b0;
          acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[q], bv, acc[a], 0, 0, 0);
        }
      }
      if (MODE >= 2) {
        ring[i] = wp[(size_t)(((it + 1) * 4 + i) & 63) * 512];
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (MODE >= 3 && (it & 3) == 3) __syncthreads();
  }
  float s = 0.f;
  for (int a = 0; a < NACC; a++)
    for (int r = 0; r < 16; r++) s += acc[a][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE, int NACC>
void run(int wgs_per_cu, const float *w, float *out) {
  const int iters = 2048, n_cu = 256;
  const size_t lds = (wgs_per_cu == 1 ? 150 : wgs_per_cu == 2 ? 76 : 50) * 1024;
  hipFuncSetAttribute(reinterpret_cast<const void *>(probe<MODE, NACC>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int rep = 0; rep < 2; rep++) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((probe<MODE, NACC>), dim3(n_cu * wgs_per_cu), dim3(256), lds, 0, w, out, iters, 0);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
  }
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  const double flop = (double)n_cu * wgs_per_cu * 4 * iters * 16.0 * NACC * 4096.0;
  printf("mode %d  acc %d  wg/cu %d : %.3f ms  %.1f TFLOP/s\n", MODE, NACC, wgs_per_cu, ms, flop / ms * 1e-9);
}

int main() {
  float *w, *out;
  hipMalloc(&w, 64 * 512 * 16 + 4096 * 16);
  hipMemset(w, 0, 64 * 512 * 16 + 4096 * 16);
  hipMalloc(&out, 256 * 3 * 256 * 4);
  for (int k = 1; k <= 3; k++) {
    run<0, 2>(k, w, out);
    run<0, 4>(k, w, out);
    run<1, 2>(k, w, out);
    run<1, 4>(k, w, out);
    run<2, 2>(k, w, out);
    run<2, 4>(k, w, out);
    run<3, 2>(k, w, out);
    run<3, 4>(k, w, out);
  }
  return 0;
}
