// Measurement probe (not part of libpcr_hip.so; hipcc --offload-arch=gfx950 -O3 -w tools/probe_clock.hip -o /tmp/p && /tmp/p):
// the shader clock a kernel actually runs at, by load.  Every wave reads the shader-clock counter (s_memtime: counts core
// cycles) and the constant 100 MHz counter (s_memrealtime) at its start and end; MHz = d(shader) / d(real) x 100.
// The peak of MI355X_MICROARCH.md (2.5 PFLOP/s dense bf16) is 256 CUs x 4 SIMDs x 1024 flop/cycle x 2.4 GHz: a kernel that
// keeps the matrix pipes busy on a chip that sustains a lower clock under that load cannot reach it, whatever its schedule.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include <algorithm>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// duty: of every 16 chunks, `duty` are 16 bf16 MFMAs (512 pipe cycles), the rest 128 VALU fmas (512 issue cycles)
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void load_kernel(unsigned long long *stamps, float *out,
                                                                                                int iters, int duty) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const unsigned long long c0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
  f32x16 acc[4];
  for (int c = 0; c < 4; c++)
    for (int r = 0; r < 16; r++) acc[c][r] = 0.f;
  float x[16];
  for (int c = 0; c < 16; c++) x[c] = lane * 0.001f + c;
  bf16x8 a, b;
  for (int i = 0; i < 8; i++) { a[i] = (__bf16)(0.001f * lane); b[i] = (__bf16)0.5f; }
  const float k = 0.999f + 1e-6f * lane;
  for (int it = 0; it < iters; it++) {
    for (int ch = 0; ch < 16; ch++) {
      if (ch < duty) {
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
          for (int c = 0; c < 4; c++) acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[c], 0, 0, 0);
      } else {
#pragma unroll
        for (int i = 0; i < 8; i++)
#pragma unroll
          for (int c = 0; c < 16; c++) x[c] = __builtin_fmaf(x[c], k, 0.5f);
      }
    }
  }
  const unsigned long long c1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0.f;
  for (int c = 0; c < 4; c++)
    for (int r = 0; r < 16; r++) s += acc[c][r];
  for (int c = 0; c < 16; c++) s += x[c];
  if (s == 12345.678f) out[0] = s;
  if (lane == 0) {
    stamps[(blockIdx.x * 8 + wave) * 2] = c1 - c0;
    stamps[(blockIdx.x * 8 + wave) * 2 + 1] = r1 - r0;
  }
}

// mode bits: 1 = random operands (eight different A / B register sets, random bf16 bits), 2 = the B operands come from LDS
// (one ds_read_b128 per 1.5 MFMAs, the ratio of the K-row SA kernel), 4 = 25 % of the chunks are VALU chunks
template <int mode>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void load2_kernel(unsigned long long *stamps, float *out,
                                                                                                 int iters) {
  __shared__ bf16x8 s_w[64 * 64];      // 64 KB
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned int seed = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
  auto rnd = [&]() { seed = seed * 1664525u + 1013904223u; return seed; };
  for (int e = threadIdx.x; e < 64 * 64; e += 512) {
    bf16x8 v;
    for (int i = 0; i < 8; i++) {
      const unsigned short bits = (mode & 1) ? (unsigned short)((rnd() >> 16) & 0xbfff) | 0x3000 : 0x3f00;   // |x| in a sane range
      v[i] = __builtin_bit_cast(__bf16, bits);
    }
    s_w[e] = v;
  }
  __syncthreads();
  const unsigned long long c0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
  f32x16 acc[4];
  for (int c = 0; c < 4; c++)
    for (int r = 0; r < 16; r++) acc[c][r] = 0.f;
  float x[16];
  for (int c = 0; c < 16; c++) x[c] = lane * 0.001f + c;
  bf16x8 a[8], b[8];
  for (int q = 0; q < 8; q++) { a[q] = s_w[(q * 64 + lane) & 4095]; b[q] = s_w[((q + 8) * 64 + lane) & 4095]; }
  const float k = 0.999f + 1e-6f * lane;
  for (int it = 0; it < iters; it++) {
    for (int ch = 0; ch < 16; ch++) {
      if ((mode & 4) && (ch & 3) == 3) {
#pragma unroll
        for (int i = 0; i < 8; i++)
#pragma unroll
          for (int c = 0; c < 16; c++) x[c] = __builtin_fmaf(x[c], k, 0.5f);
        continue;
      }
      if (mode & 2) {
        const bf16x8 *wb = s_w + ((it * 16 + ch) & 7) * 512 + lane;
#pragma unroll
        for (int q = 0; q < 8; q++) b[q] = wb[q * 64];
        asm volatile("" ::: "memory");
      }
#pragma unroll
      for (int i = 0; i < 4; i++)
#pragma unroll
        for (int c = 0; c < 4; c++) {
          const int q = (i * 4 + c) & 7;
          acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[(mode & 1) ? q : 0], b[(mode & 3) ? (q ^ (i & 1)) : 0], acc[c], 0, 0, 0);
        }
    }
    if ((it & 63) == 63)      // keep the accumulators finite
      for (int c = 0; c < 4; c++)
        for (int r = 0; r < 16; r++) acc[c][r] *= 1e-3f;
  }
  const unsigned long long c1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0.f;
  for (int c = 0; c < 4; c++)
    for (int r = 0; r < 16; r++) s += acc[c][r];
  for (int c = 0; c < 16; c++) s += x[c];
  if (s == 12345.678f) out[0] = s;
  if (lane == 0) {
    stamps[(blockIdx.x * 8 + wave) * 2] = c1 - c0;
    stamps[(blockIdx.x * 8 + wave) * 2 + 1] = r1 - r0;
  }
}

int main() {
  hipDeviceProp_t prop;
  hipGetDeviceProperties(&prop, 0);
  const int ncu = prop.multiProcessorCount;
  unsigned long long *d;
  float *o;
  hipMalloc(&d, (size_t)ncu * 8 * 2 * sizeof(unsigned long long));
  hipMalloc(&o, 64);
  std::vector<unsigned long long> h((size_t)ncu * 8 * 2);
  printf("# %s, %d CUs, clockRate %d kHz\n", prop.name, ncu, prop.clockRate);
  // (grid, label): one workgroup = the light case (one CU busy); one per CU = the whole chip
  const int grids[2] = {1, ncu};
  for (int gi = 0; gi < 2; gi++)
    for (int duty = 0; duty <= 16; duty += 4) {
      for (int ms_i = 0; ms_i < 2; ms_i++) {
        const int iters = ms_i ? 12000 : 600;     // ~0.25 ms and ~5 ms of work: a short launch may still ride a boost
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        for (int rep = 0; rep < 3; rep++) {         // (the third launch is the one reported)
          hipEventRecord(e0);
          hipLaunchKernelGGL(load_kernel, dim3(grids[gi]), dim3(512), 0, 0, d, o, iters, duty);
          hipEventRecord(e1);
          hipEventSynchronize(e1);
        }
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        hipMemcpy(h.data(), d, (size_t)grids[gi] * 8 * 2 * sizeof(unsigned long long), hipMemcpyDeviceToHost);
        std::vector<double> mhz;
        for (int w = 0; w < grids[gi] * 8; w++) mhz.push_back(100.0 * (double)h[2 * w] / (double)h[2 * w + 1]);
        std::sort(mhz.begin(), mhz.end());
        printf("workgroups %4d  matrix duty %2d/16  launch %7.3f ms  shader clock MHz: min %6.0f median %6.0f max %6.0f\n", grids[gi], duty, ms,
               mhz.front(), mhz[mhz.size() / 2], mhz.back());
      }
    }
  printf("# whole chip, 16 bf16 MFMAs per chunk; mode bits: 1 random operands, 2 B operands from LDS (ds_read_b128 per 1.5 MFMAs), 4 a quarter VALU chunks\n");
  for (int mode = 0; mode < 8; mode++)
    for (int ms_i = 0; ms_i < 2; ms_i++) {
      const int iters = ms_i ? 20000 : 600;   // ~4 ms and ~140 ms
      hipEvent_t e0, e1;
      hipEventCreate(&e0); hipEventCreate(&e1);
      for (int rep = 0; rep < 3; rep++) {
        hipEventRecord(e0);
        switch (mode) {
#define L2K(M) case M: hipLaunchKernelGGL(load2_kernel<M>, dim3(ncu), dim3(512), 0, 0, d, o, iters); break;
          L2K(0) L2K(1) L2K(2) L2K(3) L2K(4) L2K(5) L2K(6) L2K(7)
#undef L2K
        }
        hipEventRecord(e1);
        hipEventSynchronize(e1);
      }
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      hipMemcpy(h.data(), d, (size_t)ncu * 8 * 2 * sizeof(unsigned long long), hipMemcpyDeviceToHost);
      std::vector<double> mhz;
      double cyc = 0;
      for (int w = 0; w < ncu * 8; w++) { mhz.push_back(100.0 * (double)h[2 * w] / (double)h[2 * w + 1]); cyc += (double)h[2 * w]; }
      std::sort(mhz.begin(), mhz.end());
      const double mfmas = (double)ncu * 8 * iters * ((mode & 4) ? 12 : 16) * 16;
      printf("mode %d  launch %7.3f ms  %7.1f TFLOP/s bf16 issued  shader clock MHz: min %6.0f median %6.0f max %6.0f\n", mode, ms,
             mfmas * 32768.0 / (ms * 1e-3) / 1e12, mhz.front(), mhz[mhz.size() / 2], mhz.back());
    }
  return 0;
}
