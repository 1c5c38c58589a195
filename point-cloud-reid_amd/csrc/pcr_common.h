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
