// Measurement aid of bench.py (not on the hot path): the clock the matrix cores really run at under sustained
// f32 MFMA load on every CU, so that `roofline.frac` can be read against the peak AT THAT CLOCK as well as against
// the 2.4 GHz spec peak (MI355X_MICROARCH.md: 157.3 TFLOP/s = 256 CUs x 4 SIMDs x 64 FLOP/clk x 2.4 GHz).
#include "tile_dense.h"

namespace {

// One wave per SIMD (a 128 KiB LDS request keeps it at one workgroup per CU), four independent accumulators:
// v_mfma_f32_32x32x2_f32 occupies the pipe for 16 passes x 4 cycles, so `iters` rounds of 16 MFMAs are exactly
// iters * 16 * 64 matrix-pipe cycles per wave when nothing else is scheduled.  Wave 0 stamps the shader-clock
// counter (s_memtime) and the constant-rate counter (s_memrealtime) around the loop.
__global__ __launch_bounds__(256) void clock_probe_kernel(unsigned long long *out, int iters) {
  extern __shared__ float smem[];
  f32x16 acc[4];
#pragma unroll
  for (int i = 0; i < 4; i++)
#pragma unroll
    for (int r = 0; r < 16; r++) acc[i][r] = 0.f;
  const float a = 1e-3f * (float)(threadIdx.x & 7), b = 1e-3f;
  const unsigned long long c0 = clock64(), w0 = wall_clock64();
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int u = 0; u < 4; u++)
#pragma unroll
      for (int i = 0; i < 4; i++) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 4; i++)
#pragma unroll
    for (int r = 0; r < 16; r++) s += acc[i][r];
  const unsigned long long c1 = clock64(), w1 = wall_clock64();
  if (s == 12345.678f) smem[threadIdx.x] = s;   // keeps the accumulators alive
  if (threadIdx.x == 0) {
    out[2 * blockIdx.x] = c1 - c0;
    out[2 * blockIdx.x + 1] = w1 - w0;
  }
}

}  // namespace

static thread_local int g_last_arith = -1;
void pcr_note_arith(int prec) { g_last_arith = prec; }
PCR_EXPORT int pcr_last_launch_arith(void) { return g_last_arith; }

PCR_EXPORT int pcr_wall_clock_khz(void) {
  int dev = 0, khz = 0;
  if (hipGetDevice(&dev) != hipSuccess ||
      hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, dev) != hipSuccess)
    return 0;
  return khz;
}

PCR_EXPORT int pcr_clock_probe(unsigned long long *ticks, int n_wg, int iters, pcr_stream_t stream) {
  if (!ticks || n_wg < 1 || iters < 1) return PCR_ERR_INVALID;
  static bool ok = allow_big_lds(clock_probe_kernel);
  (void)ok;
  hipLaunchKernelGGL(clock_probe_kernel, dim3(n_wg), dim3(256), 128 * 1024, pcr_s(stream), ticks, iters);
  PCR_CHECK_LAUNCH();
  return PCR_OK;
}
