// micro-benchmark: bandwidth of a (B, C, L) -> (B, C, L) copy done tile by tile, TT tokens (TT * 4 contiguous bytes per row)
// per workgroup pass, as the train-dense kernels access their tensors.  hipcc --offload-arch=gfx950 -O3 tile_copy.hip -o tile_copy
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int TT>
__global__ __launch_bounds__(256) void copy_tiled(const float *__restrict__ x, float *__restrict__ y, int C, int L, int tpw) {
  constexpr int Q = TT / 4;
  const size_t b = blockIdx.y;
  const float *xb = x + b * C * L;
  float *yb = y + b * C * L;
  for (int ti = 0; ti < tpw; ti++) {
    const int t0 = (blockIdx.x * tpw + ti) * TT;
    if (t0 >= L) break;
    for (int e = threadIdx.x; e < C * Q; e += 256) {
      const int c = e / Q, q = e % Q;
      const f32x4 v = *reinterpret_cast<const f32x4 *>(xb + (size_t)c * L + t0 + 4 * q);
      *reinterpret_cast<f32x4 *>(yb + (size_t)c * L + t0 + 4 * q) = v * 1.0001f;
    }
  }
}
template <int TT>
float run(const float *x, float *y, int B, int C, int L) {
  const int ntiles = L / TT, tpw = ntiles >= 24 ? ntiles / 2 : ntiles;     // two workgroups per cloud like the real launch
  dim3 grid((ntiles + tpw - 1) / tpw, B);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; i++) hipLaunchKernelGGL(copy_tiled<TT>, grid, dim3(256), 0, 0, x, y, C, L, tpw);
  hipEventRecord(e0);
  for (int i = 0; i < 10; i++) hipLaunchKernelGGL(copy_tiled<TT>, grid, dim3(256), 0, 0, x, y, C, L, tpw);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms / 10;
}
int main() {
  const int B = 512, L = 3072;
  for (int C : {32, 64, 128}) {
    float *x, *y; size_t n = (size_t)B * C * L;
    hipMalloc(&x, n * 4); hipMalloc(&y, n * 4); hipMemset(x, 0, n * 4);
    float a = run<64>(x, y, B, C, L), b = run<128>(x, y, B, C, L), c = run<256>(x, y, B, C, L), d = run<1024>(x, y, B, C, L);
    double gb = 2.0 * n * 4 / 1e9;
    printf("C=%3d L=%d B=%d: tile 64 tokens %.0f GB/s | 128: %.0f | 256: %.0f | 1024: %.0f\n", C, L, B, gb / a * 1e3, gb / b * 1e3, gb / c * 1e3, gb / d * 1e3);
    hipFree(x); hipFree(y);
  }
  return 0;
}
