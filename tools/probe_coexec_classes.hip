// Which instruction classes execute under a running MFMA on gfx950?  One wave per SIMD (or two), instruction-level
// interleave: after every v_mfma_f32_32x32x16_bf16 (32 pipe cycles), KV instructions of one class.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

template <int OP, int KV, bool MF>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void probe(float *out, int iters) {
  __shared__ float lds[4096];
  const int lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = i;
  __syncthreads();
  f32x16 acc[4];
  for (int c = 0; c < 4; c++)
    for (int r = 0; r < 16; r++) acc[c][r] = 0.f;
  float x[16];
  int xi[16];
  f32x2 xp[8];
  for (int c = 0; c < 16; c++) { x[c] = lane * 0.001f + c; xi[c] = lane + c; }
  for (int c = 0; c < 8; c++) xp[c] = f32x2{x[c], x[c + 8]};
  bf16x8 a, b;
  for (int i = 0; i < 8; i++) { a[i] = (__bf16)(0.001f * lane); b[i] = (__bf16)0.5f; }
  const float k = 0.999f + 1e-6f * lane;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < 16; i++) {
      if (MF) acc[i & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i & 3], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int v = 0; v < KV; v++) {
        const int c = (i * KV + v) & 15;
        if (OP == 0) x[c] = __builtin_fmaf(x[c], k, 0.5f);
        else if (OP == 1) xi[c] = xi[c] + (xi[(c + 1) & 15] & 0xff);            // v_and + v_add_u32 (counts as 2)
        else if (OP == 2) xi[c] = max(xi[c], xi[(c + 5) & 15] - 1);             // v_add + v_max_i32 (2)
        else if (OP == 3) { bf16x2 h = __builtin_convertvector(f32x2{x[c], x[(c + 1) & 15]}, bf16x2); xi[c] ^= (int)__builtin_bit_cast(unsigned, h); }  // cvt_pk + xor (2)
        else if (OP == 4) x[c] += lds[(xi[c] + it) & 4095];                     // ds_read_b32 + add (+ address ops)
        else if (OP == 5) xp[c & 7] = xp[c & 7] * f32x2{k, k} + f32x2{0.5f, 0.5f};   // v_pk_fma_f32
        else if (OP == 6) x[c] = __builtin_amdgcn_exp2f(x[c] * 0.001f);           // v_exp_f32 (+ mul)
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  float s = 0.f;
  for (int c = 0; c < 4; c++)
    for (int r = 0; r < 16; r++) s += acc[c][r];
  for (int c = 0; c < 16; c++) s += x[c] + xi[c];
  for (int c = 0; c < 8; c++) s += xp[c][0] + xp[c][1];
  out[blockIdx.x * 512 + threadIdx.x] = s;
}

template <int OP, int KV, bool MF>
float run(float *out, int waves) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  float ms = 0;
  for (int rep = 0; rep < 3; rep++) {
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((probe<OP, KV, MF>), dim3(256), dim3(64 * waves), 0, 0, out, 16000);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    (void)hipEventElapsedTime(&ms, e0, e1);
  }
  return ms;
}

template <int OP>
void row(float *out, const char *name) {
  const float m = run<OP, 0, true>(out, 4);
  const float v = run<OP, 8, false>(out, 4);
  const float mv = run<OP, 8, true>(out, 4);
  const float v2 = run<OP, 4, false>(out, 4), mv2 = run<OP, 4, true>(out, 4);
  printf("%-34s M %.3f | KV=8: V %.3f  M+V %.3f (sum %.3f, max %.3f) | KV=4: V %.3f  M+V %.3f (sum %.3f)\n", name, m, v, mv, m + v,
         m > v ? m : v, v2, mv2, m + v2);
}

int main() {
  float *out;
  (void)hipMalloc(&out, 256 * 512 * 4);
  row<0>(out, "v_fma_f32");
  row<1>(out, "v_and + v_add_u32");
  row<2>(out, "v_add + v_max_i32");
  row<3>(out, "v_cvt_pk_bf16_f32 + v_xor");
  row<4>(out, "ds_read_b32 + v_add_f32 + addr");
  row<5>(out, "v_pk_fma_f32");
  row<6>(out, "v_mul + v_exp_f32");
  return 0;
}
