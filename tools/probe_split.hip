// What does the split x = hi + lo (bf16 pair) cost per eight values on gfx950, alone and beside a running MFMA stream?
//   V0: the shipped sequence (tile_dense.h: bf_split8): 4 v_cvt_pk_bf16_f32, 8 v_and / v_lshlrev, 4 v_pk_add_f32, 4 v_cvt_pk_bf16_f32
//   V1: x - hi as v_dot2c_f32_bf16 (hi . {-1, 0} + x): 4 cvt, 8 dot2c, 4 cvt
//   V2: x - hi as eight v_sub_f32 instead of four packed adds
// Every variant carries the same 16 integer instructions that keep the loop alive (xor of the images, an integer step on x).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

template <int VAR>
__device__ __forceinline__ void split8(const float (&x)[8], bf16x8 &hi, bf16x8 &lo) {
  const bf16x2 m10 = {(__bf16)-1.0f, (__bf16)0.0f}, m01 = {(__bf16)0.0f, (__bf16)-1.0f};
#pragma unroll
  for (int q = 0; q < 4; q++) {
    const f32x2 v = {x[2 * q], x[2 * q + 1]};
    const bf16x2 h2 = __builtin_convertvector(v, bf16x2);
    hi[2 * q] = h2[0];
    hi[2 * q + 1] = h2[1];
    f32x2 r;
    if (VAR == 0) r = v - __builtin_convertvector(h2, f32x2);
    else if (VAR == 1) {
      r[0] = __builtin_amdgcn_fdot2_f32_bf16(h2, m10, v[0], false);
      r[1] = __builtin_amdgcn_fdot2_f32_bf16(h2, m01, v[1], false);
    } else {
      const unsigned hb = __builtin_bit_cast(unsigned, h2);
      float a = v[0] - __builtin_bit_cast(float, hb << 16), b = v[1] - __builtin_bit_cast(float, hb & 0xffff0000u);
      asm volatile("" : "+v"(a), "+v"(b));   // (keeps the two subtractions scalar)
      r = f32x2{a, b};
    }
    const bf16x2 l2 = __builtin_convertvector(r, bf16x2);
    lo[2 * q] = l2[0];
    lo[2 * q + 1] = l2[1];
  }
}

// ROLE 0: every wave splits; 1: every wave MFMAs; 2: waves 0-3 split, 4-7 MFMA (blockDim 512); 3: one wave does both, interleaved
template <int VAR, int ROLE, int NS>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void probe(float *out, int iters, int check) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  f32x16 acc[4];
  for (int c = 0; c < 4; c++)
    for (int r = 0; r < 16; r++) acc[c][r] = 0.f;
  float x[8];
  for (int c = 0; c < 8; c++) x[c] = (lane * 0.37f + c * 1.13f + 0.001f) * (c & 1 ? -1.f : 1.f);
  i32x4 ah = {0, 0, 0, 0}, al = {0, 0, 0, 0};
  bf16x8 a, b;
  for (int i = 0; i < 8; i++) { a[i] = (__bf16)(0.001f * lane + i); b[i] = (__bf16)(0.5f + 0.01f * i * lane); }
  const bool do_m = ROLE == 1 || ROLE == 3 || (ROLE == 2 && wave >= 4);
  const bool do_v = ROLE == 0 || ROLE == 3 || (ROLE == 2 && wave < 4);
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < 8; i++) {
      if (do_m) acc[i & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i & 3], 0, 0, 0);
      if (ROLE == 3) __builtin_amdgcn_sched_barrier(0);
      if (do_v) {
#pragma unroll
        for (int s = 0; s < NS; s++) {
          bf16x8 hi, lo;
          split8<VAR>(x, hi, lo);
          ah ^= __builtin_bit_cast(i32x4, hi);
          al ^= __builtin_bit_cast(i32x4, lo);
#pragma unroll
          for (int c = 0; c < 8; c++) x[c] = __builtin_bit_cast(float, __builtin_bit_cast(int, x[c]) + 0x1235);
        }
      }
      if (ROLE == 3) __builtin_amdgcn_sched_barrier(0);
    }
  }
  float s = 0.f;
  for (int c = 0; c < 4; c++)
    for (int r = 0; r < 16; r++) s += acc[c][r];
  int z = 0;
  for (int c = 0; c < 4; c++) z ^= ah[c] ^ (al[c] * 3);
  if (check) out[blockIdx.x * 512 + threadIdx.x] = __builtin_bit_cast(float, z);
  else out[blockIdx.x * 512 + threadIdx.x] = s + z;
}

template <int VAR, int ROLE, int NS>
float run(float *out, int waves, int iters = 20000) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  float ms = 0, best = 1e9;
  for (int rep = 0; rep < 4; rep++) {
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((probe<VAR, ROLE, NS>), dim3(256), dim3(64 * waves), 0, 0, out, iters, 0);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    (void)hipEventElapsedTime(&ms, e0, e1);
    if (rep && ms < best) best = ms;
  }
  return best;
}

template <int VAR>
void row(float *out, const char *name) {
  const float v4 = run<VAR, 0, 1>(out, 4), v8 = run<VAR, 0, 1>(out, 8);
  const float m4 = run<VAR, 1, 1>(out, 4);
  const float r2 = run<VAR, 2, 1>(out, 8);
  const float i1 = run<VAR, 3, 1>(out, 4), i2 = run<VAR, 3, 2>(out, 4), i1w8 = run<VAR, 3, 1>(out, 8);
  printf("%-22s split only: 1 wave/SIMD %.3f, 2 waves/SIMD %.3f | MFMA only %.3f | 4 split + 4 MFMA waves %.3f (sum %.3f) | "
         "one wave, 1 split8 per MFMA %.3f, 2 per MFMA %.3f; two such waves/SIMD %.3f\n",
         name, v4, v8, m4, r2, v4 + m4, i1, i2, i1w8);
}

int main() {
  float *out;
  (void)hipMalloc(&out, 256 * 512 * 4);
  // same bits from the three variants?  (the kernel writes at blockIdx * 512 + threadIdx)
  int d1 = 0, d2 = 0;
  {
    unsigned *buf = new unsigned[256 * 512];
    unsigned *ref = new unsigned[256 * 512];
    hipLaunchKernelGGL((probe<0, 0, 1>), dim3(256), dim3(256), 0, 0, out, 100, 1);
    (void)hipMemcpy(ref, out, 256 * 512 * 4, hipMemcpyDeviceToHost);
    hipLaunchKernelGGL((probe<1, 0, 1>), dim3(256), dim3(256), 0, 0, out, 100, 1);
    (void)hipMemcpy(buf, out, 256 * 512 * 4, hipMemcpyDeviceToHost);
    for (int bI = 0; bI < 256; bI++) for (int t = 0; t < 256; t++) d1 += buf[bI * 512 + t] != ref[bI * 512 + t];
    hipLaunchKernelGGL((probe<2, 0, 1>), dim3(256), dim3(256), 0, 0, out, 100, 1);
    (void)hipMemcpy(buf, out, 256 * 512 * 4, hipMemcpyDeviceToHost);
    for (int bI = 0; bI < 256; bI++) for (int t = 0; t < 256; t++) d2 += buf[bI * 512 + t] != ref[bI * 512 + t];
  }
  printf("image checksums differing from the shipped sequence: dot2c %d, scalar sub %d (of 65536 lanes)\n", d1, d2);
  row<0>(out, "shipped (pk_add)");
  row<1>(out, "v_dot2c_f32_bf16");
  row<2>(out, "v_sub_f32 x 8");
  return 0;
}
