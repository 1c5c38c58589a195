// Measurement probe (not part of libpcr_hip.so; hipcc --offload-arch=gfx950 -O3 -w tools/probe_coexec.hip -o /tmp/p && /tmp/p):
// do VALU instructions of one wave execute under the MFMAs of another wave of the same SIMD (gfx950)?
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// one "M chunk" = 16 MFMAs on 4 accumulators (512 pipe cycles); one "V chunk" = 128 independent-ish VALU fmas (512 issue cycles)
__device__ __forceinline__ void m_chunk(f32x16 (&acc)[4], bf16x8 a, bf16x8 b) {
#pragma unroll
  for (int i = 0; i < 4; i++)
#pragma unroll
    for (int c = 0; c < 4; c++) acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[c], 0, 0, 0);
}
__device__ __forceinline__ void v_chunk(float (&x)[16], float k) {
#pragma unroll
  for (int i = 0; i < 8; i++)
#pragma unroll
    for (int c = 0; c < 16; c++) x[c] = __builtin_fmaf(x[c], k, 0.5f);
}

// mode 0: every wave: M only.  1: V only.  2: waves 0-3 M, waves 4-7 V.  3: every wave alternates nm M chunks / nv V chunks
// (all in step).  4: the same, waves 4-7 start with their V phase (anti-phase).  5: one stream, 1 M chunk then 1 V chunk
// interleaved by the scheduler (same basic block).
template <int MODE>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void probe(float *out, int iters, int nm, int nv) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  f32x16 acc[4];
  for (int c = 0; c < 4; c++)
    for (int r = 0; r < 16; r++) acc[c][r] = 0.f;
  float x[16];
  for (int c = 0; c < 16; c++) x[c] = lane * 0.001f + c;
  bf16x8 a, b;
  for (int i = 0; i < 8; i++) { a[i] = (__bf16)(0.001f * lane); b[i] = (__bf16)0.5f; }
  const float k = 0.999f + 1e-6f * lane;
  const bool second = wave >= 4;
  for (int it = 0; it < iters; it++) {
    if (MODE == 0 || (MODE == 2 && !second)) {
      for (int j = 0; j < nm; j++) m_chunk(acc, a, b);
    } else if (MODE == 1 || (MODE == 2 && second)) {
      for (int j = 0; j < nv; j++) v_chunk(x, k);
    } else if (MODE == 3 || (MODE == 4 && !second)) {
      for (int j = 0; j < nm; j++) m_chunk(acc, a, b);
      asm volatile("" ::: "memory");
      for (int j = 0; j < nv; j++) v_chunk(x, k);
    } else if (MODE == 4) {
      for (int j = 0; j < nv; j++) v_chunk(x, k);
      asm volatile("" ::: "memory");
      for (int j = 0; j < nm; j++) m_chunk(acc, a, b);
    } else {
      for (int j = 0; j < nm; j++) {
        m_chunk(acc, a, b);
        v_chunk(x, k);
      }
    }
  }
  float s = 0.f;
  for (int c = 0; c < 4; c++)
    for (int r = 0; r < 16; r++) s += acc[c][r];
  for (int c = 0; c < 16; c++) s += x[c];
  out[blockIdx.x * 512 + threadIdx.x] = s;
}

// mode 6: ONE stream, instruction-level interleave: after every MFMA, KV independent VALU fmas (pinned by sched_barrier)
template <int KV>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void probe_il(float *out, int iters, int nm) {
  const int lane = threadIdx.x & 63;
  f32x16 acc[4];
  for (int c = 0; c < 4; c++)
    for (int r = 0; r < 16; r++) acc[c][r] = 0.f;
  float x[16];
  for (int c = 0; c < 16; c++) x[c] = lane * 0.001f + c;
  bf16x8 a, b;
  for (int i = 0; i < 8; i++) { a[i] = (__bf16)(0.001f * lane); b[i] = (__bf16)0.5f; }
  const float k = 0.999f + 1e-6f * lane;
  for (int it = 0; it < iters * nm; it++) {
#pragma unroll
    for (int i = 0; i < 16; i++) {
      acc[i & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i & 3], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int v = 0; v < KV; v++) x[(i * KV + v) & 15] = __builtin_fmaf(x[(i * KV + v) & 15], k, 0.5f);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  float s = 0.f;
  for (int c = 0; c < 4; c++)
    for (int r = 0; r < 16; r++) s += acc[c][r];
  for (int c = 0; c < 16; c++) s += x[c];
  out[blockIdx.x * 512 + threadIdx.x] = s;
}
template <int KV>
float run_il(float *out, int waves, int iters, int nm) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  float ms = 0;
  for (int rep = 0; rep < 3; rep++) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((probe_il<KV>), dim3(256), dim3(64 * waves), 0, 0, out, iters, nm);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
  }
  return ms;
}

template <int MODE>
float run(float *out, int waves, int iters, int nm, int nv) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  float ms = 0;
  for (int rep = 0; rep < 3; rep++) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((probe<MODE>), dim3(256), dim3(64 * waves), 0, 0, out, iters, nm, nv);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
  }
  return ms;
}

int main() {
  float *out;
  hipMalloc(&out, 256 * 512 * 4);
  const int iters = 2000;
  printf("instruction-level interleave, 16 MFMA + 16 KV fma per chunk (nm = 8): M-only time 3.67 ms; V time = KV/8 x 2.07 ms\n");
  printf("  KV=0: 4 waves %.3f  8 waves %.3f\n", run_il<0>(out, 4, iters, 8), run_il<0>(out, 8, iters, 8));
  printf("  KV=4: 4 waves %.3f  8 waves %.3f\n", run_il<4>(out, 4, iters, 8), run_il<4>(out, 8, iters, 8));
  printf("  KV=8: 4 waves %.3f  8 waves %.3f\n", run_il<8>(out, 4, iters, 8), run_il<8>(out, 8, iters, 8));
  printf("  KV=12: 4 waves %.3f  8 waves %.3f\n", run_il<12>(out, 4, iters, 8), run_il<12>(out, 8, iters, 8));
  printf("  KV=16: 4 waves %.3f  8 waves %.3f\n", run_il<16>(out, 4, iters, 8), run_il<16>(out, 8, iters, 8));
  for (int nm = 8; nm <= 8; nm *= 2) {
    const int nv = nm;
    // chunk = 512 cycles: ideal per-wave ms for nm chunks x iters at 2.4 GHz
    const double unit = iters * nm * 512.0 / 2.4e6;
    printf("nm=nv=%d (phase = %d cycles); 1 phase-set per wave = %.3f ms\n", nm, nm * 512, unit);
    printf("  M only, 4 waves (1/SIMD): %.3f   8 waves (2/SIMD): %.3f\n", run<0>(out, 4, iters, nm, nv), run<0>(out, 8, iters, nm, nv));
    printf("  V only, 4 waves: %.3f   8 waves: %.3f\n", run<1>(out, 4, iters, nm, nv), run<1>(out, 8, iters, nm, nv));
    printf("  split roles (4 M + 4 V): %.3f   [perfect overlap = max of the 4-wave times, none = sum]\n", run<2>(out, 8, iters, nm, nv));
    printf("  alternating, in step, 8 waves: %.3f   4 waves: %.3f\n", run<3>(out, 8, iters, nm, nv), run<3>(out, 4, iters, nm, nv));
    printf("  alternating, anti-phase, 8 waves: %.3f\n", run<4>(out, 8, iters, nm, nv));
    printf("  one stream interleaved, 4 waves: %.3f   8 waves: %.3f\n", run<5>(out, 4, iters, nm, nv), run<5>(out, 8, iters, nm, nv));
  }
  return 0;
}
