// micro-test: what LDS layout does global_load_lds_dwordx4 produce?  hipcc --offload-arch=gfx950 glds_test.hip -o glds_test && ./glds_test
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((address_space(1))) const void *gptr_t;
typedef __attribute__((address_space(3))) void *lptr_t;
__global__ void k(const float *g, float *out) {
  __shared__ __attribute__((aligned(16))) float lds[2 * 64 * 4 + 64];
  const int lane = threadIdx.x;
  for (int i = lane; i < 2 * 64 * 4 + 64; i += 64) lds[i] = -1.f;
  __syncthreads();
  // lane l loads 16 B from g + (63 - l) * 4 floats (reversed), second instruction from g + 256 + l * 4, into lds + 256 + 8
  __builtin_amdgcn_global_load_lds((gptr_t)(g + (63 - lane) * 4), (lptr_t)lds, 16, 0, 0);
  __builtin_amdgcn_global_load_lds((gptr_t)(g + 256 + lane * 4), (lptr_t)(lds + 256 + 8), 16, 0, 0);
  __builtin_amdgcn_s_waitcnt(0);
  __syncthreads();
  for (int i = lane; i < 2 * 64 * 4 + 64; i += 64) out[i] = lds[i];
}
int main() {
  std::vector<float> h(512);
  for (int i = 0; i < 512; i++) h[i] = (float)i;
  float *g, *o;
  hipMalloc(&g, 512 * 4); hipMalloc(&o, 576 * 4);
  hipMemcpy(g, h.data(), 512 * 4, hipMemcpyHostToDevice);
  k<<<1, 64>>>(g, o);
  std::vector<float> r(576);
  hipMemcpy(r.data(), o, 576 * 4, hipMemcpyDeviceToHost);
  printf("first instr, LDS[0..11]: "); for (int i = 0; i < 12; i++) printf("%g ", r[i]); printf("\n");
  printf("LDS[252..267]: "); for (int i = 252; i < 268; i++) printf("%g ", r[i]); printf("\n");
  printf("second instr, LDS[264..275]: "); for (int i = 264; i < 276; i++) printf("%g ", r[i]); printf("\n");
  int ok1 = 1, ok2 = 1;
  for (int l = 0; l < 64; l++) for (int j = 0; j < 4; j++) { ok1 &= r[l * 4 + j] == (float)((63 - l) * 4 + j); ok2 &= r[264 + l * 4 + j] == (float)(256 + l * 4 + j); }
  printf("lane l -> lds[base + 16 l]: instr1 %s, instr2 (base offset 1056 B) %s\n", ok1 ? "yes" : "NO", ok2 ? "yes" : "NO");
  return 0;
}
