// probe: semantics of v_permlane32_swap / v_permlane16_swap on gfx950 (prints the lane layout)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned *out) {
  const unsigned a = 1000 + threadIdx.x, b = 2000 + threadIdx.x;
  auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
  out[threadIdx.x] = r[0]; out[64 + threadIdx.x] = r[1];
  auto s = __builtin_amdgcn_permlane16_swap(a, b, false, false);
  out[128 + threadIdx.x] = s[0]; out[192 + threadIdx.x] = s[1];
}
int main() {
  unsigned *d, h[256];
  hipMalloc(&d, sizeof(h));
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  const char *nm[4] = {"p32[0]", "p32[1]", "p16[0]", "p16[1]"};
  for (int q = 0; q < 4; q++) { printf("%s:", nm[q]); for (int i = 0; i < 64; i += 8) printf(" %u", h[q*64 + i]); printf("\n"); }
  return 0;
}
