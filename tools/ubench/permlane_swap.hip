// Prints what v_permlane32_swap_b32 and v_permlane16_swap_b32 do on gfx950: lane l holds a = l and b = 100 + l.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* out) {
  const unsigned a = threadIdx.x, b = 100 + threadIdx.x;
  const auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
  const auto s = __builtin_amdgcn_permlane16_swap(a, b, false, false);
  out[threadIdx.x] = r[0];
  out[64 + threadIdx.x] = r[1];
  out[128 + threadIdx.x] = s[0];
  out[192 + threadIdx.x] = s[1];
}
int main() {
  unsigned* d;
  (void)hipMalloc(&d, 256 * 4);
  k<<<1, 64>>>(d);
  unsigned h[256];
  (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  const char* names[4] = {"permlane32_swap result 0", "permlane32_swap result 1", "permlane16_swap result 0", "permlane16_swap result 1"};
  for (int v = 0; v < 4; ++v) {
    printf("%s (first lane of each row of 16):", names[v]);
    for (int r = 0; r < 4; ++r) printf(" %u", h[64 * v + 16 * r]);
    printf("\n");
  }
  return 0;
}
