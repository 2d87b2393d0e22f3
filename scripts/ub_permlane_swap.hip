#include <hip/hip_runtime.h>
__global__ void k(unsigned* out) {
  unsigned a = threadIdx.x, b = 1000 + threadIdx.x;
  auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
  out[threadIdx.x] = r[0];
  out[64 + threadIdx.x] = r[1];
}
int main() {
  unsigned* d; hipMalloc(&d, 512); k<<<1, 64>>>(d);
  unsigned h[128]; hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
  printf("vdst: lane0 %u lane31 %u lane32 %u lane63 %u\n", h[0], h[31], h[32], h[63]);
  printf("src0: lane0 %u lane31 %u lane32 %u lane63 %u\n", h[64], h[95], h[96], h[127]);
}
