// The device allocations (sizes, order, memsets) of a 512-window large-v3 engine and nothing else:
// run once under `rocprofv3 --kernel-trace` to see whether the profiler's crash on that engine
// needs anything but the allocations.   hipcc --offload-arch=gfx950 -o repro_prof_alloc repro_prof_alloc.hip
// Usage: repro_prof_alloc [windows=512] [decoder layers=32] [memset 0|1]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void touch(float* p) { p[threadIdx.x] = 1.f; }
int main(int argc, char** argv) {
  const size_t B = argc > 1 ? atoi(argv[1]) : 512, LD = argc > 2 ? atoi(argv[2]) : 32;
  const int do_memset = argc > 3 ? atoi(argv[3]) : 1;
  const size_t d = 1280, H = 20, T = 1500, M = 128, R = B * T;
  std::vector<void*> ptrs;
  auto alloc = [&](size_t bytes, bool zero) -> int {
    void* p = nullptr;
    CK(hipMalloc(&p, (bytes + 255) & ~(size_t)255));
    if (zero && do_memset) CK(hipMemset(p, 0, bytes));
    ptrs.push_back(p);
    return 0;
  };
  // weights: 32 + 32 layers, ~3.1 GB in ~1200 pieces
  for (int l = 0; l < 32 + (int)LD; ++l)
    for (size_t n : {3 * d * d, d * d, 4 * d * d, 4 * d * d, d * d, 2 * d * d, d * d})
      if (alloc(n * 2, false)) return 1;
  printf("weights ok\n"); fflush(stdout);
  const size_t pitch = (B * T + 127) / 64 * 64;
  if (alloc((B * 3002 * M + 4096) * 2, true) || alloc((B * 3002 * d + 4096) * 2, true) || alloc(R * d * 4, false) ||
      alloc(R * d * 2, false) || alloc(R * 3 * d * 2, false) || alloc(H * 64 * pitch * 2, true) ||
      alloc(R * d * 2, false) || alloc(R * 4 * d * 2, false) || alloc(R * d * 2, false)) return 1;
  printf("encoder buffers ok\n"); fflush(stdout);
  for (size_t l = 0; l < 2 * LD; ++l) if (alloc(B * H * T * 64 * 2, false)) return 1;
  printf("cross K/V ok (%zu x %.2f GB)\n", 2 * LD, B * H * T * 64 * 2 / 1e9); fflush(stdout);
  for (size_t l = 0; l < 2 * LD; ++l) if (alloc(B * H * 136 * 64 * 2, false)) return 1;
  if (alloc(B * 51866 * 4, false)) return 1;
  hipLaunchKernelGGL(touch, dim3(1), dim3(64), 0, nullptr, (float*)ptrs.back());
  CK(hipDeviceSynchronize());
  printf("all %zu allocations ok\n", ptrs.size()); fflush(stdout);
  for (void* p : ptrs) (void)hipFree(p);
  return 0;
}
