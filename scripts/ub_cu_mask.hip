// Which compute units does a stream created with hipExtStreamCreateWithCUMask use?
// Every block records the XCC and the (SE, CU) it ran on; the host prints, per mask, how many
// distinct CUs of every XCC were touched.  Build: hipcc --offload-arch=gfx950 -O2 -o ub_cu_mask ub_cu_mask.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <set>
#include <map>
#include <vector>

__global__ void where_kernel(uint32_t* out, int spin) {
  const uint32_t hw = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11));      // HW_ID
  const uint32_t xcc = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (31 << 11));    // XCC_ID
  if (threadIdx.x == 0) { out[2 * blockIdx.x] = hw; out[2 * blockIdx.x + 1] = xcc; }
  // keep the CU busy for a while so that the blocks spread over everything the mask allows
  float x = threadIdx.x;
  for (int i = 0; i < spin; ++i) x = x * 1.0001f + 0.5f;
  if (x == 12345.f) out[0] = 0;
}

static void run(const char* name, const std::vector<uint32_t>& mask) {
  hipStream_t st = nullptr;
  printf("%s ...\n", name);
  if (hipExtStreamCreateWithCUMask(&st, (uint32_t)mask.size(), mask.data()) != hipSuccess) { printf("%s: create failed\n", name); return; }
  const int blocks = 4096;
  uint32_t* d;
  hipMalloc(&d, blocks * 8);
  hipLaunchKernelGGL(where_kernel, dim3(blocks), dim3(256), 0, st, d, 20000);
  hipStreamSynchronize(st);
  std::vector<uint32_t> h(2 * blocks);
  hipMemcpy(h.data(), d, blocks * 8, hipMemcpyDeviceToHost);
  std::map<int, std::set<int>> cus;
  for (int b = 0; b < blocks; ++b) {
    const uint32_t hw = h[2 * b], xcc = h[2 * b + 1] & 15;
    const int cu = (hw >> 8) & 15, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
    cus[xcc].insert(se * 32 + sh * 16 + cu);
  }
  printf("%s:", name);
  int total = 0;
  for (auto& kv : cus) { printf(" xcc%d=%zu", kv.first, kv.second.size()); total += (int)kv.second.size(); }
  printf("  total %d CUs\n", total);
  if (cus.size() && cus.begin()->second.size() <= 32) {
    printf("   xcc%d ids:", cus.begin()->first);
    for (int id : cus.begin()->second) printf(" %d.%d.%d", id / 32, (id / 16) & 1, id & 15);
    printf("\n");
  }
  hipFree(d);
  hipStreamDestroy(st);
}

int main() {
  setvbuf(stdout, nullptr, _IONBF, 0);
  hipSetDevice(0);
  hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
  printf("device %s, %d CUs\n", prop.name, prop.multiProcessorCount);
  // Bit i of the mask is CU (i / 32) of shader engine (i / 8) % 4 of XCC i % 8: a mask of the low
  // 128 bits leaves every XCC half of its CUs (the first four of each shader engine).  A mask that
  // switches whole XCCs off (0x55555555 in every word) crashed inside the runtime on ROCm 7.2.
  std::vector<uint32_t> all(8, 0xFFFFFFFFu), low(8, 0), first32(8, 0);
  for (int i = 0; i < 4; ++i) low[i] = 0xFFFFFFFFu;
  first32[0] = 0xFFFFFFFFu;
  run("all 256", all);
  run("bits 0-127", low);
  run("bits 0-31", first32);
  return 0;
}
