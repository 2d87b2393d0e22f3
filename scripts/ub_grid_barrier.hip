// Cost of a device-wide barrier between resident blocks against the cost of a kernel boundary:
// what a one-kernel-per-token decoder step would pay per phase instead of a launch.
//   hipcc --offload-arch=gfx950 -O3 -o ub_grid_barrier scripts/ub_grid_barrier.hip && ./ub_grid_barrier
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

// One counter, monotonically increasing: barrier number n is passed once the counter reaches
// n * gridDim.x.  The spin is bounded, so every wave leaves the kernel whatever happens.
__device__ __forceinline__ bool grid_barrier(unsigned* counter, unsigned target, unsigned* failed) {
  __syncthreads();
  bool ok = true;
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned spins = 0;
    while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      __builtin_amdgcn_s_sleep(1);
      if (++spins > (1u << 22)) { ok = false; *failed = 1; break; }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
  __syncthreads();
  return ok;
}

// Two levels: blocks are dealt round-robin to the 8 XCDs, so block b first counts on the
// counter of group b % 8 (its own 256-byte line); the last arrival of a group counts on the
// top counter, and everybody polls the top counter.
__device__ __forceinline__ bool grid_barrier2(unsigned* counters, unsigned n, unsigned* failed) {
  __syncthreads();
  bool ok = true;
  if (threadIdx.x == 0) {
    const unsigned group = blockIdx.x & 7u;
    const unsigned in_group = (gridDim.x - group + 7u) / 8u;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    const unsigned mine = __hip_atomic_fetch_add(counters + 64 * (1 + group), 1u, __ATOMIC_RELAXED,
                                                 __HIP_MEMORY_SCOPE_AGENT);
    if (mine + 1 == n * in_group)
      __hip_atomic_fetch_add(counters, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned groups = gridDim.x < 8u ? gridDim.x : 8u;
    unsigned spins = 0;
    while (__hip_atomic_load(counters, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < n * groups) {
      __builtin_amdgcn_s_sleep(1);
      if (++spins > (1u << 22)) { ok = false; *failed = 1; break; }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
  __syncthreads();
  return ok;
}

__global__ __launch_bounds__(256) void barrier_loop2(unsigned* counters, unsigned* failed, float* data,
                                                     int n_barriers, int work) {
  float acc = 0.f;
  for (int n = 1; n <= n_barriers; ++n) {
    for (int w = 0; w < work; ++w)
      acc += data[(blockIdx.x * 256 + threadIdx.x + (n + w) * 4099) & 0xFFFFF];
    if (work) data[(blockIdx.x * 256 + threadIdx.x) & 0xFFFFF] = acc;
    if (!grid_barrier2(counters, (unsigned)n, failed)) return;
  }
}

__global__ __launch_bounds__(256) void barrier_loop(unsigned* counter, unsigned* failed, float* data,
                                                    int n_barriers, int work) {
  float acc = 0.f;
  for (int n = 1; n <= n_barriers; ++n) {
    // a little dependent traffic per phase, so that the barrier really orders memory
    for (int w = 0; w < work; ++w)
      acc += data[(blockIdx.x * 256 + threadIdx.x + (n + w) * 4099) & 0xFFFFF];
    if (work) data[(blockIdx.x * 256 + threadIdx.x) & 0xFFFFF] = acc;
    if (!grid_barrier(counter, (unsigned)n * gridDim.x, failed)) return;
  }
}

__global__ void tiny(float* data) { if (data == nullptr) printf("x"); }

int main() {
  unsigned *counter, *failed;
  float* data;
  hipMalloc(&counter, 4096); hipMalloc(&failed, 4); hipMalloc(&data, 4 << 20);
  hipMemset(data, 0, 4 << 20);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int n_barriers = 2000;
  for (int levels : {1, 2})
  for (int blocks : {256, 512, 1024}) {
    for (int work : {0, 4}) {
      float best = 1e9f;
      unsigned h_failed = 0;
      for (int rep = 0; rep < 3; ++rep) {
        hipMemset(counter, 0, 4096); hipMemset(failed, 0, 4);
        hipEventRecord(e0);
        if (levels == 1)
          hipLaunchKernelGGL(barrier_loop, dim3(blocks), dim3(256), 0, 0, counter, failed, data, n_barriers, work);
        else
          hipLaunchKernelGGL(barrier_loop2, dim3(blocks), dim3(256), 0, 0, counter, failed, data, n_barriers, work);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
        hipMemcpy(&h_failed, failed, 4, hipMemcpyDeviceToHost);
        if (h_failed) break;
      }
      printf("%d-level, %4d blocks, work %d: %.2f us per barrier%s\n", levels, blocks, work, 1e3f * best / n_barriers,
             h_failed ? "  (TIMED OUT: blocks not co-resident)" : "");
      if (h_failed) return 1;
    }
  }
  // the kernel boundary it would replace: dependent one-thread launches
  float best = 1e9f;
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    for (int i = 0; i < 2000; ++i) hipLaunchKernelGGL(tiny, dim3(1), dim3(1), 0, 0, data);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    best = ms < best ? ms : best;
  }
  printf("kernel boundary (one-thread kernels back to back): %.2f us per launch\n", 1e3f * best / 2000);
  return 0;
}
