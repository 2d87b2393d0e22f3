// What the matrix cores sustain on a fully loaded chip with no memory traffic at all: every wave
// issues independent v_mfma_f32_16x16x32_bf16 from registers.  Calibrates the distance between the
// GEMM main loops and the nominal 2.5 PFLOP/s.
//   hipcc --offload-arch=gfx950 -O3 -o ub_mfma_peak scripts/ub_mfma_peak.hip && ./ub_mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf8_t;
typedef __attribute__((ext_vector_type(4))) float f4_t;

template <int NACC>
__global__ __launch_bounds__(256) void mfma_loop(float* out, int iters) {
  f4_t acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = f4_t{0.f, 0.f, 0.f, 0.f};
  bf8_t a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(float)(threadIdx.x + i); b[i] = (__bf16)(float)(i + 1); }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NACC>
static void run(const char* what, int blocks, int threads, float* out) {
  const int iters = 4000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e9f;
  for (int rep = 0; rep < 4; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(mfma_loop<NACC>, dim3(blocks), dim3(threads), 0, 0, out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    best = ms < best ? ms : best;
  }
  const double flop = (double)blocks * (threads / 64) * iters * NACC * 16.0 * 16 * 32 * 2;
  printf("%-34s %7.3f ms  %7.1f TFLOP/s\n", what, best, flop / best / 1e9);
}

int main() {
  float* out;
  hipMalloc(&out, 4096 * 512 * 4);
  run<16>("1 wave / SIMD, 16 accumulators", 256, 256, out);
  run<16>("2 waves / SIMD, 16 accumulators", 512, 256, out);
  run<16>("4 waves / SIMD, 16 accumulators", 1024, 256, out);
  run<4>("2 waves / SIMD, 4 accumulators", 512, 256, out);
  // a long run: the clock a sustained load settles at
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  for (int i = 0; i < 60; ++i) hipLaunchKernelGGL(mfma_loop<16>, dim3(512), dim3(256), 0, 0, out, 4000);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("%-34s %7.3f ms  %7.1f TFLOP/s\n", "sustained (60 launches back to back)", ms,
         60.0 * 512 * 4 * 4000 * 16 * 16384.0 / ms / 1e9);
  return 0;
}
