// Issue cost of the instructions a two-lanes-per-frame log-mel kernel would be made of, with one
// and with two waves per SIMD (256-thread blocks, occupancy set through the LDS size), every CU busy.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
#define REP 256
#define OUTER 512
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, unsigned long long* cyc, float c0) {
  extern __shared__ float lds[];
  float a[8];
  for (int i = 0; i < 8; ++i) a[i] = (float)threadIdx.x * 0.001f + i + 1.0f;
  v2f p[8];
  for (int i = 0; i < 8; ++i) p[i] = v2f{a[i], a[i] + 0.5f};
  if (threadIdx.x == 0) lds[0] = c0;
  __syncthreads();
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  // the compiler does the masking: a divergent branch around the whole measured loop
  const bool on = MODE == 1 ? (threadIdx.x & 63) < 32 : MODE == 2 ? (threadIdx.x & 1) == 0 : true;
  if (on) {
  asm volatile("" ::: "memory");
#pragma unroll 1
  for (int r = 0; r < OUTER; ++r) {
#pragma unroll
    for (int i = 0; i < REP / 8; ++i) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        if (MODE <= 2) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[j]) : "s"(c0), "v"(a[(j + 1) & 7]));
        else if (MODE == 3) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[j]) : "v"(p[(j + 3) & 7]), "v"(p[(j + 1) & 7]));
        else if (MODE == 4) asm volatile("v_mul_f32_dpp %0, %1, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "=v"(a[j]) : "v"(a[(j + 1) & 7]), "v"(a[(j + 2) & 7]));
        else if (MODE == 5) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(a[j]), "+v"(a[(j + 4) & 7]));
        else if (MODE == 6) asm volatile("v_sqrt_f32 %0, %1" : "=v"(a[j]) : "v"(a[(j + 1) & 7]));
        else if (MODE == 7) asm volatile("v_log_f32 %0, %1" : "=v"(a[j]) : "v"(a[(j + 1) & 7]));
        else if (MODE == 8) asm volatile("v_fmac_f32_dpp %0, %1, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(a[j]) : "v"(a[(j + 1) & 7]), "v"(a[(j + 2) & 7]));
        else if (MODE == 9) asm volatile("v_add_f32 %0, %1, %2" : "=v"(a[j]) : "v"(a[(j + 1) & 7]), "v"(a[(j + 2) & 7]));
        else if (MODE == 10) asm volatile("v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "=v"(a[j]) : "v"(a[(j + 1) & 7]));
      }
    }
  }
  }
  asm volatile("" ::: "memory");
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
  for (int i = 0; i < 8; ++i) s += a[i] + p[i].x + p[i].y;
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[MODE] = t1 - t0;
}
template <int MODE>
void run(const char* name, float* out, unsigned long long* cyc) {
  for (int wps = 1; wps <= 2; ++wps) {
    const size_t lds = wps == 1 ? 100 * 1024 : 70 * 1024;
    hipFuncSetAttribute((const void*)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    k<MODE><<<256 * wps, 256, lds>>>(out, cyc, 0.999f);
    hipEventRecord(a);
    k<MODE><<<256 * wps, 256, lds>>>(out, cyc, 0.999f);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    unsigned long long h; hipMemcpy(&h, cyc + MODE, 8, hipMemcpyDeviceToHost);
    const double n = (double)OUTER * REP;
    printf("%-28s %d wave/SIMD: %6.2f memtime ticks per wave-instr, wall %7.3f ms -> %5.2f ns per instr per SIMD (all waves)\n",
           name, wps, h / n, ms, ms * 1e6 / (n * wps));
  }
}
int main() {
  float* out; unsigned long long* cyc; hipMalloc(&out, 512 * 256 * 4); hipMalloc(&cyc, 128);
  run<0>("v_fma_f32 full exec", out, cyc);
  run<1>("v_fma_f32 exec=low32", out, cyc);
  run<2>("v_fma_f32 exec=even lanes", out, cyc);
  run<9>("v_add_f32", out, cyc);
  run<3>("v_pk_fma_f32", out, cyc);
  run<4>("v_mul_f32_dpp quad_perm", out, cyc);
  run<8>("v_fmac_f32_dpp quad_perm", out, cyc);
  run<10>("v_mov_b32_dpp quad_perm", out, cyc);
  run<5>("v_permlane32_swap_b32", out, cyc);
  run<6>("v_sqrt_f32", out, cyc);
  run<7>("v_log_f32", out, cyc);
  return 0;
}
