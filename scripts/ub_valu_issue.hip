#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
#define REP 512
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, unsigned* cyc, v2f c0, v2f c1) {
  v2f a[8];
  for (int i = 0; i < 8; ++i) a[i] = v2f{(float)threadIdx.x + i, 1.0f + i};
  v2f cv = c1;
  asm volatile("" : "+v"(cv));
  __syncthreads();
  unsigned t0 = (unsigned)__builtin_amdgcn_s_memtime();
  asm volatile("" ::: "memory");
#pragma unroll 1
  for (int r = 0; r < 64; ++r) {
#pragma unroll
    for (int i = 0; i < REP / 64; ++i) {
      if (MODE == 0) {        // 8 independent chains, SGPR constant operands
#pragma unroll
        for (int j = 0; j < 8; ++j) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[j]) : "s"(c0), "v"(cv));
      } else if (MODE == 1) { // one dependent chain
#pragma unroll
        for (int j = 0; j < 8; ++j) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[0]) : "s"(c0), "v"(cv));
      } else if (MODE == 2) { // 2 chains alternating
#pragma unroll
        for (int j = 0; j < 8; ++j) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[j & 1]) : "s"(c0), "v"(cv));
      } else if (MODE == 3) { // independent, all-VGPR operands
#pragma unroll
        for (int j = 0; j < 8; ++j) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[j]) : "v"(a[(j + 1) & 7]), "v"(a[(j + 2) & 7]));
      } else if (MODE == 4) { // plain v_fma_f32 independent
#pragma unroll
        for (int j = 0; j < 8; ++j) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[j].x) : "s"(c0.x), "v"(cv.x));
      } else if (MODE == 5) { // pk with op_sel / neg modifiers
#pragma unroll
        for (int j = 0; j < 8; ++j) asm volatile("v_pk_fma_f32 %0, %0, %1, %2 op_sel_hi:[1,0,1] neg_lo:[0,0,1] neg_hi:[0,0,1]" : "+v"(a[j]) : "s"(c0), "v"(a[(j + 3) & 7]));
      } else if (MODE == 6) { // 4 chains
#pragma unroll
        for (int j = 0; j < 8; ++j) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[j & 3]) : "s"(c0), "v"(cv));
      } else if (MODE == 7) { // accvgpr write/read pairs
#pragma unroll
        for (int j = 0; j < 8; ++j) { float t; asm volatile("v_accvgpr_write_b32 a0, %1\n\tv_accvgpr_read_b32 %0, a1" : "=v"(t) : "v"(a[j].x) : "a0"); a[j].y += 0.f * t; }
      }
    }
  }
  asm volatile("" ::: "memory");
  unsigned t1 = (unsigned)__builtin_amdgcn_s_memtime();
  float s = 0; for (int i = 0; i < 8; ++i) s += a[i].x + a[i].y;
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[MODE] = t1 - t0;
}
int main() {
  float* out; unsigned* cyc; hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 64);
  hipMemset(cyc, 0, 64);
  v2f c0 = {0.999f, 1.001f}, c1 = {0.001f, -0.001f};
  k<0><<<256, 256>>>(out, cyc, c0, c1); k<1><<<256, 256>>>(out, cyc, c0, c1); k<2><<<256, 256>>>(out, cyc, c0, c1);
  k<3><<<256, 256>>>(out, cyc, c0, c1); k<4><<<256, 256>>>(out, cyc, c0, c1); k<5><<<256, 256>>>(out, cyc, c0, c1);
  k<6><<<256, 256>>>(out, cyc, c0, c1); k<7><<<256, 256>>>(out, cyc, c0, c1);
  hipDeviceSynchronize();
  unsigned h[16]; hipMemcpy(h, cyc, 64, hipMemcpyDeviceToHost);
  const char* names[] = {"pk_fma 8 indep (sgpr ops)", "pk_fma dependent chain", "pk_fma 2 chains", "pk_fma indep all-vgpr",
                         "v_fma_f32 indep", "pk_fma op_sel/neg + vgpr", "pk_fma 4 chains", "accvgpr write+read (+1 fma)"};
  for (int i = 0; i < 8; ++i) printf("%-32s %.2f ticks per instruction (%u)\n", names[i], h[i] / (64.0 * REP / 64 * 8), h[i]);
  return 0;
}
