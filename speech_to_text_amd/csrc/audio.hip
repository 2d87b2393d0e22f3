// Audio pre-steps of the Whisper wrapper on the GPU (gfx950): what the reference does on the host
// between reading a file and the mel (utils/audio/audio_io.py:249-264).
//   stt_pcm_normalize    utils/audio/audio_processing.py:48-60  normalize_audio(max_val = 1.):
//                        zero mean, peak scaled to 1, float32 out.  int16 input is reduced and
//                        scaled in float64 exactly as numpy does (the sum of int16 samples is an
//                        exact integer), so the result is bit-identical to the host function.
//   stt_pcm_block_peaks  utils/audio/audio_processing.py:120-127  the per-block peak of x^2 that
//                        decides which 10 ms blocks are silent; the run-length logic over the
//                        (clip / 10 ms) flags stays on the host.
// All three passes of the normalisation stream the clip at the HBM rate; partial results of the
// grid are reduced by the last block to arrive (ticket counter in the workspace).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/stt_mi355x.h"
#include "stt_common.h"

namespace {

constexpr int kBlocks = 1024;          // partial results per pass
constexpr int kThreads = 256;

struct NormWs {                        // layout of the caller's workspace (device)
  double partial[kBlocks];
  double mean;
  double gain;
  unsigned ticket[2];
};

template <typename T>
__device__ __forceinline__ double load_as_double(const T* p, long i) { return (double)p[i]; }

__device__ __forceinline__ double block_sum(double v, double* sh) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  v = sh[0] + sh[1] + sh[2] + sh[3];
  __syncthreads();
  return v;
}
__device__ __forceinline__ double block_max(double v, double* sh) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o, 64));
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  v = fmax(fmax(sh[0], sh[1]), fmax(sh[2], sh[3]));
  __syncthreads();
  return v;
}

// true in exactly one block: the last one to arrive.  Its view of the other blocks' partials is
// made current by the release / acquire pair around the ticket.
__device__ __forceinline__ bool last_block(unsigned* ticket, int* sh_flag) {
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    const unsigned t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    *sh_flag = t == gridDim.x - 1;
    if (*sh_flag) {
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next call
    }
  }
  __syncthreads();
  return *sh_flag != 0;
}

// pass 1: mean.  int16 samples are summed as integers (exact), float samples in float64.
template <typename T>
__global__ __launch_bounds__(kThreads) void pcm_sum_kernel(const T* __restrict__ x, long n, NormWs* ws) {
  __shared__ double sh[4];
  __shared__ int flag;
  double acc = 0.0;
  if (sizeof(T) == 2) {
    long long s = 0;
    for (long i = (long)blockIdx.x * kThreads + threadIdx.x; i < n; i += (long)gridDim.x * kThreads) s += (long long)x[i];
    acc = (double)s;                   // |s| < 2^15 * n / grid: exact in float64 for any clip
  } else {
    for (long i = (long)blockIdx.x * kThreads + threadIdx.x; i < n; i += (long)gridDim.x * kThreads) acc += load_as_double(x, i);
  }
  acc = block_sum(acc, sh);
  if (threadIdx.x == 0) __hip_atomic_store(&ws->partial[blockIdx.x], acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (!last_block(&ws->ticket[0], &flag)) return;
  double t = 0.0;
  for (int i = threadIdx.x; i < (int)gridDim.x; i += kThreads)
    t += __hip_atomic_load(&ws->partial[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  t = block_sum(t, sh);
  if (threadIdx.x == 0) ws->mean = t / (double)n;
}

// pass 2: peak of |x - mean|, then gain = 1 / peak (1 when the clip is flat: peak <= 1e-9)
template <typename T>
__global__ __launch_bounds__(kThreads) void pcm_peak_kernel(const T* __restrict__ x, long n, NormWs* ws) {
  __shared__ double sh[4];
  __shared__ int flag;
  const double mean = ws->mean;
  double acc = 0.0;
  for (long i = (long)blockIdx.x * kThreads + threadIdx.x; i < n; i += (long)gridDim.x * kThreads)
    acc = fmax(acc, fabs(load_as_double(x, i) - mean));
  acc = block_max(acc, sh);
  if (threadIdx.x == 0) __hip_atomic_store(&ws->partial[blockIdx.x], acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (!last_block(&ws->ticket[1], &flag)) return;
  double t = 0.0;
  for (int i = threadIdx.x; i < (int)gridDim.x; i += kThreads)
    t = fmax(t, __hip_atomic_load(&ws->partial[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
  t = block_max(t, sh);
  if (threadIdx.x == 0) ws->gain = t > 1e-9 ? 1.0 / t : 1.0;
}

// pass 3: out = float32((x - mean) * gain), the product in float64 like numpy's
template <typename T>
__global__ __launch_bounds__(kThreads) void pcm_scale_kernel(const T* __restrict__ x, long n, const NormWs* ws,
                                                             float* __restrict__ out) {
  const double mean = ws->mean, gain = ws->gain;
  for (long i = (long)blockIdx.x * kThreads + threadIdx.x; i < n; i += (long)gridDim.x * kThreads)
    out[i] = (float)((load_as_double(x, i) - mean) * gain);
}

// float32 clips: numpy stays in float32 here (mean by pairwise float32 sums, x - mean, x * (1 / peak)
// all float32), which no parallel sum reproduces bit for bit; the mean is taken in float64 and
// rounded once, the rest is done in float32 in numpy's order.
__global__ __launch_bounds__(kThreads) void pcm_scale_f32_kernel(const float* __restrict__ x, long n,
                                                                 const NormWs* ws, float* __restrict__ out) {
  const float mean = (float)ws->mean;
  const float gain = (float)ws->gain;
  for (long i = (long)blockIdx.x * kThreads + threadIdx.x; i < n; i += (long)gridDim.x * kThreads)
    out[i] = (x[i] - mean) * gain;
}
__global__ __launch_bounds__(kThreads) void pcm_peak_f32_kernel(const float* __restrict__ x, long n, NormWs* ws) {
  __shared__ double sh[4];
  __shared__ int flag;
  const float mean = (float)ws->mean;
  float acc = 0.f;
  for (long i = (long)blockIdx.x * kThreads + threadIdx.x; i < n; i += (long)gridDim.x * kThreads)
    acc = fmaxf(acc, fabsf(x[i] - mean));
  double a = block_max((double)acc, sh);
  if (threadIdx.x == 0) __hip_atomic_store(&ws->partial[blockIdx.x], a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (!last_block(&ws->ticket[1], &flag)) return;
  double t = 0.0;
  for (int i = threadIdx.x; i < (int)gridDim.x; i += kThreads)
    t = fmax(t, __hip_atomic_load(&ws->partial[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
  t = block_max(t, sh);
  // numpy: audio * (max_val / max_audio_val) with both float32
  if (threadIdx.x == 0) ws->gain = t > 1e-9 ? (double)(1.0f / (float)t) : 1.0;
}

// one wave per block of `block` samples: peak of x * x (float32 product, as numpy's `** 2`)
__global__ __launch_bounds__(kThreads) void pcm_block_peaks_kernel(const float* __restrict__ x, long n, int block,
                                                                   long n_blocks, float* __restrict__ peaks) {
  const int lane = threadIdx.x & 63;
  for (long b = (long)blockIdx.x * 4 + (threadIdx.x >> 6); b < n_blocks; b += (long)gridDim.x * 4) {
    const long lo = b * block;
    long hi = lo + block;
    hi = hi < n ? hi : n;
    float m = 0.f;                      // the padding of a ragged last block is zeros
    for (long i = lo + lane; i < hi; i += 64) m = fmaxf(m, x[i] * x[i]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    if (lane == 0) peaks[b] = m;
  }
}

int grid_for(int64_t n) {
  const int64_t g = (n + kThreads - 1) / kThreads;
  return (int)(g < 1 ? 1 : g > kBlocks ? kBlocks : g);
}

}  // namespace

extern "C" {

size_t stt_pcm_workspace_bytes(void) { return sizeof(NormWs); }

int stt_pcm_normalize(const void* d_in, int dtype, int64_t n, float* d_out, void* d_workspace,
                      size_t workspace_bytes, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!d_in || !d_out || !d_workspace) return stt_fail(STT_ERR_INVALID_ARGUMENT, "null argument");
  if (n <= 0) return stt_fail(STT_ERR_INVALID_ARGUMENT, "empty clip");
  if (dtype != STT_PCM_F32 && dtype != STT_PCM_I16 && dtype != STT_PCM_F64)
    return stt_fail(STT_ERR_UNSUPPORTED, "pcm dtype %d: float32 (0), int16 (1) or float64 (2)", dtype);
  if (workspace_bytes < sizeof(NormWs))
    return stt_fail(STT_ERR_WORKSPACE, "workspace too small: %zu < %zu", workspace_bytes, sizeof(NormWs));
  if ((uintptr_t)d_workspace & 7) return stt_fail(STT_ERR_INVALID_ARGUMENT, "workspace must be 8-byte aligned");
  NormWs* ws = (NormWs*)d_workspace;
  // the two ticket counters start at zero: they are reset by the block that draws the last ticket,
  // but a fresh workspace holds garbage
  STT_HIP_OR_FAIL(hipMemsetAsync(ws->ticket, 0, sizeof(ws->ticket), stream), (void)0);
  const int grid = grid_for(n);
  if (dtype == STT_PCM_I16) {
    const int16_t* x = (const int16_t*)d_in;
    hipLaunchKernelGGL(pcm_sum_kernel<int16_t>, dim3(grid), dim3(kThreads), 0, stream, x, (long)n, ws);
    hipLaunchKernelGGL(pcm_peak_kernel<int16_t>, dim3(grid), dim3(kThreads), 0, stream, x, (long)n, ws);
    hipLaunchKernelGGL(pcm_scale_kernel<int16_t>, dim3(grid), dim3(kThreads), 0, stream, x, (long)n, ws, d_out);
  } else if (dtype == STT_PCM_F64) {
    // what scipy's resampler hands normalize_audio: every step in float64, one cast at the end
    const double* x = (const double*)d_in;
    hipLaunchKernelGGL(pcm_sum_kernel<double>, dim3(grid), dim3(kThreads), 0, stream, x, (long)n, ws);
    hipLaunchKernelGGL(pcm_peak_kernel<double>, dim3(grid), dim3(kThreads), 0, stream, x, (long)n, ws);
    hipLaunchKernelGGL(pcm_scale_kernel<double>, dim3(grid), dim3(kThreads), 0, stream, x, (long)n, ws, d_out);
  } else {
    const float* x = (const float*)d_in;
    hipLaunchKernelGGL(pcm_sum_kernel<float>, dim3(grid), dim3(kThreads), 0, stream, x, (long)n, ws);
    hipLaunchKernelGGL(pcm_peak_f32_kernel, dim3(grid), dim3(kThreads), 0, stream, x, (long)n, ws);
    hipLaunchKernelGGL(pcm_scale_f32_kernel, dim3(grid), dim3(kThreads), 0, stream, x, (long)n, ws, d_out);
  }
  STT_HIP_OR_FAIL(hipGetLastError(), (void)0);
  return STT_OK;
}

int stt_pcm_block_peaks(const float* d_pcm, int64_t n, int block_size, float* d_peaks, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!d_pcm || !d_peaks) return stt_fail(STT_ERR_INVALID_ARGUMENT, "null argument");
  if (n <= 0 || block_size <= 0) return stt_fail(STT_ERR_INVALID_ARGUMENT, "empty clip or block");
  const int64_t n_blocks = (n + block_size - 1) / block_size;
  int64_t grid = (n_blocks + 3) / 4;
  grid = grid > 4096 ? 4096 : grid;
  hipLaunchKernelGGL(pcm_block_peaks_kernel, dim3((int)grid), dim3(kThreads), 0, stream, d_pcm, (long)n,
                     block_size, (long)n_blocks, d_peaks);
  STT_HIP_OR_FAIL(hipGetLastError(), (void)0);
  return STT_OK;
}

}  // extern "C"
