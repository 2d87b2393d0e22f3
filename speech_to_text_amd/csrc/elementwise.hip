// Normalisation, embedding and layout kernels of the Whisper engine (gfx950).  All of them are
// HBM-bound row kernels: one wave per row, 16-byte accesses.
#include "kernels.h"

__device__ __forceinline__ bf16_t f2bf(float f) {
  uint32_t u = __float_as_uint(f);
  u += 0x7FFFu + ((u >> 16) & 1u);
  return (bf16_t)(u >> 16);
}
__device__ __forceinline__ float bf2f(bf16_t h) { return __uint_as_float(((uint32_t)h) << 16); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// x += sum of pending partial sums (+ bias); out = LayerNorm(x) (keras LayerNormalization,
// epsilon inside the square root; residual_multi_head_attention.py:220-225,
// transformer_arch.py:480-481, 593-597).  One wave per row, d <= 64 * 4 * LN_MAX_V4.
#define LN_MAX_V4 8
__global__ __launch_bounds__(256) void add_layernorm_kernel(
    float* __restrict__ x, const float* __restrict__ slabs, int n_slabs, long slab_stride,
    const float* __restrict__ add_bias, const float* __restrict__ gamma,
    const float* __restrict__ beta, bf16_t* __restrict__ out_bf16, float* __restrict__ out_f32,
    int m, int d, float eps, int row_mul, int row_off) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= m) return;
  const int nv = d >> 2;
  float4 v[LN_MAX_V4];
  const long in_row = (long)row * row_mul + row_off;
  float* xr = x + in_row * d;
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < LN_MAX_V4; ++i) {
    const int c = lane + 64 * i;
    if (c < nv) {
      float4 t = ((const float4*)xr)[c];
      if (n_slabs > 0) {
        // four slabs requested at a time, added in the order a plain loop would add them
        for (int s0 = 0; s0 < n_slabs; s0 += 4) {
          float4 p[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int ss = s0 + j < n_slabs ? s0 + j : n_slabs - 1;
            p[j] = ((const float4*)(slabs + ss * slab_stride + in_row * d))[c];
          }
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (s0 + j < n_slabs) { t.x += p[j].x; t.y += p[j].y; t.z += p[j].z; t.w += p[j].w; }
        }
        if (add_bias) {
          const float4 b = ((const float4*)add_bias)[c];
          t.x += b.x; t.y += b.y; t.z += b.z; t.w += b.w;
        }
        ((float4*)xr)[c] = t;
      }
      v[i] = t;
      sum += t.x + t.y + t.z + t.w;
    }
  }
  const float mean = wave_sum(sum) / d;
  float sq = 0.f;
#pragma unroll
  for (int i = 0; i < LN_MAX_V4; ++i) {
    const int c = lane + 64 * i;
    if (c < nv) {
      const float a = v[i].x - mean, b = v[i].y - mean, cc = v[i].z - mean, e = v[i].w - mean;
      sq += a * a + b * b + cc * cc + e * e;
    }
  }
  const float rstd = rsqrtf(wave_sum(sq) / d + eps);
#pragma unroll
  for (int i = 0; i < LN_MAX_V4; ++i) {
    const int c = lane + 64 * i;
    if (c < nv) {
      const float4 g = ((const float4*)gamma)[c], b = ((const float4*)beta)[c];
      float4 o;
      o.x = (v[i].x - mean) * rstd * g.x + b.x;
      o.y = (v[i].y - mean) * rstd * g.y + b.y;
      o.z = (v[i].z - mean) * rstd * g.z + b.z;
      o.w = (v[i].w - mean) * rstd * g.w + b.w;
      if (out_bf16) {
        uint2 p;
        p.x = (uint32_t)f2bf(o.x) | ((uint32_t)f2bf(o.y) << 16);
        p.y = (uint32_t)f2bf(o.z) | ((uint32_t)f2bf(o.w) << 16);
        ((uint2*)(out_bf16 + (long)row * d))[c] = p;
      }
      if (out_f32) ((float4*)(out_f32 + (long)row * d))[c] = o;
    }
  }
}

// Same computation with one 256-thread block per row: decode steps have few rows, and a row's
// pending partial sums (up to tens of KB) are then read by 256 lanes instead of 64.
__global__ __launch_bounds__(256) void add_layernorm_row_kernel(
    float* __restrict__ x, const float* __restrict__ slabs, int n_slabs, long slab_stride,
    const float* __restrict__ add_bias, const float* __restrict__ gamma,
    const float* __restrict__ beta, bf16_t* __restrict__ out_bf16, float* __restrict__ out_f32,
    int m, int d, float eps, int row_mul, int row_off) {
  __shared__ float sh[8];
  const int row = blockIdx.x, tid = threadIdx.x;
  const int nv = d >> 2;
  const long in_row = (long)row * row_mul + row_off;
  float* xr = x + in_row * d;
  float4 v[2], gm[2], bt[2];
  // Every load of the kernel is issued before the first wait: the row, its pending partial sums,
  // the bias and the scale / shift vectors (needed last) all travel in one memory latency.
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int c = tid + 256 * i;
    if (c < nv) {
      gm[i] = ((const float4*)gamma)[c];
      bt[i] = ((const float4*)beta)[c];
    }
  }
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int c = tid + 256 * i;
    if (c < nv) {
      float4 t = ((const float4*)xr)[c];
      if (n_slabs > 0) {
        // at most 8 slabs (pick_split_k): all of them requested before the first is added, in the
        // same order as a plain loop would add them
        float4 p[8];
#pragma unroll
        for (int s = 0; s < 8; ++s) {
          const int ss = s < n_slabs ? s : n_slabs - 1;
          p[s] = ((const float4*)(slabs + ss * slab_stride + in_row * d))[c];
        }
#pragma unroll
        for (int s = 0; s < 8; ++s)
          if (s < n_slabs) { t.x += p[s].x; t.y += p[s].y; t.z += p[s].z; t.w += p[s].w; }
        if (add_bias) {
          const float4 b = ((const float4*)add_bias)[c];
          t.x += b.x; t.y += b.y; t.z += b.z; t.w += b.w;
        }
        ((float4*)xr)[c] = t;
      }
      v[i] = t;
      sum += t.x + t.y + t.z + t.w;
    }
  }
  sum = wave_sum(sum);
  if ((tid & 63) == 0) sh[tid >> 6] = sum;
  __syncthreads();
  const float mean = (sh[0] + sh[1] + sh[2] + sh[3]) / d;
  float sq = 0.f;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int c = tid + 256 * i;
    if (c < nv) {
      const float a = v[i].x - mean, b = v[i].y - mean, cc = v[i].z - mean, e = v[i].w - mean;
      sq += a * a + b * b + cc * cc + e * e;
    }
  }
  sq = wave_sum(sq);
  if ((tid & 63) == 0) sh[4 + (tid >> 6)] = sq;
  __syncthreads();
  const float rstd = rsqrtf((sh[4] + sh[5] + sh[6] + sh[7]) / d + eps);
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int c = tid + 256 * i;
    if (c < nv) {
      const float4 g = gm[i], b = bt[i];
      float4 o;
      o.x = (v[i].x - mean) * rstd * g.x + b.x;
      o.y = (v[i].y - mean) * rstd * g.y + b.y;
      o.z = (v[i].z - mean) * rstd * g.z + b.z;
      o.w = (v[i].w - mean) * rstd * g.w + b.w;
      if (out_bf16) {
        uint2 p;
        p.x = (uint32_t)f2bf(o.x) | ((uint32_t)f2bf(o.y) << 16);
        p.y = (uint32_t)f2bf(o.z) | ((uint32_t)f2bf(o.w) << 16);
        ((uint2*)(out_bf16 + (long)row * d))[c] = p;
      }
      if (out_f32) ((float4*)(out_f32 + (long)row * d))[c] = o;
    }
  }
}

void launch_add_layernorm(float* x, const float* slabs, int n_slabs, long slab_stride,
                          const float* add_bias, const float* gamma, const float* beta,
                          bf16_t* out_bf16, float* out_f32, int m, int d, float eps,
                          int row_mul, int row_off, hipStream_t stream) {
  if (m <= 1024) {     // d <= 2048 = 256 lanes x 2 float4
    hipLaunchKernelGGL(add_layernorm_row_kernel, dim3(m), dim3(256), 0, stream, x, slabs, n_slabs,
                       slab_stride, add_bias, gamma, beta, out_bf16, out_f32, m, d, eps, row_mul,
                       row_off);
    return;
  }
  hipLaunchKernelGGL(add_layernorm_kernel, dim3((m + 3) / 4), dim3(256), 0, stream, x, slabs,
                     n_slabs, slab_stride, add_bias, gamma, beta, out_bf16, out_f32, m, d, eps,
                     row_mul, row_off);
}

// mel [B][frames][n_mels] fp32 -> [B][3002][n_mels] bf16: row 0 and row 3001 zero (the conv's
// ZeroPadding1D, current_blocks.py:297-307), rows beyond `frames` zero (pad_or_trim with
// pad_mel_value = 0, models/stt/base_stt.py:227-236).
__global__ void pack_mel_kernel(const float* __restrict__ mel, int frames, int n_mels,
                                bf16_t* __restrict__ out) {
  const int b = blockIdx.y;
  const long per_b = 3002L * n_mels;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < per_b; i += (long)gridDim.x * blockDim.x) {
    const int row = (int)(i / n_mels), c = (int)(i - (long)row * n_mels);
    float v = 0.f;
    if (row >= 1 && row <= frames && row <= 3000) v = mel[((long)b * frames + (row - 1)) * n_mels + c];
    out[b * per_b + i] = f2bf(v);
  }
}
void launch_pack_mel(const float* mel, int batch, int frames, int n_mels, bf16_t* out,
                     hipStream_t stream) {
  hipLaunchKernelGGL(pack_mel_kernel, dim3(240, batch), dim3(256), 0, stream, mel, frames, n_mels, out);
}

// token + learned position embedding, no scaling (text_transformer_arch.py:173-193).
__global__ void embed_kernel(const int* __restrict__ tokens, const int* __restrict__ positions,
                             const bf16_t* __restrict__ emb, const float* __restrict__ pos_emb,
                             float* __restrict__ x, int rows, int d, int n_vocab, int n_ctx) {
  const int row = blockIdx.x;
  if (row >= rows) return;
  int tok = tokens[row]; tok = tok < 0 ? 0 : (tok >= n_vocab ? n_vocab - 1 : tok);
  int p = positions[row]; p = p < 0 ? 0 : (p >= n_ctx ? n_ctx - 1 : p);
  for (int c = threadIdx.x; c < d; c += blockDim.x)
    x[(long)row * d + c] = bf2f(emb[(long)tok * d + c]) + pos_emb[(long)p * d + c];
}
void launch_embed(const int* tokens, const int* positions, const bf16_t* emb, const float* pos_emb,
                  float* x, int rows, int d, int n_vocab, int n_ctx, hipStream_t stream) {
  hipLaunchKernelGGL(embed_kernel, dim3(rows), dim3(256), 0, stream, tokens, positions, emb,
                     pos_emb, x, rows, d, n_vocab, n_ctx);
}
