// bf16 MFMA GEMMs of the Whisper engine (gfx950): C[m][n] = sum_k A[m][k] * W[n][k].
//
// Both kernels feed v_mfma_f32_16x16x32_bf16 with the weight fragment as the A operand and the
// activation fragment as the B operand, so that a lane ends up with four consecutive output
// features n of one row m: bias loads and the stores are then 8/16-byte vectors.
//
// Replaces the Dense / Conv1D calls of the reference model (keras.layers.Dense in
// architectures/layers/residual_multi_head_attention.py:108-126 and
// architectures/transformers/transformer_arch.py:259-266, Conv1D of the encoder stem
// architectures/transformers/whisper_arch.py:121-136; a stride-s 'valid' Conv1D over zero-padded
// rows is a GEMM whose A rows overlap: row stride = s * channels, K = 3 * channels).
#include "kernels.h"
#include <cstdlib>
#include <cstdlib>

typedef __attribute__((ext_vector_type(8))) __bf16 bf8_t;
typedef __attribute__((ext_vector_type(4))) float f4_t;

__device__ __forceinline__ bf16_t f2bf(float f) {
  uint32_t u = __float_as_uint(f);
  u += 0x7FFFu + ((u >> 16) & 1u);
  return (bf16_t)(u >> 16);
}
__device__ __forceinline__ float bf2f(bf16_t h) { return __uint_as_float(((uint32_t)h) << 16); }

__device__ __forceinline__ float gelu_f(float x, int tanh_form) {
  if (tanh_form) {
    // 0.5 x (1 + tanh(u)) = x / (1 + exp(-2u)),  u = sqrt(2/pi) (x + 0.044715 x^3)
    const float u = 0.7978845608028654f * (x + 0.044715f * x * x * x);
    return x / (1.0f + __expf(-2.0f * u));
  }
  return 0.5f * x * (1.0f + erff(x * 0.7071067811865476f));
}

typedef __attribute__((address_space(3))) void gb_lds_void;
typedef const __attribute__((address_space(1))) void gb_glb_void;
#ifndef GB_BM
#define GB_BM 256
#endif
#ifndef GB_GM
#define GB_GM 8
#endif
#define GB_BK 64

// two floats -> packed bf16 (round to nearest even) in one v_cvt_pk_bf16_f32
typedef __bf16 gb_bf2 __attribute__((ext_vector_type(2)));
typedef float gb_f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t gb_pack2(float a, float b) {
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(gb_f2{a, b}, gb_bf2));
}
// GELU of four values.  TANH: the tanh form on packed pairs (v_pk_mul / v_pk_fma / v_pk_add, one
// exp2 and one reciprocal per value), straight-line code so that the values' dependence chains
// interleave; otherwise the exact erf form.
template <bool TANH>
__device__ __forceinline__ f4_t gelu4(f4_t v) {
  if (TANH) {
    const float c1 = -2.0f * 0.7978845608028654f * 1.4426950408889634f, c2 = c1 * 0.044715f;
    const f4_t t = (v * v * c2 + c1) * v;
    f4_t den;
#pragma unroll
    for (int r = 0; r < 4; ++r) den[r] = __builtin_amdgcn_exp2f(t[r]);
    den += 1.0f;
#pragma unroll
    for (int r = 0; r < 4; ++r) den[r] = __builtin_amdgcn_rcpf(den[r]);
    return v * den;
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) v[r] = 0.5f * v[r] * (1.0f + erff(v[r] * 0.7071067811865476f));
  return v;
}

// Epilogue of the tiled kernels: the lane holds C[m = mbase + 16 tm + fr][n = nbase + 16 tn + 4 fg + r].
// Everything is done on 4-wide vectors (16-byte bias / position / residual accesses, 8-byte bf16
// stores); a ragged last column group falls back to scalars.
template <int EPI, int TM, bool TANH>
__device__ __forceinline__ void big_epilogue_t(const GemmArgs& g, f4_t (&acc)[4][TM], int mbase, int nbase,
                                               int fr, int fg) {
  const int M = g.m, N = g.n;
  const long out_base = (long)blockIdx.z * g.out_batch_stride + (long)blockIdx.y * g.split_stride;
#pragma unroll
  for (int tn = 0; tn < 4; ++tn) {
    const int n = nbase + tn * 16 + 4 * fg;
    if (n >= N) continue;
    const int nv = N - n < 4 ? N - n : 4;
    f4_t bias = f4_t{0.f, 0.f, 0.f, 0.f};
    if (g.bias) {
      if (nv == 4) bias = *(const f4_t*)(g.bias + n);
      else for (int r = 0; r < nv; ++r) bias[r] = g.bias[n + r];
    }
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
      const int m = mbase + tm * 16 + fr;
      if (m >= M) continue;
      f4_t v = acc[tn][tm] + bias;
      if (EPI == EPI_GELU_BF16 || EPI == EPI_GELU_POS_F32) v = gelu4<TANH>(v);
      if (EPI == EPI_KV_HEADS) {               // N % 128 == 0: a 4-group never straddles a head
        const int dm = N >> 1, sel = n >= dm, nn = sel ? n - dm : n;
        const int bi = m / g.kv_t, t = m - bi * g.kv_t;
        bf16_t* out = (bf16_t*)(sel ? g.out2 : g.out);
        uint2 p;
        p.x = gb_pack2(v[0], v[1]);
        p.y = gb_pack2(v[2], v[3]);
        *(uint2*)(out + (((long)bi * (dm >> 6) + (nn >> 6)) * g.kv_t + t) * 64 + (nn & 63)) = p;
        continue;
      }
      if (EPI == EPI_QKV_ENC) {                // a 4-group never straddles q | k | v or a head
        const int dm = N / 3;
        if (n < dm) v *= g.q_scale;
        if (n >= 2 * dm) {
          bf16_t* vt = (bf16_t*)g.out2 + (long)(n - 2 * dm) * g.vt_pitch + m;
#pragma unroll
          for (int r = 0; r < 4; ++r) vt[(long)r * g.vt_pitch] = f2bf(v[r]);
          continue;
        }
      }
      const long o = out_base + (long)m * g.ldc + n;
      const bool vec = nv == 4 && ((o & 3) == 0);
      if (EPI == EPI_BF16 || EPI == EPI_GELU_BF16 || EPI == EPI_QKV_ENC) {
        bf16_t* out = (bf16_t*)g.out;
        if (vec) {
          uint2 p;
          p.x = gb_pack2(v[0], v[1]);
          p.y = gb_pack2(v[2], v[3]);
          *(uint2*)(out + o) = p;
        } else {
          for (int r = 0; r < nv; ++r) out[o + r] = f2bf(v[r]);
        }
      } else {
        float* out = (float*)g.out;
        if (EPI == EPI_GELU_POS_F32) {
          const long po = (long)m * g.pos_ld + n;
          if (nv == 4 && ((po & 3) == 0)) v += *(const f4_t*)(g.pos + po);
          else for (int r = 0; r < nv; ++r) v[r] += g.pos[po + r];
        }
        if (vec) {
          if (EPI == EPI_RESID_F32) v += *(const f4_t*)(out + o);
          *(f4_t*)(out + o) = v;
        } else if (EPI == EPI_RESID_F32) {
          for (int r = 0; r < nv; ++r) out[o + r] += v[r];
        } else {
          for (int r = 0; r < nv; ++r) out[o + r] = v[r];
        }
      }
    }
  }
}

// the GELU form is chosen once per kernel, outside the loops over the lane's outputs
template <int EPI, int TM>
__device__ __forceinline__ void big_epilogue(const GemmArgs& g, f4_t (&acc)[4][TM], int mbase, int nbase,
                                             int fr, int fg) {
  if ((EPI == EPI_GELU_BF16 || EPI == EPI_GELU_POS_F32) && !g.gelu_tanh)
    big_epilogue_t<EPI, TM, false>(g, acc, mbase, nbase, fr, fg);
  else
    big_epilogue_t<EPI, TM, true>(g, acc, mbase, nbase, fr, fg);
}

// V tile of the encoder's q | k | v projection (EPI_QKV_ENC): the wave's 64 columns are one head
// of V.  Its 128 x 64 tile is parked transposed in the wave's private 16 KB of LDS, [column][128
// rows] in 256-byte lines whose 8-byte pieces are XOR-swizzled by the lane group, and leaves as
// 8-byte pieces of whole 256-byte runs of 128 positions of one feature (the image is indexed by the
// GEMM's row, so a run starts on a 256-byte boundary).  rows_valid (a multiple of 4) cuts the
// tile at the end of the matrix.
__device__ __forceinline__ void qkv_v_tile_store(const GemmArgs& g, f4_t (&acc)[4][8], int mbase, int nbase,
                                                 int lane, unsigned char* stage, int rows_valid) {
  const int fr = lane & 15, fg = lane >> 4;
#pragma unroll
  for (int tn = 0; tn < 4; ++tn) {
    const int n = nbase + tn * 16 + 4 * fg;
    const f4_t bias = g.bias ? *(const f4_t*)(g.bias + n) : f4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int tm = 0; tm < 8; ++tm) {
      const f4_t v = acc[tn][tm] + bias;
      const int row = tm * 16 + fr;
      const unsigned in_line = ((((unsigned)row >> 2) ^ ((unsigned)fg << 2)) << 3) + (row & 3) * 2;
      const uint32_t p01 = gb_pack2(v[0], v[1]), p23 = gb_pack2(v[2], v[3]);
      const int col = tn * 16 + 4 * fg;
      *(bf16_t*)(stage + (col + 0) * 256 + in_line) = (bf16_t)(p01 & 0xFFFFu);
      *(bf16_t*)(stage + (col + 1) * 256 + in_line) = (bf16_t)(p01 >> 16);
      *(bf16_t*)(stage + (col + 2) * 256 + in_line) = (bf16_t)(p23 & 0xFFFFu);
      *(bf16_t*)(stage + (col + 3) * 256 + in_line) = (bf16_t)(p23 >> 16);
    }
  }
  __builtin_amdgcn_s_waitcnt(0xC07F);      // lgkmcnt(0); the region is private to the wave
  const int pc = lane & 31;
  if (4 * pc >= rows_valid) return;
  bf16_t* vt = (bf16_t*)g.out2 + (long)(nbase - 2 * (g.n / 3)) * g.vt_pitch + mbase + 4 * pc;
#pragma unroll
  for (int it = 0; it < 32; ++it) {
    const int col = it * 2 + (lane >> 5);
    const uint2 q = *(const uint2*)(stage + col * 256 + ((pc ^ (((col >> 2) & 3) << 2)) << 3));
    *(uint2*)(vt + (long)col * g.vt_pitch) = q;
  }
}

// bf16 epilogue of the wide kernel through LDS: a lane's accumulators are 4 consecutive features
// of one row, so direct stores are 8-byte pieces, 16 rows per instruction.  Each wave instead
// parks its 128 x 64 bf16 tile in its own 16 KB of the (now idle) operand stages (16-byte pieces
// XOR-swizzled by row) and writes it back out as whole 128-byte rows, 8 rows per instruction.  Only full
// tiles take this path (the caller checks); `stage` is this wave's private region.
template <int EPI, bool TANH>
__device__ __forceinline__ void wide_epilogue_bf16_t(const GemmArgs& g, f4_t (&acc)[4][8], int mbase,
                                                     int nbase, int lane, unsigned char* stage) {
  const int fr = lane & 15, fg = lane >> 4;
  if (EPI == EPI_QKV_ENC && nbase >= 2 * (g.n / 3)) {
    qkv_v_tile_store(g, acc, mbase, nbase, lane, stage, 128);
    return;
  }
  const float scale = (EPI == EPI_QKV_ENC && nbase < g.n / 3) ? g.q_scale : 1.0f;
#pragma unroll
  for (int tn = 0; tn < 4; ++tn) {
    const int n = nbase + tn * 16 + 4 * fg;
    const f4_t bias = g.bias ? *(const f4_t*)(g.bias + n) : f4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int tm = 0; tm < 8; ++tm) {
      f4_t v = acc[tn][tm] + bias;
      if (EPI == EPI_QKV_ENC) v *= scale;
      if (EPI == EPI_GELU_BF16) v = gelu4<TANH>(v);
      uint2 p;
      p.x = gb_pack2(v[0], v[1]);
      p.y = gb_pack2(v[2], v[3]);
      const int row = tm * 16 + fr, pc = tn * 2 + (fg >> 1);
      *(uint2*)(stage + row * 128 + ((pc ^ (row & 7)) << 4) + (fg & 1) * 8) = p;
    }
  }
  __builtin_amdgcn_s_waitcnt(0xC07F);        // lgkmcnt(0); the region is private to the wave
  const int piece = lane & 7, r0 = lane >> 3;
  if (EPI == EPI_KV_HEADS) {                 // the wave's 64 columns are exactly one head of K or of V
    const int dm = g.n >> 1, sel = nbase >= dm, nn = sel ? nbase - dm : nbase;
    bf16_t* out = (bf16_t*)(sel ? g.out2 : g.out);
#pragma unroll
    for (int it = 0; it < 16; ++it) {
      const int row = it * 8 + r0, m = mbase + row;
      const int bi = m / g.kv_t, t = m - bi * g.kv_t;
      const uint4 q = *(const uint4*)(stage + row * 128 + ((piece ^ (row & 7)) << 4));
      *(uint4*)(out + (((long)bi * (dm >> 6) + (nn >> 6)) * g.kv_t + t) * 64 + piece * 8) = q;
    }
    return;
  }
  bf16_t* out = (bf16_t*)g.out + (long)blockIdx.z * g.out_batch_stride;
#pragma unroll
  for (int it = 0; it < 16; ++it) {
    const int row = it * 8 + r0;
    const uint4 q = *(const uint4*)(stage + row * 128 + ((piece ^ (row & 7)) << 4));
    *(uint4*)(out + (long)(mbase + row) * g.ldc + nbase + piece * 8) = q;
  }
}

template <int EPI>
__device__ __forceinline__ void wide_epilogue_bf16(const GemmArgs& g, f4_t (&acc)[4][8], int mbase,
                                                   int nbase, int lane, unsigned char* stage) {
  if (EPI == EPI_GELU_BF16 && !g.gelu_tanh) wide_epilogue_bf16_t<EPI, false>(g, acc, mbase, nbase, lane, stage);
  else wide_epilogue_bf16_t<EPI, true>(g, acc, mbase, nbase, lane, stage);
}

// fp32 residual epilogue of the wide kernel, same idea: the wave's 128 x 64 tile goes through its
// 16 KB of LDS in two halves of 32 columns (128 bytes per row, pieces XOR-swizzled by row), and
// the read-add-write of the residual stream is done on whole 128-byte rows.
__device__ __forceinline__ void wide_epilogue_resid(const GemmArgs& g, f4_t (&acc)[4][8], int mbase,
                                                    int nbase, int lane, unsigned char* stage) {
  const int fr = lane & 15, fg = lane >> 4;
  float* out = (float*)g.out + (long)blockIdx.z * g.out_batch_stride;
  const int piece = lane & 7, r0 = lane >> 3;
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    // The residual values of the whole half are requested before anything else: a load issued
    // after a store to the same array may not be moved above it by the compiler, and dependent
    // round trips to HBM, during which the block's matrix cores idle, cost about 11 us per tile
    // (a K = 1280 main loop takes 33).
    f4_t res[16];
#pragma unroll
    for (int it = 0; it < 16; ++it)
      res[it] = *(const f4_t*)(out + (long)(mbase + it * 8 + r0) * g.ldc + nbase + half * 32 + piece * 4);
#pragma unroll
    for (int t2 = 0; t2 < 2; ++t2) {
      const int tn = half * 2 + t2;
      const int n = nbase + tn * 16 + 4 * fg;
      const f4_t bias = g.bias ? *(const f4_t*)(g.bias + n) : f4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int tm = 0; tm < 8; ++tm) {
        const int row = tm * 16 + fr, pc = t2 * 4 + fg;
        *(f4_t*)(stage + row * 128 + ((pc ^ (row & 7)) << 4)) = acc[tn][tm] + bias;
      }
    }
    __builtin_amdgcn_s_waitcnt(0xC07F);      // lgkmcnt(0); the region is private to the wave
#pragma unroll
    for (int it = 0; it < 16; ++it) {
      const int row = it * 8 + r0;
      const f4_t v = *(const f4_t*)(stage + row * 128 + ((piece ^ (row & 7)) << 4));
      float* dst = out + (long)(mbase + row) * g.ldc + nbase + half * 32 + piece * 4;
      *(f4_t*)dst = res[it] + v;
    }
    __builtin_amdgcn_s_waitcnt(0xC07F);      // reads done before the second half overwrites
  }
}

// XCD-aware tile order shared by the tiled kernels (see gemm_big_kernel).
__device__ __forceinline__ void big_tile_index(int n_tiles, int tiles_n, int& tm_idx, int& tn_idx) {
  const int b = blockIdx.x, xcd = b & 7, q = n_tiles >> 3, r = n_tiles & 7;
  const int tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
  const int tiles_m = n_tiles / tiles_n;
  const int per_group = GB_GM * tiles_n;
  const int grp = tile / per_group, in = tile - grp * per_group;
  const int gm = tiles_m - grp * GB_GM < GB_GM ? tiles_m - grp * GB_GM : GB_GM;
  tn_idx = in / gm;
  tm_idx = grp * GB_GM + (in - tn_idx * gm);
}

// ---------------------------------------------------------------------------------------------
// Wide variant for N % 256 == 0: 256 x 256 x 64 block tile, 8 waves (2 x 4) with the same
// 128 x 64 wave tile.  The tiled kernels are bound by the L2-to-CU stream of operand tiles
// (waves sit in s_waitcnt for half their cycles); a 256 x 256 tile moves a third fewer bytes
// per flop than 256 x 128.  One block per CU, so the overlap
// comes from inside the block: two 64 KB LDS stages, the DMA of K tile t + 1 is issued right
// after the barrier that publishes tile t and is waited for at the top of the next iteration.
// ---------------------------------------------------------------------------------------------
template <int EPI>
__global__ __launch_bounds__(512) void gemm_wide_kernel(const GemmArgs g, int tiles_n, int n_tiles) {
  constexpr int kStage = (256 + 256) * GB_BK * 2, kBOff = 256 * GB_BK * 2;
  __shared__ __attribute__((aligned(1024))) unsigned char smem[2 * kStage];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  int tm_idx, tn_idx;
  big_tile_index(n_tiles, tiles_n, tm_idx, tn_idx);
  const int m0 = tm_idx * 256, n0 = tn_idx * 256;
  const bf16_t* __restrict__ A = g.a + (long)blockIdx.z * g.a_batch_stride;
  const bf16_t* __restrict__ W = g.w;
  const int M = g.m, N = g.n, K = g.k;
  const int nk = K / GB_BK;

  // DMA instruction j = wave + 8 i covers rows 8j..8j+7 of an operand (32 instructions each)
  const bf16_t* a_src[4];
  const bf16_t* w_src[4];
  const int piece = (lane & 7) ^ (lane >> 3);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = 8 * (wave + 8 * i) + (lane >> 3);
    int gm = m0 + row; gm = gm < M ? gm : M - 1;
    int gn = n0 + row; gn = gn < N ? gn : N - 1;
    a_src[i] = A + (long)gm * g.lda + piece * 8;
    w_src[i] = W + (long)gn * K + piece * 8;
  }
  auto issue = [&](int kt) {
    unsigned char* dst = smem + (kt & 1) * kStage;
#pragma unroll
    for (int i = 0; i < 4; ++i)
      __builtin_amdgcn_global_load_lds((gb_glb_void*)(a_src[i] + kt * GB_BK),
                                       (gb_lds_void*)(dst + (wave + 8 * i) * 1024), 16, 0, 0);
#pragma unroll
    for (int i = 0; i < 4; ++i)
      __builtin_amdgcn_global_load_lds((gb_glb_void*)(w_src[i] + kt * GB_BK),
                                       (gb_lds_void*)(dst + kBOff + (wave + 8 * i) * 1024), 16, 0, 0);
  };

  f4_t acc[4][8];   // [tn][tm]
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[i][j] = f4_t{0.f, 0.f, 0.f, 0.f};

  const int fr = lane & 15, fg = lane >> 4;
  issue(0);
  for (int kt = 0; kt < nk; ++kt) {
    __builtin_amdgcn_s_waitcnt(0x0F70);     // vmcnt(0): this wave's pieces of tile kt have landed
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();           // ... everyone's have, and everyone is done with tile kt - 1
    asm volatile("" ::: "memory");
    if (kt + 1 < nk) issue(kt + 1);
    const unsigned char* cur = smem + (kt & 1) * kStage;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf8_t wf[4], af[8];
      const int ch = ks * 4 + fg;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int rw = wn * 64 + t * 16 + fr;
        wf[t] = __builtin_bit_cast(bf8_t, *(const uint4*)(cur + kBOff + rw * 128 + ((ch ^ (rw & 7)) << 4)));
      }
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        const int rm = wm * 128 + t * 16 + fr;
        af[t] = __builtin_bit_cast(bf8_t, *(const uint4*)(cur + rm * 128 + ((ch ^ (rm & 7)) << 4)));
      }
#pragma unroll
      for (int tn = 0; tn < 4; ++tn)
#pragma unroll
        for (int tm = 0; tm < 8; ++tm)
          acc[tn][tm] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[tn], af[tm], acc[tn][tm], 0, 0, 0);
    }
    __builtin_amdgcn_s_waitcnt(0xC07F);     // lgkmcnt(0): this wave's reads of the stage are done
  }
  if ((EPI == EPI_BF16 || EPI == EPI_GELU_BF16 || EPI == EPI_KV_HEADS || EPI == EPI_QKV_ENC) && m0 + 256 <= M &&
      (g.ldc & 7) == 0 && ((blockIdx.z * g.out_batch_stride) & 7) == 0) {
    __syncthreads();                         // every wave is done with the operand stages
    wide_epilogue_bf16<EPI>(g, acc, m0 + wm * 128, n0 + wn * 64, lane, smem + wave * (128 * 128));
  } else if (EPI == EPI_RESID_F32 && m0 + 256 <= M && (g.ldc & 3) == 0 &&
             ((blockIdx.z * g.out_batch_stride) & 3) == 0) {
    __syncthreads();
    wide_epilogue_resid(g, acc, m0 + wm * 128, n0 + wn * 64, lane, smem + wave * (128 * 128));
  } else {
    big_epilogue<EPI, 8>(g, acc, m0 + wm * 128, n0 + wn * 64, fr, fg);
  }
}

// ---------------------------------------------------------------------------------------------
// Large-M kernel: GB_BM x 128 x 64 block tile, 4 waves (2 x 2), each wave (GB_BM / 2) x 64 =
// (GB_BM / 32) x 4 MFMA tiles (a 128-row wave tile reads 12 KB of fragments per 32 MFMAs where a
// 64-row one reads 8 KB per 16: LDS bandwidth, not the matrix cores, bounds the smaller tile).
// One LDS stage per block, filled by LDS-DMA (16 B per lane, no staging registers), two
// barriers per K step; two or three blocks share a CU, and it is the other blocks' MFMAs that
// cover a block's load latency.
// LDS rows are 128 B (64 k values); the 16-byte piece c of row r sits at slot c ^ (r & 7), so the
// sixteen rows a fragment read touches spread over all banks.  An LDS-DMA instruction writes
// 1 KB = 8 whole rows in lane order (lane -> row lane / 8, slot lane % 8), so the swizzle is
// applied on the source side: the lane fetches piece (lane % 8) ^ (lane / 8) of its row.
// Blocks are renumbered so that the blocks an XCD runs (every 8th) walk neighbouring tiles and
// share activation panels in that XCD's L2.
// ---------------------------------------------------------------------------------------------
#ifndef SKINNY_NT2_MIN
#define SKINNY_NT2_MIN 0
#endif
#define GB_BN 128
#define GB_TM (GB_BM / 32)       // MFMA row tiles per wave

template <int EPI>
__global__ __launch_bounds__(256, 2) void gemm_big_kernel(const GemmArgs g, int tiles_n, int n_tiles) {
  // operand stage (48 KB); the q | k | v projection parks its V tiles here afterwards (16 KB per wave)
  constexpr int kSmem = (EPI == EPI_QKV_ENC && GB_TM == 8 && 4 * 16384 > (GB_BM + GB_BN) * GB_BK * 2)
                            ? 4 * 16384 : (GB_BM + GB_BN) * GB_BK * 2;
  __shared__ __attribute__((aligned(1024))) unsigned char smem[kSmem];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  // XCD-aware renumbering (bijective for any tile count): hardware block b runs on XCD b % 8 and
  // takes tile b / 8 of that XCD's contiguous share.  Tiles are walked in groups of GB_GM row
  // panels, row panel fastest: the ~64 tiles an XCD has in flight then form a GB_GM x 8 patch that
  // shares both its activation panels and its weight tiles in that XCD's L2 (walking one row
  // panel at a time shares the activations only and streams the whole weight matrix from the
  // memory side for every panel).
  int tm_idx, tn_idx;
  big_tile_index(n_tiles, tiles_n, tm_idx, tn_idx);
  const int m0 = tm_idx * GB_BM, n0 = tn_idx * GB_BN;
  const bf16_t* __restrict__ A = g.a + (long)blockIdx.z * g.a_batch_stride;
  const bf16_t* __restrict__ W = g.w;
  const int M = g.m, N = g.n, K = g.k;
  const int kper = g.k_split > 1 ? K / g.k_split : K;      // this block's K slice (split-K over blockIdx.y)
  const int kbeg = blockIdx.y * kper;
  const int nk = kper / GB_BK;
  constexpr int kBOff = GB_BM * GB_BK * 2;

  // this lane's source rows of A and of W (DMA instruction j = wave + 4 i covers rows 8j..8j+7)
  constexpr int kADma = GB_BM / 32, kWDma = GB_BN / 32;
  const bf16_t* a_src[kADma];
  const bf16_t* w_src[kWDma];
  const int piece = (lane & 7) ^ (lane >> 3);
#pragma unroll
  for (int i = 0; i < kADma; ++i) {
    const int row = 8 * (wave + 4 * i) + (lane >> 3);
    int gm = m0 + row; gm = gm < M ? gm : M - 1;
    a_src[i] = A + (long)gm * g.lda + kbeg + piece * 8;
  }
#pragma unroll
  for (int i = 0; i < kWDma; ++i) {
    const int row = 8 * (wave + 4 * i) + (lane >> 3);
    int gn = n0 + row; gn = gn < N ? gn : N - 1;
    w_src[i] = W + (long)gn * K + kbeg + piece * 8;
  }

  f4_t acc[4][GB_TM];   // [tn][tm]
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < GB_TM; ++j) acc[i][j] = f4_t{0.f, 0.f, 0.f, 0.f};

  const int fr = lane & 15, fg = lane >> 4;
  for (int kt = 0; kt < nk; ++kt) {
#pragma unroll
    for (int i = 0; i < kADma; ++i)
      __builtin_amdgcn_global_load_lds((gb_glb_void*)(a_src[i] + kt * GB_BK),
                                       (gb_lds_void*)(smem + (wave + 4 * i) * 1024), 16, 0, 0);
#pragma unroll
    for (int i = 0; i < kWDma; ++i)
      __builtin_amdgcn_global_load_lds((gb_glb_void*)(w_src[i] + kt * GB_BK),
                                       (gb_lds_void*)(smem + kBOff + (wave + 4 * i) * 1024), 16, 0, 0);
    __builtin_amdgcn_s_waitcnt(0x0F70);     // vmcnt(0): this wave's pieces have landed
    __syncthreads();
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf8_t wf[4], af[GB_TM];
      const int ch = ks * 4 + fg;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int rw = wn * 64 + t * 16 + fr;
        wf[t] = __builtin_bit_cast(bf8_t, *(const uint4*)(smem + kBOff + rw * 128 + ((ch ^ (rw & 7)) << 4)));
      }
#pragma unroll
      for (int t = 0; t < GB_TM; ++t) {
        const int rm = wm * (GB_BM / 2) + t * 16 + fr;
        af[t] = __builtin_bit_cast(bf8_t, *(const uint4*)(smem + rm * 128 + ((ch ^ (rm & 7)) << 4)));
      }
#pragma unroll
      for (int tn = 0; tn < 4; ++tn)
#pragma unroll
        for (int tm = 0; tm < GB_TM; ++tm)
          acc[tn][tm] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[tn], af[tm], acc[tn][tm], 0, 0, 0);
    }
    __syncthreads();                        // everyone is done reading before the next fill
  }

  if constexpr (EPI == EPI_QKV_ENC && GB_TM == 8) {
    const int mbase = m0 + wm * (GB_BM / 2), nbase = n0 + wn * 64;
    if (nbase >= 2 * (N / 3) && nbase + 64 <= N) {          // the last barrier of the loop freed the stage
      const int left = M - mbase;
      if (left > 0) qkv_v_tile_store(g, acc, mbase, nbase, lane, smem + wave * 16384, left < 128 ? left : 128);
      return;
    }
  }
  big_epilogue<EPI, GB_TM>(g, acc, m0 + wm * (GB_BM / 2), n0 + wn * 64, fr, fg);
}

void launch_gemm_big(const GemmArgs& g, int epilogue, hipStream_t stream) {
  static const bool no_wide = getenv("STT_MI355X_NO_WIDE") != nullptr;
  const int ksplit = g.k_split > 1 ? g.k_split : 1;
  if (!no_wide && ksplit == 1 && g.n % 256 == 0 && g.m >= 2048) {
    const int tiles_n = g.n / 256, tiles_m = (g.m + 255) / 256;
    const int n_tiles = tiles_n * tiles_m;
    dim3 grid(n_tiles, 1, g.batch > 0 ? g.batch : 1);
    switch (epilogue) {
      case EPI_BF16: hipLaunchKernelGGL(gemm_wide_kernel<EPI_BF16>, grid, dim3(512), 0, stream, g, tiles_n, n_tiles); break;
      case EPI_GELU_BF16: hipLaunchKernelGGL(gemm_wide_kernel<EPI_GELU_BF16>, grid, dim3(512), 0, stream, g, tiles_n, n_tiles); break;
      case EPI_RESID_F32: hipLaunchKernelGGL(gemm_wide_kernel<EPI_RESID_F32>, grid, dim3(512), 0, stream, g, tiles_n, n_tiles); break;
      case EPI_GELU_POS_F32: hipLaunchKernelGGL(gemm_wide_kernel<EPI_GELU_POS_F32>, grid, dim3(512), 0, stream, g, tiles_n, n_tiles); break;
      case EPI_KV_HEADS: hipLaunchKernelGGL(gemm_wide_kernel<EPI_KV_HEADS>, grid, dim3(512), 0, stream, g, tiles_n, n_tiles); break;
      case EPI_QKV_ENC: hipLaunchKernelGGL(gemm_wide_kernel<EPI_QKV_ENC>, grid, dim3(512), 0, stream, g, tiles_n, n_tiles); break;
      default: hipLaunchKernelGGL(gemm_wide_kernel<EPI_F32>, grid, dim3(512), 0, stream, g, tiles_n, n_tiles); break;
    }
    return;
  }
  const int tiles_n = (g.n + GB_BN - 1) / GB_BN, tiles_m = (g.m + GB_BM - 1) / GB_BM;
  const int n_tiles = tiles_n * tiles_m;
  dim3 grid(n_tiles, ksplit, g.batch > 0 ? g.batch : 1);
  switch (epilogue) {
    case EPI_BF16: hipLaunchKernelGGL(gemm_big_kernel<EPI_BF16>, grid, dim3(256), 0, stream, g, tiles_n, n_tiles); break;
    case EPI_GELU_BF16: hipLaunchKernelGGL(gemm_big_kernel<EPI_GELU_BF16>, grid, dim3(256), 0, stream, g, tiles_n, n_tiles); break;
    case EPI_RESID_F32: hipLaunchKernelGGL(gemm_big_kernel<EPI_RESID_F32>, grid, dim3(256), 0, stream, g, tiles_n, n_tiles); break;
    case EPI_GELU_POS_F32: hipLaunchKernelGGL(gemm_big_kernel<EPI_GELU_POS_F32>, grid, dim3(256), 0, stream, g, tiles_n, n_tiles); break;
    case EPI_KV_HEADS: hipLaunchKernelGGL(gemm_big_kernel<EPI_KV_HEADS>, grid, dim3(256), 0, stream, g, tiles_n, n_tiles); break;
    case EPI_QKV_ENC: hipLaunchKernelGGL(gemm_big_kernel<EPI_QKV_ENC>, grid, dim3(256), 0, stream, g, tiles_n, n_tiles); break;
    default: hipLaunchKernelGGL(gemm_big_kernel<EPI_F32>, grid, dim3(256), 0, stream, g, tiles_n, n_tiles); break;
  }
}

// ---------------------------------------------------------------------------------------------
// Small-M kernel (decode): a block owns 16 output features; its 4 waves take interleaved
// 64-wide K steps, so that per iteration the block reads 512 contiguous bytes of each of the 16
// weight rows straight from HBM into registers (lane group g reads the 16-byte pieces 2g and
// 2g+1 of its wave's 128 bytes; the activation fragment uses the same k order).  Activations
// (M x K, a few hundred KB at most) come from L2.  The four partial tiles are summed through
// LDS and the epilogue is applied once.  blockIdx.y selects a K slice (split-K across blocks,
// raw partial sums to slabs).
// ---------------------------------------------------------------------------------------------
template <int MT, int NT, int UW, int EPI>   // MT = ceil(M / 16); a block owns 16 * NT output features
__global__ __launch_bounds__(256) void gemm_skinny_kernel(const GemmArgs g, int k_per_split,
                                                          float* __restrict__ slabs) {
  __shared__ f4_t part[3][NT * MT][64];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, fg = lane >> 4;
  const int n0 = blockIdx.x * (16 * NT);
  const int M = g.m, N = g.n, K = g.k;
  const int kbeg = blockIdx.y * k_per_split;
  const bf16_t* __restrict__ wrow[NT];
#pragma unroll
  for (int u = 0; u < NT; ++u) {
    int wn = n0 + 16 * u + fr; wn = wn < N ? wn : N - 1;
    wrow[u] = g.w + (long)wn * K + kbeg + 16 * fg;
  }
  const bf16_t* __restrict__ arow[MT];
#pragma unroll
  for (int t = 0; t < MT; ++t) {
    int m = t * 16 + fr; m = m < M ? m : M - 1;
    arow[t] = g.a + (long)m * g.lda + kbeg + 16 * fg;
  }
  f4_t acc[NT][MT];
#pragma unroll
  for (int u = 0; u < NT; ++u)
#pragma unroll
    for (int t = 0; t < MT; ++t) acc[u][t] = f4_t{0.f, 0.f, 0.f, 0.f};

  typedef __attribute__((ext_vector_type(4))) unsigned int u4_t;
  // This wave's steps are s = wave, wave + 4, ...  The layer is a latency-bound stream: all the
  // loads of up to UW steps are issued before the first MFMA, so a K slice of 4 UW steps costs
  // one memory round trip instead of one per step.  Steps past the end re-read the last step
  // (no branches around the loads) and are skipped at the MFMAs.
  const int nsteps = k_per_split / 64;
  const int nmy = (nsteps - wave + 3) >> 2;            // wave-uniform
  for (int c = 0; c < nmy; c += UW) {
    bf8_t wf[UW][NT][2], af[UW][MT][2];
#pragma unroll
    for (int i = 0; i < UW; ++i) {
      const int s = wave + 4 * (c + i < nmy ? c + i : nmy - 1);
#pragma unroll
      for (int u = 0; u < NT; ++u) {
        wf[i][u][0] = __builtin_bit_cast(bf8_t, __builtin_nontemporal_load((const u4_t*)(wrow[u] + s * 64)));
        wf[i][u][1] = __builtin_bit_cast(bf8_t, __builtin_nontemporal_load((const u4_t*)(wrow[u] + s * 64 + 8)));
      }
    }
#pragma unroll
    for (int i = 0; i < UW; ++i) {
      const int s = wave + 4 * (c + i < nmy ? c + i : nmy - 1);
#pragma unroll
      for (int t = 0; t < MT; ++t) {
        af[i][t][0] = __builtin_bit_cast(bf8_t, *(const uint4*)(arow[t] + s * 64));
        af[i][t][1] = __builtin_bit_cast(bf8_t, *(const uint4*)(arow[t] + s * 64 + 8));
      }
    }
#pragma unroll
    for (int i = 0; i < UW; ++i) {
      if (c + i < nmy) {
#pragma unroll
        for (int t = 0; t < MT; ++t)
#pragma unroll
          for (int u = 0; u < NT; ++u) {
            acc[u][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[i][u][0], af[i][t][0], acc[u][t], 0, 0, 0);
            acc[u][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[i][u][1], af[i][t][1], acc[u][t], 0, 0, 0);
          }
      }
    }
  }
  if (wave > 0) {
#pragma unroll
    for (int u = 0; u < NT; ++u)
#pragma unroll
      for (int t = 0; t < MT; ++t) part[wave - 1][u * MT + t][lane] = acc[u][t];
  }
  __syncthreads();
  if (wave > 0) return;
#pragma unroll
  for (int u = 0; u < NT; ++u)
#pragma unroll
    for (int t = 0; t < MT; ++t)
      acc[u][t] += part[0][u * MT + t][lane] + part[1][u * MT + t][lane] + part[2][u * MT + t][lane];

#pragma unroll
  for (int u = 0; u < NT; ++u) {
    const int n = n0 + 16 * u + 4 * fg;
    if (n >= N) continue;
    const int nv = N - n < 4 ? N - n : 4;
#pragma unroll
    for (int t = 0; t < MT; ++t) {
      const int m = t * 16 + fr;
      if (m >= M) continue;
      float v[4] = {acc[u][t][0], acc[u][t][1], acc[u][t][2], acc[u][t][3]};
      if (slabs) {
        float* o = slabs + ((long)blockIdx.y * M + m) * N + n;
        if (nv == 4) *(float4*)o = float4{v[0], v[1], v[2], v[3]};
        else for (int r = 0; r < nv; ++r) o[r] = v[r];
        continue;
      }
      if (g.bias) for (int r = 0; r < nv; ++r) v[r] += g.bias[n + r];
      if (EPI == EPI_GELU_BF16) for (int r = 0; r < 4; ++r) v[r] = gelu_f(v[r], g.gelu_tanh);
      const long o = (long)m * g.ldc + n;
      if (EPI == EPI_F32) {
        float* out = (float*)g.out;
        for (int r = 0; r < nv; ++r) out[o + r] = v[r];
      } else {
        bf16_t* out = (bf16_t*)g.out;
        if (nv == 4 && ((o & 3) == 0)) {
          uint2 p;
          p.x = (uint32_t)f2bf(v[0]) | ((uint32_t)f2bf(v[1]) << 16);
          p.y = (uint32_t)f2bf(v[2]) | ((uint32_t)f2bf(v[3]) << 16);
          *(uint2*)(out + o) = p;
        } else {
          for (int r = 0; r < nv; ++r) out[o + r] = f2bf(v[r]);
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Small-M kernel with the preceding LayerNorm as its prologue (M <= 4 rows: a single clip, greedy
// or a few beams).  A decode step of that size is a chain of ~350 kernels of about 5 us each, and
// most of those 5 us are the kernel boundary (dispatch, the first dependent load from memory,
// the write-back at the end), not work: a LayerNorm of one row is such a kernel.  Here every block
// computes the few rows' LayerNorm itself, into LDS, while its first weight loads are in flight,
// and multiplies from there; the arithmetic and its order are those of add_layernorm_row_kernel
// and gemm_skinny_kernel<1, NT, 1, EPI>, so both paths give the same bits.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float gs_wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

template <int NT, int EPI>
__global__ __launch_bounds__(256) void gemm_skinny_ln_kernel(const GemmArgs g, const LnFuse f, int k_per_split,
                                                             float* __restrict__ slabs) {
  __shared__ f4_t part[3][NT][64];
  __shared__ __attribute__((aligned(16))) bf16_t hs[4][2048];
  __shared__ float sh[8];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, fg = lane >> 4;
  const int n0 = blockIdx.x * (16 * NT);
  const int M = g.m, N = g.n, K = g.k, d = g.k;
  const int kbeg = blockIdx.y * k_per_split;
  const bf16_t* __restrict__ wrow[NT];
#pragma unroll
  for (int u = 0; u < NT; ++u) {
    int wn = n0 + 16 * u + fr; wn = wn < N ? wn : N - 1;
    wrow[u] = g.w + (long)wn * K + kbeg + 16 * fg;
  }
  typedef __attribute__((ext_vector_type(4))) unsigned int u4_t;
  const int nsteps = k_per_split / 64;
  const int nmy = (nsteps - wave + 3) >> 2;            // wave-uniform; this wave's steps are wave + 4 c
  // Weights: with one or two feature tiles per block every step of this wave (at most 8: K <= 2048)
  // is requested here, so that all of them travel while the prologue runs; with four tiles that
  // would take 256 registers, and the loop requests one step ahead instead.
  constexpr int kPre = NT <= 2 ? 8 : 1;
  bf8_t wpre[kPre][NT][2];
  auto load_w = [&](int slot, int c) {
    const int s = wave + 4 * (c < nmy ? c : (nmy > 0 ? nmy - 1 : 0));
#pragma unroll
    for (int u = 0; u < NT; ++u) {
      wpre[slot][u][0] = __builtin_bit_cast(bf8_t, __builtin_nontemporal_load((const u4_t*)(wrow[u] + s * 64)));
      wpre[slot][u][1] = __builtin_bit_cast(bf8_t, __builtin_nontemporal_load((const u4_t*)(wrow[u] + s * 64 + 8)));
    }
  };
  if (kPre == 8) {
#pragma unroll
    for (int c = 0; c < 8; ++c) if (c < nmy) load_w(c, c);
  } else if (nmy > 0) {
    load_w(0, 0);
  }

  // ---- prologue: LayerNorm of the M rows into hs (bf16), as add_layernorm_row_kernel does it
  const int nv = d >> 2;
  const bool writer = f.n_slabs > 0 && f.x_out != nullptr && blockIdx.x == 0 && blockIdx.y == 0;
  for (int m = 0; m < M; ++m) {
    const long in_row = (long)m * f.row_mul + f.row_off;
    const float* xr = f.x_in + in_row * d;
    float4 v[2], gm[2], bt[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int c = tid + 256 * i;
      if (c < nv) {
        gm[i] = ((const float4*)f.gamma)[c];
        bt[i] = ((const float4*)f.beta)[c];
      }
    }
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int c = tid + 256 * i;
      if (c < nv) {
        float4 t = ((const float4*)xr)[c];
        if (f.n_slabs > 0) {
          float4 p[8];                                   // all requested before the first is added
#pragma unroll
          for (int s = 0; s < 8; ++s) {
            const int ss = s < f.n_slabs ? s : f.n_slabs - 1;
            p[s] = ((const float4*)(f.slabs + ss * f.slab_stride + in_row * d))[c];
          }
#pragma unroll
          for (int s = 0; s < 8; ++s)
            if (s < f.n_slabs) { t.x += p[s].x; t.y += p[s].y; t.z += p[s].z; t.w += p[s].w; }
          if (f.add_bias) {
            const float4 b = ((const float4*)f.add_bias)[c];
            t.x += b.x; t.y += b.y; t.z += b.z; t.w += b.w;
          }
          if (writer) ((float4*)(f.x_out + in_row * d))[c] = t;
        }
        v[i] = t;
        sum += t.x + t.y + t.z + t.w;
      }
    }
    sum = gs_wave_sum(sum);
    __syncthreads();                                   // sh is free again (previous row)
    if (lane == 0) sh[wave] = sum;
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
    sq = gs_wave_sum(sq);
    if (lane == 0) sh[4 + wave] = sq;
    __syncthreads();
    const float rstd = rsqrtf((sh[4] + sh[5] + sh[6] + sh[7]) / d + f.eps);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int c = tid + 256 * i;
      if (c < nv) {
        const float4 gg = gm[i], b = bt[i];
        float4 o;
        o.x = (v[i].x - mean) * rstd * gg.x + b.x;
        o.y = (v[i].y - mean) * rstd * gg.y + b.y;
        o.z = (v[i].z - mean) * rstd * gg.z + b.z;
        o.w = (v[i].w - mean) * rstd * gg.w + b.w;
        uint2 p;
        p.x = (uint32_t)f2bf(o.x) | ((uint32_t)f2bf(o.y) << 16);
        p.y = (uint32_t)f2bf(o.z) | ((uint32_t)f2bf(o.w) << 16);
        ((uint2*)&hs[m][0])[c] = p;
      }
    }
  }
  __syncthreads();

  // ---- the product, step by step as gemm_skinny_kernel<1, NT, 1, EPI> does it
  f4_t acc[NT];
#pragma unroll
  for (int u = 0; u < NT; ++u) acc[u] = f4_t{0.f, 0.f, 0.f, 0.f};
  const bf16_t* hrow = &hs[fr < M ? fr : M - 1][kbeg + 16 * fg];
  if (kPre == 8) {
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      if (c < nmy) {
        const int s = wave + 4 * c;
        const bf8_t a0 = __builtin_bit_cast(bf8_t, *(const uint4*)(hrow + s * 64));
        const bf8_t a1 = __builtin_bit_cast(bf8_t, *(const uint4*)(hrow + s * 64 + 8));
#pragma unroll
        for (int u = 0; u < NT; ++u) {
          acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wpre[c][u][0], a0, acc[u], 0, 0, 0);
          acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wpre[c][u][1], a1, acc[u], 0, 0, 0);
        }
      }
    }
  } else {
    for (int c = 0; c < nmy; ++c) {
      bf8_t wf[NT][2];
#pragma unroll
      for (int u = 0; u < NT; ++u) { wf[u][0] = wpre[0][u][0]; wf[u][1] = wpre[0][u][1]; }
      if (c + 1 < nmy) load_w(0, c + 1);
      const int s = wave + 4 * c;
      const bf8_t a0 = __builtin_bit_cast(bf8_t, *(const uint4*)(hrow + s * 64));
      const bf8_t a1 = __builtin_bit_cast(bf8_t, *(const uint4*)(hrow + s * 64 + 8));
#pragma unroll
      for (int u = 0; u < NT; ++u) {
        acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[u][0], a0, acc[u], 0, 0, 0);
        acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[u][1], a1, acc[u], 0, 0, 0);
      }
    }
  }
  if (wave > 0) {
#pragma unroll
    for (int u = 0; u < NT; ++u) part[wave - 1][u][lane] = acc[u];
  }
  __syncthreads();
  if (wave > 0) return;
#pragma unroll
  for (int u = 0; u < NT; ++u) acc[u] += part[0][u][lane] + part[1][u][lane] + part[2][u][lane];

#pragma unroll
  for (int u = 0; u < NT; ++u) {
    const int n = n0 + 16 * u + 4 * fg;
    if (n >= N) continue;
    const int nv4 = N - n < 4 ? N - n : 4;
    const int m = fr;
    if (m >= M) continue;
    float v[4] = {acc[u][0], acc[u][1], acc[u][2], acc[u][3]};
    if (slabs) {
      float* o = slabs + ((long)blockIdx.y * M + m) * N + n;
      if (nv4 == 4) *(float4*)o = float4{v[0], v[1], v[2], v[3]};
      else for (int r = 0; r < nv4; ++r) o[r] = v[r];
      continue;
    }
    if (g.bias) for (int r = 0; r < nv4; ++r) v[r] += g.bias[n + r];
    if (EPI == EPI_GELU_BF16) for (int r = 0; r < 4; ++r) v[r] = gelu_f(v[r], g.gelu_tanh);
    const long o = (long)m * g.ldc + n;
    if (EPI == EPI_F32) {
      float* out = (float*)g.out;
      for (int r = 0; r < nv4; ++r) out[o + r] = v[r];
    } else {
      bf16_t* out = (bf16_t*)g.out;
      if (nv4 == 4 && ((o & 3) == 0)) {
        uint2 p;
        p.x = (uint32_t)f2bf(v[0]) | ((uint32_t)f2bf(v[1]) << 16);
        p.y = (uint32_t)f2bf(v[2]) | ((uint32_t)f2bf(v[3]) << 16);
        *(uint2*)(out + o) = p;
      } else {
        for (int r = 0; r < nv4; ++r) out[o + r] = f2bf(v[r]);
      }
    }
  }
}

template <int NT>
static void launch_skinny_ln_nt(const GemmArgs& g, const LnFuse& f, int epilogue, int split_k, float* slabs_out,
                                hipStream_t stream) {
  dim3 grid((g.n + 16 * NT - 1) / (16 * NT), split_k, 1);
  const int kps = g.k / split_k;
  float* sl = split_k > 1 || epilogue < 0 ? slabs_out : nullptr;
  if (sl) hipLaunchKernelGGL((gemm_skinny_ln_kernel<NT, EPI_F32>), grid, dim3(256), 0, stream, g, f, kps, sl);
  else if (epilogue == EPI_GELU_BF16) hipLaunchKernelGGL((gemm_skinny_ln_kernel<NT, EPI_GELU_BF16>), grid, dim3(256), 0, stream, g, f, kps, sl);
  else if (epilogue == EPI_F32) hipLaunchKernelGGL((gemm_skinny_ln_kernel<NT, EPI_F32>), grid, dim3(256), 0, stream, g, f, kps, sl);
  else hipLaunchKernelGGL((gemm_skinny_ln_kernel<NT, EPI_BF16>), grid, dim3(256), 0, stream, g, f, kps, sl);
}

// 64 features per block for the vocabulary projection (as launch_skinny_nt<1> chooses), 16 otherwise:
// the activations sit in LDS here, so a block gains nothing from more features, and FC1 as 320
// blocks of 16 ran faster than as 160 of 32 (1.76 -> 1.73 ms per token).  The features per block
// do not change a result.
void launch_gemm_skinny_ln(const GemmArgs& g, const LnFuse& f, int epilogue, int split_k, float* slabs_out,
                           hipStream_t stream) {
  const long blocks16 = (long)((g.n + 15) / 16) * split_k;
  if (blocks16 >= 8 * 256) launch_skinny_ln_nt<4>(g, f, epilogue, split_k, slabs_out, stream);
  else launch_skinny_ln_nt<1>(g, f, epilogue, split_k, slabs_out, stream);
}

// ---------------------------------------------------------------------------------------------
// Decode-step kernel for 32 < M <= 640 rows: a block owns 64 output features and up to 256 rows,
// so the activations are read once per 64 features instead of once per 16.  Both operands go
// through a three-stage LDS ring filled by LDS-DMA in K chunks of 64 (same 128-byte rows and
// source-side swizzle as the large kernel); stage t + 2 is requested while stage t is being
// multiplied, the wait is a counted vmcnt that leaves the younger stage in flight, and there
// is one (fence-free) barrier per chunk: it publishes stage t and, because every wave has
// finished stage t - 1 by then, frees the slot that the new request overwrites.
// Wave w multiplies row tiles w, w + 4, ... (TM of them) by the 4 feature tiles.
// blockIdx.y selects a K slice; with slabs != null the raw partial sums go to slabs[y][M][N].
// blockIdx.z selects a group of 64 * TM rows: layers whose K cannot be split (bias / GELU
// epilogues) and that have too few 64-feature blocks to cover the chip split their rows.
// ---------------------------------------------------------------------------------------------
template <int TM, int EPI>
__global__ __launch_bounds__(256) void gemm_mid_kernel(const GemmArgs g, int k_per_split,
                                                       float* __restrict__ slabs) {
  constexpr int ROWS = 64 * TM;                       // activation rows held per stage
  constexpr int kStage = (ROWS + 64) * 128;           // bytes
  constexpr int kWOff = ROWS * 128;
  constexpr int G = 2 * TM + 2;                       // DMA instructions per wave and stage
  __shared__ __attribute__((aligned(1024))) unsigned char smem[3 * kStage];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, fg = lane >> 4;
  const int M = g.m, N = g.n, K = g.k;
  const int n0 = blockIdx.x * 64;
  const int kbeg = blockIdx.y * k_per_split;
  const int nt = k_per_split / 64;
  const int row0 = blockIdx.z * ROWS;

  // DMA instruction j covers rows 8j..8j+7 of an operand: lane -> row 8j + lane / 8 and, because
  // the LDS image is written in lane order, the source piece (lane % 8) ^ (lane / 8)
  const int piece = (lane & 7) ^ (lane >> 3);
  const bf16_t* a_src[2 * TM];
  const bf16_t* w_src[2];
#pragma unroll
  for (int i = 0; i < 2 * TM; ++i) {
    int row = row0 + 8 * (wave + 4 * i) + (lane >> 3);
    row = row < M ? row : M - 1;
    a_src[i] = g.a + (long)row * g.lda + kbeg + piece * 8;
  }
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    int wn = n0 + 8 * (wave + 4 * i) + (lane >> 3);
    wn = wn < N ? wn : N - 1;
    w_src[i] = g.w + (long)wn * K + kbeg + piece * 8;
  }
  auto issue = [&](int t) {
    unsigned char* dst = smem + (t % 3) * kStage;
#pragma unroll
    for (int i = 0; i < 2 * TM; ++i)
      __builtin_amdgcn_global_load_lds((gb_glb_void*)(a_src[i] + t * 64),
                                       (gb_lds_void*)(dst + (wave + 4 * i) * 1024), 16, 0, 0);
#pragma unroll
    for (int i = 0; i < 2; ++i)
      __builtin_amdgcn_global_load_lds((gb_glb_void*)(w_src[i] + t * 64),
                                       (gb_lds_void*)(dst + kWOff + (wave + 4 * i) * 1024), 16, 0, 0);
  };

  f4_t acc[4][TM];
#pragma unroll
  for (int u = 0; u < 4; ++u)
#pragma unroll
    for (int t = 0; t < TM; ++t) acc[u][t] = f4_t{0.f, 0.f, 0.f, 0.f};

  issue(0);
  if (nt > 1) issue(1);
  for (int t = 0; t < nt; ++t) {
    // vmcnt(G): everything but the youngest stage has landed; vmcnt(0) on the last chunk
    if (t + 1 < nt) __builtin_amdgcn_s_waitcnt((G & 15) | (7 << 4) | (15 << 8) | ((G >> 4) << 14));
    else __builtin_amdgcn_s_waitcnt(0x0F70);
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (t + 2 < nt) issue(t + 2);
    const unsigned char* cur = smem + (t % 3) * kStage;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int ch = ks * 4 + fg;
      bf8_t wf[4], af[TM];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int rw = u * 16 + fr;
        wf[u] = __builtin_bit_cast(bf8_t, *(const uint4*)(cur + kWOff + rw * 128 + ((ch ^ (rw & 7)) << 4)));
      }
#pragma unroll
      for (int m = 0; m < TM; ++m) {
        const int rm = (wave + 4 * m) * 16 + fr;
        af[m] = __builtin_bit_cast(bf8_t, *(const uint4*)(cur + rm * 128 + ((ch ^ (rm & 7)) << 4)));
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int m = 0; m < TM; ++m)
          acc[u][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[u], af[m], acc[u][m], 0, 0, 0);
    }
    __builtin_amdgcn_s_waitcnt(0xC07F);               // lgkmcnt(0): this wave's LDS reads are done
  }

  // epilogue: lane holds C[m = row0 + 16 (wave + 4 mt) + fr][n = n0 + 16 u + 4 fg + r]
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int n = n0 + 16 * u + 4 * fg;
    if (n >= N) continue;
    const int nv = N - n < 4 ? N - n : 4;
    float bias4[4] = {0.f, 0.f, 0.f, 0.f};
    if (!slabs && g.bias) {
#pragma unroll
      for (int r = 0; r < 4; ++r) bias4[r] = g.bias[n + r < N ? n + r : N - 1];
    }
#pragma unroll
    for (int mt = 0; mt < TM; ++mt) {
      const int m = row0 + (wave + 4 * mt) * 16 + fr;
      if (m >= M) continue;
      float v[4] = {acc[u][mt][0], acc[u][mt][1], acc[u][mt][2], acc[u][mt][3]};
      if (slabs) {
        float* o = slabs + ((long)blockIdx.y * M + m) * N + n;
        if (nv == 4) *(float4*)o = float4{v[0], v[1], v[2], v[3]};
        else {
#pragma unroll
          for (int r = 0; r < 4; ++r) if (r < nv) o[r] = v[r];
        }
        continue;
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] += bias4[r];
      if (EPI == EPI_GELU_BF16) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = gelu_f(v[r], g.gelu_tanh);
      }
      const long o = (long)m * g.ldc + n;
      if (EPI == EPI_F32) {
        float* out = (float*)g.out;
#pragma unroll
        for (int r = 0; r < 4; ++r) if (r < nv) out[o + r] = v[r];
      } else {
        bf16_t* out = (bf16_t*)g.out;
        if (nv == 4 && ((o & 3) == 0)) {
          uint2 p;
          p.x = (uint32_t)f2bf(v[0]) | ((uint32_t)f2bf(v[1]) << 16);
          p.y = (uint32_t)f2bf(v[2]) | ((uint32_t)f2bf(v[3]) << 16);
          *(uint2*)(out + o) = p;
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r) if (r < nv) out[o + r] = f2bf(v[r]);
        }
      }
    }
  }
}

template <int TM>
static void launch_mid_tm(const GemmArgs& g, int epilogue, int split_k, float* slabs, hipStream_t stream) {
  dim3 grid((g.n + 63) / 64, split_k, (g.m + 64 * TM - 1) / (64 * TM));
  const int kps = g.k / split_k;
  float* sl = split_k > 1 || epilogue < 0 ? slabs : nullptr;
  if (sl) hipLaunchKernelGGL((gemm_mid_kernel<TM, EPI_F32>), grid, dim3(256), 0, stream, g, kps, sl);
  else if (epilogue == EPI_GELU_BF16) hipLaunchKernelGGL((gemm_mid_kernel<TM, EPI_GELU_BF16>), grid, dim3(256), 0, stream, g, kps, sl);
  else if (epilogue == EPI_F32) hipLaunchKernelGGL((gemm_mid_kernel<TM, EPI_F32>), grid, dim3(256), 0, stream, g, kps, sl);
  else hipLaunchKernelGGL((gemm_mid_kernel<TM, EPI_BF16>), grid, dim3(256), 0, stream, g, kps, sl);
}

// Split K across blocks only when there are too few 16-feature blocks to cover the chip, and
// never into more than 8 slices (the consumer re-reads every slice).
static int mid_from_tiles() {
  static const int v = getenv("STT_MI355X_MID_FROM") ? atoi(getenv("STT_MI355X_MID_FROM")) : 3;
  return v;
}

int pick_split_k(int m, int n, int k, int n_cu) {
  const int steps = k / 64;
  const int mt = (m + 15) / 16;
  int best = 1;
  if (mt >= mid_from_tiles() && mt <= 40) {
    // 64-feature blocks: split until about half the CUs have a block, at least 4 chunks each
    static const int force_small = getenv("STT_MI355X_SPLIT_K1280") ? atoi(getenv("STT_MI355X_SPLIT_K1280")) : 0;
    static const int force_large = getenv("STT_MI355X_SPLIT_K5120") ? atoi(getenv("STT_MI355X_SPLIT_K5120")) : 0;
    if (k <= 2048 && force_small > 0 && steps % force_small == 0) return force_small;     // (measurement hooks)
    if (k > 2048 && force_large > 0 && steps % force_large == 0) return force_large;
    const int col_blocks = (n + 63) / 64;
    if (mt > 16) {
      // Above 256 rows the rows are dealt to blocks of 128 (launch_gemm_skinny), two of which share
      // a CU: the largest split that keeps every block in one round (<= 2 blocks per CU) and gives
      // it at least 10 K chunks, so that its chunk loop outweighs its fill and its slab writes.
      // Measured at 600 rows of large-v3: K = N = 1280 in 2 slices 9.2 us (5 slices 10.3, 1 slice
      // 12.2) and a cheaper LayerNorm behind it; FC2 (K = 5120) in 5 slices 15.8 us (8 slices: two
      // rounds, 23.1; 4 slices 19.1; 2 slices 21.4).
      const int row_groups = (mt + 7) / 8;
      for (int s = 1; s <= 8 && s <= steps; ++s) {
        if (steps % s) continue;
        if (steps / s < 10 || (long)col_blocks * s * row_groups > 2L * n_cu) break;
        best = s;
      }
      return best;
    }
    for (int s = 1; s <= 8 && s <= steps; ++s) {
      if (steps % s) continue;
      if (steps / s < 4) break;
      best = s;
      if (col_blocks * s * 2 >= n_cu) break;
    }
    return best;
  }
  const int col_blocks = (n + 15) / 16;
  static const int min_steps = getenv("STT_MI355X_SKINNY_MINSTEPS") ? atoi(getenv("STT_MI355X_SKINNY_MINSTEPS")) : 4;
  for (int s = 1; s <= 8 && s <= steps; ++s) {
    if (steps % s) continue;
    if (steps / s < min_steps) break;      // at least one step per wave (8 measured slower: 1.93 against 1.87 ms per token, one clip)
    best = s;
    if ((long)col_blocks * s >= 2L * n_cu) break;
  }
  return best;
}

template <int MT, int NT, int UW>
static void launch_skinny_uw(const GemmArgs& g, int epilogue, int split_k, float* slabs,
                             hipStream_t stream) {
  dim3 grid((g.n + 16 * NT - 1) / (16 * NT), split_k, 1);
  const int kps = g.k / split_k;
  float* sl = split_k > 1 || epilogue < 0 ? slabs : nullptr;
  if (sl) {
    hipLaunchKernelGGL((gemm_skinny_kernel<MT, NT, UW, EPI_F32>), grid, dim3(256), 0, stream, g, kps, sl);
  } else if (epilogue == EPI_GELU_BF16) {
    hipLaunchKernelGGL((gemm_skinny_kernel<MT, NT, UW, EPI_GELU_BF16>), grid, dim3(256), 0, stream, g, kps, sl);
  } else if (epilogue == EPI_F32) {
    hipLaunchKernelGGL((gemm_skinny_kernel<MT, NT, UW, EPI_F32>), grid, dim3(256), 0, stream, g, kps, sl);
  } else {
    hipLaunchKernelGGL((gemm_skinny_kernel<MT, NT, UW, EPI_BF16>), grid, dim3(256), 0, stream, g, kps, sl);
  }
}

// UW = 1 (one step in flight per wave) is the measured optimum at 64 rows: hoisting the loads of
// 3 or 5 steps ahead of the MFMAs costs the registers that let a second block share the CU and
// was slower on every decoder shape of large-v3 (vocabulary projection 78 -> 94 us, FC1
// 17.6 -> 20.1 us); the template parameter stays for smaller row counts.
template <int MT, int NT>
static void launch_skinny_mt(const GemmArgs& g, int epilogue, int split_k, float* slabs,
                             hipStream_t stream) {
  launch_skinny_uw<MT, NT, 1>(g, epilogue, split_k, slabs, stream);
}

// Every block re-reads the whole activation matrix (from L2), so wide layers give a block
// several 16-feature tiles to amortise that read: 64 features when that still leaves two
// blocks per CU (the vocabulary projection), 32 when it leaves at least half a block per CU.
template <int MT>
static void launch_skinny_nt(const GemmArgs& g, int epilogue, int split_k, float* slabs,
                             hipStream_t stream) {
  const long blocks16 = (long)((g.n + 15) / 16) * split_k;
  if (MT <= 4 && blocks16 >= 8 * 256) launch_skinny_mt<MT, (MT <= 4 ? 4 : 1)>(g, epilogue, split_k, slabs, stream);
  else if (MT <= 8 && split_k == 1 && blocks16 >= 256 + SKINNY_NT2_MIN) launch_skinny_mt<MT, (MT <= 8 ? 2 : 1)>(g, epilogue, split_k, slabs, stream);
  else launch_skinny_mt<MT, 1>(g, epilogue, split_k, slabs, stream);
}

// epilogue < 0: always write raw partial sums to slabs (even with split_k == 1).
void launch_gemm_skinny(const GemmArgs& g, int epilogue, int split_k, float* slabs,
                        hipStream_t stream) {
  const int mt = (g.m + 15) / 16;
  if (mt >= mid_from_tiles() && mt <= 40 && (g.k / split_k) % 64 == 0) {
    // All rows in one block (the weights are then read once) unless that leaves fewer blocks than
    // CUs: then the rows per block are halved until the chip is covered.  At 256 rows of
    // large-v3 this took the token step from 13.7 to 12.7 ms (QKV and FC1 had 60 and 80 blocks).
    static const int kFill = getenv("STT_MI355X_MID_FILL") ? atoi(getenv("STT_MI355X_MID_FILL")) : 256;
    static const int force_tm = getenv("STT_MI355X_MID_TM") ? atoi(getenv("STT_MI355X_MID_TM")) : 0;
    int tm = (mt + 3) / 4;
    tm = tm > 4 ? 4 : tm;                  // at most 256 rows per block; more rows, more row groups
    const long col_blocks = (g.n + 63) / 64;
    while (tm > 1 && col_blocks * split_k * ((mt + 4 * tm - 1) / (4 * tm)) < kFill) tm = (tm + 1) / 2;
    // Above 256 rows 128 rows per block are best for every decoder shape of large-v3 (two blocks
    // share a CU and hide each other's chunk barriers): at 512 rows FC2 22.6 -> 18.7 us (the rule
    // above picks 256 rows for it), QKV 15.4 -> 14.4 us (64 rows), FC1 and the K = N projections
    // stay where the rule puts them; 64, 192 and 256 rows per block are slower for all of them.
    if (mt > 16) tm = 2;
    if (force_tm >= 1 && force_tm <= 4) tm = force_tm;
    switch (tm) {
      case 1: launch_mid_tm<1>(g, epilogue, split_k, slabs, stream); return;
      case 2: launch_mid_tm<2>(g, epilogue, split_k, slabs, stream); return;
      case 3: launch_mid_tm<3>(g, epilogue, split_k, slabs, stream); return;
      default: launch_mid_tm<4>(g, epilogue, split_k, slabs, stream); return;
    }
  }
  switch (mt) {
    case 1: launch_skinny_nt<1>(g, epilogue, split_k, slabs, stream); break;
    case 2: launch_skinny_nt<2>(g, epilogue, split_k, slabs, stream); break;
    case 3: launch_skinny_nt<3>(g, epilogue, split_k, slabs, stream); break;
    case 4: launch_skinny_nt<4>(g, epilogue, split_k, slabs, stream); break;
    case 5: case 6: launch_skinny_nt<6>(g, epilogue, split_k, slabs, stream); break;
    case 7: case 8: launch_skinny_nt<8>(g, epilogue, split_k, slabs, stream); break;
    case 9: case 10: launch_skinny_nt<10>(g, epilogue, split_k, slabs, stream); break;
    case 11: case 12: launch_skinny_nt<12>(g, epilogue, split_k, slabs, stream); break;
    default: launch_skinny_nt<16>(g, epilogue, split_k, slabs, stream); break;
  }
}
