// Attention kernels of the Whisper engine (gfx950), head size 64.
//
// Reference semantics (architectures/layers/residual_multi_head_attention.py:167-208):
// softmax(q k^T / sqrt(64), where(mask, ., -inf)) v, computed per head.
#include <cstdlib>

#include "kernels.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf8_t;
typedef __attribute__((ext_vector_type(16))) float f16_t;

__device__ __forceinline__ bf16_t f2bf(float f) {
  uint32_t u = __float_as_uint(f);
  u += 0x7FFFu + ((u >> 16) & 1u);
  return (bf16_t)(u >> 16);
}
__device__ __forceinline__ float bf2f(bf16_t h) { return __uint_as_float(((uint32_t)h) << 16); }
// two floats -> packed bf16 (round to nearest even) in one v_cvt_pk_bf16_f32
typedef __bf16 bf2_t __attribute__((ext_vector_type(2)));
typedef float f2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pack2(float a, float b) {
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(f2_t{a, b}, bf2_t));
}
// ---------------------------------------------------------------------------------------------
// Encoder self-attention (no mask), flash style, one wave = 64 queries, 32 keys per step.
// S^T = K Q^T with v_mfma_f32_32x32x16_bf16 puts a query on each lane (column) and 16 of the 32
// keys in its registers (the other 16 in lane ^ 32), so the softmax statistics are per lane.
// The S^T accumulator, rounded to bf16, is directly the B operand of O^T += V^T P^T, whose k
// order inside a 16-key step is (register j of half h) -> key 16 s + 8 (j >> 2) + 4 h + (j & 3);
// V^T is read from an image [head * 64 + feature][position of the whole batch] with that same order.
//
// q arrives scaled by log2(e) / 8 (the projection's epilogue does it in fp32, before the one
// rounding to bf16), so scores are in base-2 units.  The softmax keeps a per-query reference r
// instead of the exact running maximum m: r <= m <= r + 8.  The accumulator of the score product
// starts at -r, so the matrix cores deliver s - r and a key costs one v_exp_f32, one add and half
// a conversion; only when a tile raises some query's maximum more than lift_at (8) above its
// reference (or on the first tile) does the wave take the branch that moves r and rescales that
// query's sums.  Weights are at most 2^8, at least one of them is >= 1, so fp32 sums neither
// overflow nor vanish.
// Per tile the two 32-query halves are issued as QK(0), QK(1), softmax(0), PV(0), softmax(1),
// PV(1): the matrix cores work on one half while the vector unit does the other's softmax.
// ---------------------------------------------------------------------------------------------
#define AE_K_BYTES (32 * 128)
#define AE_TILE_BYTES (AE_K_BYTES + 64 * 72)
template <int NX>       // query halves of 32 per wave: 2, or 1 when that is what it takes to cover the chip
__global__ __launch_bounds__(256, NX == 2 ? 2 : 3) void attn_encoder_kernel(
    const bf16_t* __restrict__ q, const bf16_t* __restrict__ k, long ld, long batch_stride,
    const bf16_t* __restrict__ vt, long vt_pitch, bf16_t* __restrict__ out, int t, int heads,
    int n_batch, int nx, float lift_at) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = lane & 31, hh = lane >> 5;
  // Blocks are dispatched to the XCDs round robin by their linear index.  The nx query blocks of
  // one (window, head) read the same 384 KB of K and V^T: numbered so that they are consecutive
  // blocks of ONE XCD, they fetch it from memory once and share it in that XCD's L2 (numbered
  // naively they land on nx different XCDs and every one of them pulls its own copy).
  int xb, pair;
  {
    const int n_pairs = heads * n_batch, L = blockIdx.x;
    if ((n_pairs & 7) == 0) {
      const int slot = L >> 3;
      xb = slot % nx;
      pair = (slot / nx) * 8 + (L & 7);
    } else {
      xb = L % nx;
      pair = L / nx;
    }
  }
  const int b = pair / heads, h = pair - b * heads;
  __shared__ __attribute__((aligned(16))) unsigned char tile[2 * AE_TILE_BYTES];
  const int q0 = xb * (128 * NX) + wave * (32 * NX);             // waves past the end still help stage the tiles
  const int d_model = heads * 64;
  const bf16_t* qb = q + b * batch_stride + h * 64;
  const bf16_t* kb = k + b * batch_stride + h * 64;
  const bf16_t* vb = vt + (long)h * 64 * vt_pitch + (long)b * t;   // 8-byte aligned: t % 4 == 0

  bf8_t qf[NX][4];
#pragma unroll
  for (int x = 0; x < NX; ++x) {
    int qi = q0 + 32 * x + c; qi = qi < t ? qi : t - 1;
#pragma unroll
    for (int s = 0; s < 4; ++s)
      qf[x][s] = __builtin_bit_cast(bf8_t, *(const uint4*)(qb + (long)qi * ld + 16 * s + 8 * hh));
  }
  f16_t o[NX][2], nref[NX];                          // nref = -r in all sixteen registers
#pragma unroll
  for (int x = 0; x < NX; ++x) {
#pragma unroll
    for (int r = 0; r < 16; ++r) nref[x][r] = 0.f;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[x][dt][r] = 0.f;
  }
  float lsum[NX];
#pragma unroll
  for (int x = 0; x < NX; ++x) lsum[x] = 0.f;                      // this lane's half of the row sum

  // The block's four waves walk the same keys, so a 32-key tile of K ([32][64], 128-byte rows,
  // 16-byte pieces XOR-swizzled by (row / 2) % 8: two rows fill the 256-byte bank row, and the rows
  // of one parity inside a 16-lane group of ds_read_b128 - lanes 0-3, 12-15, 20-27 and so on - then
  // sit on eight different pieces) and of V^T ([64][32], rows padded to 72 bytes) is staged
  // once per block in LDS (two buffers; the next tile's global loads are in flight during the
  // MFMAs) instead of once per wave from L2.
  const int tid = threadIdx.x;
  const int k_row = tid >> 3, k_piece = tid & 7;          // K tile: 32 rows x 8 pieces
  const int v_row = tid >> 2, v_piece = tid & 3;          // V^T tile: 64 rows x 4 pieces
  const unsigned k_dst = k_row * 128 + ((k_piece ^ ((k_row >> 1) & 7)) << 4);
  const unsigned v_dst = AE_K_BYTES + v_row * 72 + v_piece * 16;
  uint4 k_reg;
  uint2 v_reg[2];
  auto fetch = [&](int key0) {
    int ki = key0 + k_row; ki = ki < t ? ki : t - 1;
    k_reg = *(const uint4*)(kb + (long)ki * ld + 8 * k_piece);
    const bf16_t* vp = vb + (long)v_row * vt_pitch + key0 + 8 * v_piece;
    v_reg[0] = *(const uint2*)vp;          // keys past t belong to the next window (or the
    v_reg[1] = *(const uint2*)(vp + 4);    // padding): finite values under a weight of zero
  };
  auto stash = [&](int buf) {
    unsigned char* base = tile + buf * AE_TILE_BYTES;
    *(uint4*)(base + k_dst) = k_reg;
    *(uint2*)(base + v_dst) = v_reg[0];
    *(uint2*)(base + v_dst + 8) = v_reg[1];
  };
  fetch(0);
  stash(0);
  __syncthreads();

  for (int key0 = 0, it = 0; key0 < t; key0 += 32, ++it) {
    const bool more = key0 + 32 < t;
    if (more) fetch(key0 + 32);
    const unsigned char* cur = tile + (it & 1) * AE_TILE_BYTES;
    bf8_t kf[4];
#pragma unroll
    for (int s = 0; s < 4; ++s)
      kf[s] = __builtin_bit_cast(bf8_t, *(const uint4*)(cur + c * 128 + (((2 * s + hh) ^ ((c >> 1) & 7)) << 4)));
    bf8_t vf[2][2];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        const unsigned char* p = cur + AE_K_BYTES + (32 * dt + c) * 72 + (16 * s2 + 4 * hh) * 2;
        const uint2 lo = *(const uint2*)p, hi = *(const uint2*)(p + 16);
        vf[dt][s2] = __builtin_bit_cast(bf8_t, uint4{lo.x, lo.y, hi.x, hi.y});
      }
    const bool first = it == 0, ragged = key0 + 32 > t;
    f16_t sc[NX];
#pragma unroll
    for (int x = 0; x < NX; ++x) {
      sc[x] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[0], qf[x][0], nref[x], 0, 0, 0);
#pragma unroll
      for (int ss = 1; ss < 4; ++ss)
        sc[x] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[ss], qf[x][ss], sc[x], 0, 0, 0);
    }
#pragma unroll
    for (int x = 0; x < NX; ++x) {
      f16_t& s = sc[x];
      if (ragged) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = key0 + (r & 3) + 8 * (r >> 2) + 4 * hh;
          if (key >= t) s[r] = -__builtin_inff();
        }
      }
      float mx = fmaxf(fmaxf(s[0], s[1]), s[2]);
#pragma unroll
      for (int r = 3; r < 15; r += 2) mx = fmaxf(fmaxf(mx, s[r]), s[r + 1]);
      mx = fmaxf(mx, s[15]);
      if (first || __builtin_amdgcn_ballot_w64(mx > lift_at) != 0) {
        // move the reference of the queries whose maximum ran away (all of them on the first tile)
        const float mq = fmaxf(mx, __shfl_xor(mx, 32, 64));
        float lift = (first || mq > lift_at) ? mq : 0.f;
        if (lift == -__builtin_inff()) lift = 0.f;
        const float alpha = first ? 1.0f : __builtin_amdgcn_exp2f(-lift);
#pragma unroll
        for (int r = 0; r < 16; ++r) { s[r] -= lift; nref[x][r] -= lift; }
        lsum[x] *= alpha;
        o[x][0] *= alpha;
        o[x][1] *= alpha;
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) s[r] = __builtin_amdgcn_exp2f(s[r]);
      // plain adds in two chains: beside MFMAs a packed add costs more issue time than two scalar ones
      float pa = s[0], pb = s[1];
#pragma unroll
      for (int r = 2; r < 16; r += 2) { pa += s[r]; pb += s[r + 1]; }
      lsum[x] += pa + pb;
      bf8_t pf[2];
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        uint4 u;
        u.x = pack2(s[8 * s2 + 0], s[8 * s2 + 1]); u.y = pack2(s[8 * s2 + 2], s[8 * s2 + 3]);
        u.z = pack2(s[8 * s2 + 4], s[8 * s2 + 5]); u.w = pack2(s[8 * s2 + 6], s[8 * s2 + 7]);
        pf[s2] = __builtin_bit_cast(bf8_t, u);
      }
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
          o[x][dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[dt][s2], pf[s2], o[x][dt], 0, 0, 0);
    }
    if (more) stash((it + 1) & 1);
    __syncthreads();
  }
#pragma unroll
  for (int x = 0; x < NX; ++x) {
    const int qi = q0 + 32 * x + c;
    const float l = lsum[x] + __shfl_xor(lsum[x], 32, 64);
    if (qi >= t) continue;
    const float inv = 1.0f / l;
    bf16_t* dst = out + ((long)b * t + qi) * d_model + h * 64;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {      // registers 4g..4g+3 = features 32 dt + 8 g + 4 hh + 0..3
        uint2 p;
        p.x = pack2(o[x][dt][4 * g] * inv, o[x][dt][4 * g + 1] * inv);
        p.y = pack2(o[x][dt][4 * g + 2] * inv, o[x][dt][4 * g + 3] * inv);
        *(uint2*)(dst + 32 * dt + 8 * g + 4 * hh) = p;
      }
  }
}

void launch_attn_encoder(const bf16_t* q, const bf16_t* k, long ld, long batch_stride,
                         const bf16_t* vt, long vt_pitch, bf16_t* out, int batch, int t, int heads,
                         float lift_at, hipStream_t stream) {
  // 256 queries per block, unless that leaves compute units without a block (one or two windows):
  // then 128 queries per block (32 per wave, 9 % slower per score, twice the blocks)
  const int nx = (t + 255) / 256;
  if ((long)nx * heads * batch <= 256) {
    const int nx1 = (t + 127) / 128;
    hipLaunchKernelGGL(attn_encoder_kernel<1>, dim3(nx1 * heads * batch), dim3(256), 0, stream,
                       q, k, ld, batch_stride, vt, vt_pitch, out, t, heads, batch, nx1, lift_at);
    return;
  }
  hipLaunchKernelGGL(attn_encoder_kernel<2>, dim3(nx * heads * batch), dim3(256), 0, stream,
                     q, k, ld, batch_stride, vt, vt_pitch, out, t, heads, batch, nx, lift_at);
}

// ---------------------------------------------------------------------------------------------
// Decoder self-attention over the KV cache: one wave per (sequence, head).  Position p of
// sequence i lives in cache slot slot_of[i][p] (beam search re-links histories instead of
// copying them; generation_utils.py:652-676 gathers the whole cache).  New tokens are appended
// to the sequence's own slot.  Causal inside the new tokens (transformer_arch.py:186-195).
// ---------------------------------------------------------------------------------------------
// Lane = (key group l >> 3, 16-byte piece l & 7): one load instruction covers 8 keys x 128 B and
// SELF_U of them are in flight before any is used; the score of a key is summed over its 8 lanes
// with DPP moves; the weights are exponentiated on the fly in the P V pass.
#define SELF_MAX_LEN 512
#define SELF_U 4
typedef __attribute__((ext_vector_type(4))) unsigned int self_u4;
__device__ __forceinline__ float self_sum8(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));
  return v;
}
__global__ __launch_bounds__(64) void attn_decoder_self_kernel(const SelfAttnArgs a) {
  __shared__ float probs[SELF_MAX_LEN];
  __shared__ int slots[SELF_MAX_LEN];
  __shared__ float qs[64];
  const int lane = threadIdx.x;
  const int piece = lane & 7, kg = lane >> 3;
  const int h = blockIdx.x, i = blockIdx.y;
  if (a.finished && a.finished[i]) return;     // block-uniform: the sequence has ended
  const int d_model = a.heads * 64;
  const int base = a.base_len[i];
  const long slot_stride = (long)a.heads * a.max_len * 64;
  const long head_off = (long)h * a.max_len * 64;
  bf16_t* kown = a.k_cache + (long)i * slot_stride + head_off;
  bf16_t* vown = a.v_cache + (long)i * slot_stride + head_off;
  for (int j = 0; j < a.s_new; ++j) {
    const int p = base + j;
    if (p < a.max_len) {
      const bf16_t* row = a.qkv + (long)(i * a.s_new + j) * 3 * d_model + h * 64;
      kown[(long)p * 64 + lane] = row[d_model + lane];
      vown[(long)p * 64 + lane] = row[2 * d_model + lane];
    }
  }
  const int n_all = base + a.s_new < a.max_len ? base + a.s_new : a.max_len;
  for (int p = lane; p < n_all; p += 64) slots[p] = p >= base ? i : a.slot_of[(long)i * a.max_len + p];
  __syncthreads();
  for (int j = 0; j < a.s_new; ++j) {
    const int n_keys = base + j + 1 < a.max_len ? base + j + 1 : a.max_len;
    qs[lane] = bf2f(a.qkv[(long)(i * a.s_new + j) * 3 * d_model + h * 64 + lane]) * 0.125f;
    __syncthreads();
    float qr[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) qr[e] = qs[8 * piece + e];
    float mx = -__builtin_inff();
    for (int k0 = kg; k0 < n_keys; k0 += 8 * SELF_U) {
      self_u4 u[SELF_U];
#pragma unroll
      for (int t = 0; t < SELF_U; ++t) {
        const int key = k0 + 8 * t < n_keys ? k0 + 8 * t : n_keys - 1;
        u[t] = *(const self_u4*)(a.k_cache + (long)slots[key] * slot_stride + head_off + (long)key * 64 + 8 * piece);
      }
#pragma unroll
      for (int t = 0; t < SELF_U; ++t) {
        const int key = k0 + 8 * t;
        float sv = 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e)
          sv += __uint_as_float(u[t][e] << 16) * qr[2 * e] + __uint_as_float(u[t][e] & 0xFFFF0000u) * qr[2 * e + 1];
        sv = self_sum8(sv);
        if (key < n_keys) {
          mx = fmaxf(mx, sv);
          if (piece == 0) probs[key] = sv;
        }
      }
    }
    mx = fmaxf(mx, __shfl_xor(mx, 8, 64)); mx = fmaxf(mx, __shfl_xor(mx, 16, 64)); mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    __syncthreads();
    float acc[8], sum = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.f;
    for (int k0 = kg; k0 < n_keys; k0 += 8 * SELF_U) {
      self_u4 u[SELF_U];
#pragma unroll
      for (int t = 0; t < SELF_U; ++t) {
        const int key = k0 + 8 * t < n_keys ? k0 + 8 * t : n_keys - 1;
        u[t] = *(const self_u4*)(a.v_cache + (long)slots[key] * slot_stride + head_off + (long)key * 64 + 8 * piece);
      }
#pragma unroll
      for (int t = 0; t < SELF_U; ++t) {
        const int key = k0 + 8 * t;
        const float pw = key < n_keys ? __expf(probs[key] - mx) : 0.f;
        sum += pw;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          acc[2 * e] += pw * __uint_as_float(u[t][e] << 16);
          acc[2 * e + 1] += pw * __uint_as_float(u[t][e] & 0xFFFF0000u);
        }
      }
    }
    sum += __shfl_xor(sum, 8, 64); sum += __shfl_xor(sum, 16, 64); sum += __shfl_xor(sum, 32, 64);
    const float inv = 1.0f / sum;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float v = acc[e];
      v += __shfl_xor(v, 8, 64); v += __shfl_xor(v, 16, 64); v += __shfl_xor(v, 32, 64);
      acc[e] = v * inv;
    }
    if (kg == 0) {
      uint4 o;
      o.x = pack2(acc[0], acc[1]); o.y = pack2(acc[2], acc[3]);
      o.z = pack2(acc[4], acc[5]); o.w = pack2(acc[6], acc[7]);
      *(uint4*)(a.out + (long)(i * a.s_new + j) * d_model + h * 64 + 8 * piece) = o;
    }
    if (a.probs) {
      float* dst = a.probs + (((long)i * a.heads + h) * a.s_new + j) * a.probs_len;
      for (int key = lane; key < a.probs_len; key += 64) dst[key] = key < n_keys ? __expf(probs[key] - mx) * inv : 0.f;
    }
    __syncthreads();
  }
}

void launch_attn_decoder_self(const SelfAttnArgs& a, hipStream_t stream) {
  hipLaunchKernelGGL(attn_decoder_self_kernel, dim3(a.heads, a.n_seq), dim3(64), 0, stream, a);
}

// ---------------------------------------------------------------------------------------------
// Decoder cross-attention: K/V of the encoder output are read once per (window, head) and step
// for all the sequences of that window (its beams), in 1 KiB wave accesses (8 keys x 128 B).
// Lane = (key group l >> 3, 16-byte piece l & 7).  Queries are handled CQ at a time.
// The kernel is a pure HBM stream (384 KB per block and query group), so each wave keeps
// CROSS_U independent 1 KiB loads in flight before it touches any of them; exp() is applied on
// the fly in the P V pass (every lane of a key's 8-lane group recomputes the same weight), which
// leaves two passes and two barriers per query group.
// ---------------------------------------------------------------------------------------------
#define CROSS_MAX_T 1536
typedef __attribute__((ext_vector_type(4))) unsigned int cross_u4;

// sum over the 8 lanes that share a key: xor 1, xor 2 inside the quad, then the mirrored quad
__device__ __forceinline__ float sum8(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));
  return v;
}

// NWV waves per block: 4 when there are enough (window, head) blocks to cover the chip, 16 for a
// few windows (one clip: 20 blocks), where a block of 4 waves cannot keep enough loads in flight
// to pull its 384 KB at more than ~20 GB/s.
template <int CQ, int NWV>
__global__ __launch_bounds__(64 * NWV) void attn_decoder_cross_kernel(const CrossAttnArgs a) {
  constexpr int U = CQ == 1 ? 6 : CQ == 2 ? 4 : 2;     // loads in flight per wave
  __shared__ float sc[CQ][CROSS_MAX_T];
  __shared__ float qs[CQ][64];
  __shared__ float red[NWV][CQ][64];
  __shared__ float wmax[NWV][CQ];
  __shared__ float wsum[NWV][CQ];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int piece = lane & 7, kg = lane >> 3;
  const int h = blockIdx.x, win = blockIdx.y;
  if (a.finished) {                            // block-uniform: every sequence of the window has ended
    int live = 0;
    for (int sq = 0; sq < a.seq_per_win; ++sq) live |= !a.finished[win * a.seq_per_win + sq];
    if (!live) return;
  }
  const int d_model = a.heads * 64;
  const int nq_total = a.seq_per_win * a.s_new;
  const long row0 = (long)win * nq_total;          // first query row of this window
  const bf16_t* kb = a.k + ((long)win * a.heads + h) * a.t_enc * 64 + 8 * piece;
  const bf16_t* vb = a.v + ((long)win * a.heads + h) * a.t_enc * 64 + 8 * piece;
  const int T = a.t_enc;

  for (int qc = 0; qc < nq_total; qc += CQ) {
    const int nq = nq_total - qc < CQ ? nq_total - qc : CQ;
    for (int e = tid; e < CQ * 64; e += 64 * NWV) {
      const int qi = e >> 6, d = e & 63;
      float qv = 0.f;
      if (qi < nq) {
        const long at = (row0 + qc + qi) * d_model + h * 64 + d;
        if (a.q_slabs) {
          qv = a.q_bias ? a.q_bias[h * 64 + d] : 0.f;
          float ps[8];                                   // at most 8 slabs: requested together, added in order
#pragma unroll
          for (int sl = 0; sl < 8; ++sl) ps[sl] = a.q_slabs[(sl < a.n_q_slabs ? sl : a.n_q_slabs - 1) * a.q_slab_stride + at];
#pragma unroll
          for (int sl = 0; sl < 8; ++sl) if (sl < a.n_q_slabs) qv += ps[sl];
        } else {
          qv = bf2f(a.q[at]);
        }
      }
      qs[qi][d] = qv * 0.125f;
    }
    __syncthreads();
    float qr[CQ][8];
#pragma unroll
    for (int qi = 0; qi < CQ; ++qi)
#pragma unroll
      for (int e = 0; e < 8; ++e) qr[qi][e] = qs[qi][8 * piece + e];

    // scores (and this thread's running maximum per query)
    float mx[CQ];
#pragma unroll
    for (int qi = 0; qi < CQ; ++qi) mx[qi] = -__builtin_inff();
    for (int k0 = wave * 8 + kg; k0 < T; k0 += 8 * NWV * U) {
      cross_u4 u[U];
#pragma unroll
      for (int j = 0; j < U; ++j) {
        const int key = k0 + 8 * NWV * j < T ? k0 + 8 * NWV * j : T - 1;
        u[j] = __builtin_nontemporal_load((const cross_u4*)(kb + (long)key * 64));
      }
#pragma unroll
      for (int j = 0; j < U; ++j) {
        const int key = k0 + 8 * NWV * j;
        float kv[8];
#pragma unroll
        for (int e = 0; e < 4; ++e) { kv[2 * e] = __uint_as_float(u[j][e] << 16); kv[2 * e + 1] = __uint_as_float(u[j][e] & 0xFFFF0000u); }
#pragma unroll
        for (int qi = 0; qi < CQ; ++qi) {
          float sv = 0.f;
          if (CQ == 1) {
#pragma unroll
            for (int e = 0; e < 8; ++e) sv += kv[e] * qr[qi][e];
          } else {
            // several queries per key are bound by the vector unit, not by the stream: the dot
            // product as four packed multiply-adds (even and odd elements in two chains)
            f2_t s2 = f2_t{0.f, 0.f};
#pragma unroll
            for (int e = 0; e < 8; e += 2) s2 += f2_t{kv[e], kv[e + 1]} * f2_t{qr[qi][e], qr[qi][e + 1]};
            sv = s2[0] + s2[1];
          }
          sv = sum8(sv);
          if (key < T) {
            mx[qi] = fmaxf(mx[qi], sv);
            if (piece == 0) sc[qi][key] = sv;
          }
        }
      }
    }
#pragma unroll
    for (int qi = 0; qi < CQ; ++qi) {
      float m = mx[qi];
      m = fmaxf(m, __shfl_xor(m, 8, 64)); m = fmaxf(m, __shfl_xor(m, 16, 64)); m = fmaxf(m, __shfl_xor(m, 32, 64));
      if (lane == 0) wmax[wave][qi] = m;
    }
    __syncthreads();
#pragma unroll
    for (int qi = 0; qi < CQ; ++qi) {
      float m = wmax[0][qi];
#pragma unroll
      for (int w = 1; w < NWV; ++w) m = fmaxf(m, wmax[w][qi]);
      mx[qi] = m;
    }

    // P V with the weights exponentiated on the fly
    float acc[CQ][8], sum[CQ];
#pragma unroll
    for (int qi = 0; qi < CQ; ++qi) {
      sum[qi] = 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[qi][e] = 0.f;
    }
    for (int k0 = wave * 8 + kg; k0 < T; k0 += 8 * NWV * U) {
      cross_u4 u[U];
#pragma unroll
      for (int j = 0; j < U; ++j) {
        const int key = k0 + 8 * NWV * j < T ? k0 + 8 * NWV * j : T - 1;
        u[j] = __builtin_nontemporal_load((const cross_u4*)(vb + (long)key * 64));
      }
#pragma unroll
      for (int j = 0; j < U; ++j) {
        const int key = k0 + 8 * NWV * j;
        float vv[8];
#pragma unroll
        for (int e = 0; e < 4; ++e) { vv[2 * e] = __uint_as_float(u[j][e] << 16); vv[2 * e + 1] = __uint_as_float(u[j][e] & 0xFFFF0000u); }
#pragma unroll
        for (int qi = 0; qi < CQ; ++qi) {
          const float pw = key < T ? __expf(sc[qi][key] - mx[qi]) : 0.f;
          sum[qi] += pw;
          if (CQ == 1) {
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[qi][e] += pw * vv[e];
          } else {                 // the same products and sums, two per instruction
#pragma unroll
            for (int e = 0; e < 8; e += 2) {
              const f2_t r = f2_t{acc[qi][e], acc[qi][e + 1]} + f2_t{pw, pw} * f2_t{vv[e], vv[e + 1]};
              acc[qi][e] = r[0]; acc[qi][e + 1] = r[1];
            }
          }
        }
      }
    }
#pragma unroll
    for (int qi = 0; qi < CQ; ++qi) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float v = acc[qi][e];
        v += __shfl_xor(v, 8, 64); v += __shfl_xor(v, 16, 64); v += __shfl_xor(v, 32, 64);
        if (kg == 0) red[wave][qi][8 * piece + e] = v;
      }
      float sm = sum[qi];
      sm += __shfl_xor(sm, 8, 64); sm += __shfl_xor(sm, 16, 64); sm += __shfl_xor(sm, 32, 64);
      if (lane == 0) wsum[wave][qi] = sm;
    }
    __syncthreads();
    for (int e = tid; e < nq * 64; e += 64 * NWV) {
      const int qi = e >> 6, d = e & 63;
      float den = 0.f, num = 0.f;
      if (NWV == 4) {            // the summation order of the four-wave kernel as it always was
        den = wsum[0][qi] + wsum[1][qi] + wsum[2][qi] + wsum[3][qi];
        num = red[0][qi][d] + red[1][qi][d] + red[2][qi][d] + red[3][qi][d];
      } else {
#pragma unroll
        for (int w = 0; w < NWV; ++w) { den += wsum[w][qi]; num += red[w][qi][d]; }
      }
      const float inv = 1.0f / den;
      const float v = num * inv;
      a.out[(row0 + qc + qi) * d_model + h * 64 + d] = f2bf(v);
    }
    if (a.probs) {
#pragma unroll
      for (int qi = 0; qi < CQ; ++qi) {
        if (qi >= nq) break;
        float den = 0.f;
#pragma unroll
        for (int w = 0; w < NWV; ++w) den += wsum[w][qi];
        const float inv = 1.0f / den;
        const long row = row0 + qc + qi, seq = row / a.s_new, j = row - seq * a.s_new;
        float* dst = a.probs + ((seq * a.heads + h) * a.s_new + j) * T;
        for (int key = tid; key < T; key += 64 * NWV) dst[key] = __expf(sc[qi][key] - mx[qi]) * inv;
      }
    }
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------------------
// Decoder cross-attention for several queries per window (beam search: beams x new tokens <= 16)
// on the matrix cores.  The one-query kernel above is a pure stream; with several queries its
// per-(key, query) dot products, lane sums and exponentials are vector work that caps it near
// 4 TB/s.  Here a block = (head, window) still reads the window's K and V once; its four waves
// take 32-key tiles round robin and each keeps a flash-style running softmax of its own keys:
//   S^T = K Q^T   v_mfma_f32_16x16x32_bf16, A = K tile (rows = keys; fetched in whole 1 KiB wave
//                 accesses into a wave-private LDS image and read back in the operand layout - loading
//                 the operand layout straight from memory touches sixteen 128-byte lines per quarter
//                 wave for 16 bytes each, 127 against 1xx us per launch), B = Q^T (the
//                 window's queries padded to 16 columns, scaled by log2(e) / 8, rounded to bf16).
//                 A lane then holds query (lane % 16) against 8 keys.  The two products of a tile
//                 take the keys 8a + b and 8a + 4 + b (a, b < 4), so that registers j = 0..7 of
//                 lane group g = lane / 16 are the keys 8g + j in order:
//   O^T += V^T P^T  exactly the k order of the B operand P^T (the exponentiated scores, rounded to
//                 bf16, never leave their registers).  A = V^T: the V tile goes through a
//                 wave-private LDS image [32 keys][64 features] (144-byte rows) as it comes from
//                 memory and is read back transposed by ds_read_b64_tr_b16.
// All statistics of a query (maximum, sum, rescale factor) live in the lanes of its column, so the
// running rescale is lane-local; only the tile maximum crosses lanes (two shuffles).  The four
// waves' partial (maximum, sum, O^T) are merged through LDS at the end.
// ---------------------------------------------------------------------------------------------
typedef __attribute__((ext_vector_type(4))) float cm_f4;
typedef __attribute__((ext_vector_type(4))) short cm_s4;
#define CM_ROW 144                       // bytes per key row of the V image (128 + 16: spreads the banks)
#define CM_BUF (32 * CM_ROW)

__global__ __launch_bounds__(256, 3) void attn_decoder_cross_mfma_kernel(const CrossAttnArgs a) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[4][2 * CM_BUF];      // per wave: K image | V image
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int col = lane & 15, g = lane >> 4;
  const int h = blockIdx.x, win = blockIdx.y;
  if (a.finished) {                            // block-uniform: every sequence of the window has ended
    int live = 0;
    for (int sq = 0; sq < a.seq_per_win; ++sq) live |= !a.finished[win * a.seq_per_win + sq];
    if (!live) return;
  }
  const int d_model = a.heads * 64;
  const int nq = a.seq_per_win * a.s_new;          // <= 16
  const long row0 = (long)win * nq;
  const int T = a.t_enc;
  const bf16_t* kb = a.k + ((long)win * a.heads + h) * T * 64;
  const bf16_t* vb = a.v + ((long)win * a.heads + h) * T * 64;
  unsigned char* my = lds[wave];

  // B operand Q^T: column = query `col`, k = features 32 c + 8 g .. + 7
  bf8_t qf[2];
  {
    const float scale = 0.125f * 1.4426950408889634f;
    float qv[2][8];
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int e = 0; e < 8; ++e) qv[c][e] = 0.f;
    if (col < nq) {
      const long at = (row0 + col) * d_model + h * 64 + 8 * g;
      if (a.q_slabs) {
        // at most 8 slabs: all requested before any is added (a loop of dependent loads would cost a
        // memory round trip per slab), added in slab order like the one-query kernel does
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          cm_f4 ps[8][2];
#pragma unroll
          for (int sl = 0; sl < 8; ++sl) {
            const float* src = a.q_slabs + (sl < a.n_q_slabs ? sl : a.n_q_slabs - 1) * a.q_slab_stride + at + 32 * c;
            ps[sl][0] = *(const cm_f4*)src;
            ps[sl][1] = *(const cm_f4*)(src + 4);
          }
          if (a.q_bias) {
            const cm_f4 b0 = *(const cm_f4*)(a.q_bias + h * 64 + 32 * c + 8 * g), b1 = *(const cm_f4*)(a.q_bias + h * 64 + 32 * c + 8 * g + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) { qv[c][e] = b0[e]; qv[c][4 + e] = b1[e]; }
          }
#pragma unroll
          for (int sl = 0; sl < 8; ++sl)
            if (sl < a.n_q_slabs) {
#pragma unroll
              for (int e = 0; e < 4; ++e) { qv[c][e] += ps[sl][0][e]; qv[c][4 + e] += ps[sl][1][e]; }
            }
        }
      } else {
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          const uint4 raw = *(const uint4*)(a.q + at + 32 * c);
          const uint32_t w4[4] = {raw.x, raw.y, raw.z, raw.w};
#pragma unroll
          for (int e = 0; e < 4; ++e) { qv[c][2 * e] = __uint_as_float(w4[e] << 16); qv[c][2 * e + 1] = __uint_as_float(w4[e] & 0xFFFF0000u); }
        }
      }
    }
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      uint4 u;
      u.x = pack2(qv[c][0] * scale, qv[c][1] * scale); u.y = pack2(qv[c][2] * scale, qv[c][3] * scale);
      u.z = pack2(qv[c][4] * scale, qv[c][5] * scale); u.w = pack2(qv[c][6] * scale, qv[c][7] * scale);
      qf[c] = __builtin_bit_cast(bf8_t, u);
    }
  }

  const int n_tiles = (T + 31) / 32;
  // A operand rows of the two score products: keys 8 (col / 4) + col % 4 and that + 4
  const int krow = 8 * (col >> 2) + (col & 3);
  unsigned char* kimg = my;
  unsigned char* img = my + CM_BUF;
  // Both tiles are fetched as the 4 KB they are in memory (lane -> consecutive 16-byte pieces, 1 KiB
  // per wave instruction) and land in registers one tile ahead of their use.
  auto load_tile = [&](int tile, uint4 (&kk)[4], uint4 (&vv)[4]) {
    const int k0 = tile * 32;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int ci = lane + 64 * i;                 // 16-byte piece of the 4 KB tile
      int key = k0 + (ci >> 3);
      key = key < T ? key : T - 1;
      kk[i] = __builtin_bit_cast(uint4, __builtin_nontemporal_load((const cross_u4*)(kb + (long)key * 64 + 8 * (ci & 7))));
      vv[i] = __builtin_bit_cast(uint4, __builtin_nontemporal_load((const cross_u4*)(vb + (long)key * 64 + 8 * (ci & 7))));
    }
  };

  cm_f4 o[4];
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) o[nt] = cm_f4{0.f, 0.f, 0.f, 0.f};
  float m = -__builtin_inff(), lsum = 0.f;
  // one tile: LDS images, scores, running softmax, O^T += V^T P^T
  auto process = [&](int tile, const uint4 (&kk)[4], const uint4 (&vv)[4]) {
    // (the wave's LDS operations execute in order: this tile's writes follow the last tile's reads)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int ci = lane + 64 * i;
      *(uint4*)(kimg + (ci >> 3) * CM_ROW + (ci & 7) * 16) = kk[i];
      *(uint4*)(img + (ci >> 3) * CM_ROW + (ci & 7) * 16) = vv[i];
    }
    // scores, base-2 units: registers r of s1 / s2 are keys k0 + 8 g + r and k0 + 8 g + 4 + r
    cm_f4 s1 = cm_f4{0.f, 0.f, 0.f, 0.f}, s2 = s1;
    {
      const unsigned char* ka = kimg + krow * CM_ROW + 16 * g;
      const bf8_t a00 = __builtin_bit_cast(bf8_t, *(const uint4*)(ka));
      const bf8_t a01 = __builtin_bit_cast(bf8_t, *(const uint4*)(ka + 64));
      const bf8_t a10 = __builtin_bit_cast(bf8_t, *(const uint4*)(ka + 4 * CM_ROW));
      const bf8_t a11 = __builtin_bit_cast(bf8_t, *(const uint4*)(ka + 4 * CM_ROW + 64));
      s1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a00, qf[0], s1, 0, 0, 0);
      s1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a01, qf[1], s1, 0, 0, 0);
      s2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a10, qf[0], s2, 0, 0, 0);
      s2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a11, qf[1], s2, 0, 0, 0);
    }
    const int kfirst = tile * 32 + 8 * g;
    if (tile * 32 + 32 > T) {                       // ragged last tile (wave-uniform)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if (kfirst + r >= T) s1[r] = -__builtin_inff();
        if (kfirst + 4 + r >= T) s2[r] = -__builtin_inff();
      }
    }
    float tm = fmaxf(fmaxf(fmaxf(s1[0], s1[1]), fmaxf(s1[2], s1[3])), fmaxf(fmaxf(s2[0], s2[1]), fmaxf(s2[2], s2[3])));
    tm = fmaxf(tm, __shfl_xor(tm, 16, 64));
    tm = fmaxf(tm, __shfl_xor(tm, 32, 64));
    const float mn = fmaxf(m, tm);                  // finite: every tile holds at least one key
    const float alpha = __builtin_amdgcn_exp2f(m - mn);
    m = mn;
    float p[8];
#pragma unroll
    for (int r = 0; r < 4; ++r) { p[r] = __builtin_amdgcn_exp2f(s1[r] - mn); p[4 + r] = __builtin_amdgcn_exp2f(s2[r] - mn); }
    lsum = lsum * alpha + (((p[0] + p[1]) + (p[2] + p[3])) + ((p[4] + p[5]) + (p[6] + p[7])));
    uint4 pu;
    pu.x = pack2(p[0], p[1]); pu.y = pack2(p[2], p[3]); pu.z = pack2(p[4], p[5]); pu.w = pack2(p[6], p[7]);
    const bf8_t pf = __builtin_bit_cast(bf8_t, pu);
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      // transposed read: lane 4 q + p of a 16-lane group addresses key row 8 g + 4 half + q, features
      // 16 nt + 4 p ..; lane i receives feature 16 nt + i of the four keys
      const unsigned char* at = img + (8 * g + (col >> 2)) * CM_ROW + (16 * nt + 4 * (col & 3)) * 2;
      const cm_s4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) cm_s4*)at);
      const cm_s4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) cm_s4*)(at + 4 * CM_ROW));
      uint4 vu;
      const uint2 l2 = __builtin_bit_cast(uint2, lo), h2 = __builtin_bit_cast(uint2, hi);
      vu.x = l2.x; vu.y = l2.y; vu.z = h2.x; vu.w = h2.y;
      o[nt] = o[nt] * alpha;
      o[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf8_t, vu), pf, o[nt], 0, 0, 0);
    }
  };
  // Two register sets alternate between "in flight" and "being used", so the loads of a wave's
  // next tile are outstanding during the whole of the current one (a single set copied at the top
  // of the loop makes the compiler wait for the new loads before it loops).
  uint4 ka[4], va[4], kb2[4], vb2[4];
  if (wave < n_tiles) load_tile(wave, ka, va);
  for (int tile = wave; tile < n_tiles; tile += 8) {
    if (tile + 4 < n_tiles) load_tile(tile + 4, kb2, vb2);
    process(tile, ka, va);
    if (tile + 8 < n_tiles) load_tile(tile + 8, ka, va);
    if (tile + 4 < n_tiles) process(tile + 4, kb2, vb2);
  }
  lsum += __shfl_xor(lsum, 16, 64);
  lsum += __shfl_xor(lsum, 32, 64);
  // merge the waves: red[feature][query] of this wave in its own (now idle) LDS region
  float* red = (float*)my;                           // [64][16] floats = 4 KB, then m[16], l[16]
#pragma unroll
  for (int nt = 0; nt < 4; ++nt)
#pragma unroll
    for (int r = 0; r < 4; ++r) red[(16 * nt + 4 * g + r) * 16 + col] = o[nt][r];
  if (g == 0) { red[1024 + col] = m; red[1040 + col] = lsum; }
  __syncthreads();
  for (int e = tid; e < nq * 64; e += 256) {
    const int qi = e >> 6, d = e & 63;
    float mw[4], mx = -__builtin_inff();
#pragma unroll
    for (int w = 0; w < 4; ++w) { mw[w] = ((const float*)lds[w])[1024 + qi]; mx = fmaxf(mx, mw[w]); }
    float num = 0.f, den = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const float f = __builtin_amdgcn_exp2f(mw[w] - mx);
      num += f * ((const float*)lds[w])[d * 16 + qi];
      den += f * ((const float*)lds[w])[1040 + qi];
    }
    a.out[(row0 + qi) * d_model + h * 64 + d] = f2bf(num / den);
  }
}

void launch_attn_decoder_cross(const CrossAttnArgs& a, hipStream_t stream) {
  const int nq = a.seq_per_win * a.s_new;
  const dim3 grid(a.heads, a.n_win);
  static const bool no_mfma = getenv("STT_MI355X_CROSS_VALU") != nullptr;      // A/B and test hook
  if (nq >= 3 && nq <= 16 && !no_mfma && !a.probs) {      // (the weights are written by the vector-unit kernel)
    hipLaunchKernelGGL(attn_decoder_cross_mfma_kernel, grid, dim3(256), 0, stream, a);
    return;
  }
  if (a.heads * a.n_win <= 96 && nq <= 2) {          // a few windows: sixteen waves per block
    if (nq == 1) hipLaunchKernelGGL((attn_decoder_cross_kernel<1, 16>), grid, dim3(1024), 0, stream, a);
    else hipLaunchKernelGGL((attn_decoder_cross_kernel<2, 16>), grid, dim3(1024), 0, stream, a);
    return;
  }
  if (nq == 1) hipLaunchKernelGGL((attn_decoder_cross_kernel<1, 4>), grid, dim3(256), 0, stream, a);
  else if (nq == 2) hipLaunchKernelGGL((attn_decoder_cross_kernel<2, 4>), grid, dim3(256), 0, stream, a);
  else if (nq <= 4) hipLaunchKernelGGL((attn_decoder_cross_kernel<4, 4>), grid, dim3(256), 0, stream, a);
  else if (nq <= 6 || (nq > 8 && nq % 8 != 0 && nq % 6 == 0))
    hipLaunchKernelGGL((attn_decoder_cross_kernel<6, 4>), grid, dim3(256), 0, stream, a);
  else hipLaunchKernelGGL((attn_decoder_cross_kernel<8, 4>), grid, dim3(256), 0, stream, a);
}
