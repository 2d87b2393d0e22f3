// Whisper log-mel frontend for gfx950 (MI355X).
//
// Replaces, for filter_length = win_length = 400 and hop_length = 160, the chain
//   utils/audio/stft.py:238-270  STFT.transform  (reflect pad, windowed DFT as a
//                                strided conv1d, magnitude)
//   utils/audio/stft.py:346-360  WhisperSTFT.mel_spectrogram (drop last frame,
//                                mel projection, log10 / clamp, global max-8,
//                                (x+4)/4)
// with two kernels:
//   logmel_pair      two lanes = one frame (lane l and lane l + 32 of a wave).  A
//                    256-lane workgroup owns a tile of 118 consecutive frames whose
//                    PCM span is brought into LDS once (LDS-DMA, each sample leaves
//                    HBM once apart from a 240-sample halo per tile), the next
//                    tile's span is fetched while the current one is transformed in
//                    registers (gen_logmel.py), and every lane writes its part of
//                    its frame's output row.  It stores z = (log10(mel)+4)/4, the
//                    clip maximum of log10(mel) and, per wave, the minimum z.
//   logmel_finalize  applies max(z, ((gmax-8)+4)/4) only to the 32-row segments
//                    whose recorded minimum is below the threshold.  x -> (x+4)/4
//                    is monotone in float arithmetic, so clamping after the
//                    scaling gives bit-identical results to clamping before it.
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef const float __attribute__((address_space(4)))* lm_const_ptr;   // scalar-load (constant) view of the table

#define LM_HOP 160
#define LM_NFFT 400
#define LM_PITCH 164                       // floats per LDS row (160 + 4 pad: rows stay 16-byte aligned)
#define LM_CHUNKS_PER_ROW 41               // 16-byte pieces per row (40 data + 1 pad)

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ float wave_min(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o, 64));
  return v;
}

#ifndef LM_DMA_PARTS
#define LM_DMA_PARTS 4
#endif
struct RowDma {
  const char* base;     // uniform: &pcm[g0r + wave * 160]
  unsigned lds0;        // uniform: LDS byte address of row `wave`
  unsigned lane_off;    // per lane: source byte offset inside the instruction
};
// s_waitcnt immediates (vmcnt[3:0] | expcnt[6:4] | lgkmcnt[11:8] | vmcnt[15:14])
#define LM_WAIT_VM0 0x0F70      // vector memory (incl. LDS-DMA) drained
#define LM_WAIT_LGKM0 0xC07F    // LDS / scalar loads drained

// Workgroup barrier without the memory-model fences of __syncthreads(): the loop below
// orders its LDS traffic itself and must not wait for output stores still in flight.
__device__ __forceinline__ void block_barrier() {
  asm volatile("" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

struct LogmelArgs {
  const float* pcm;      // pcm[0] = sample `origin` of the clip
  long origin;           // clip index of pcm[0]
  long lo, hi_al;        // DMA-readable clip range [lo, hi_al): lo = origin, hi_al = origin + 4k
  long n_total;          // clip length
  long frame_begin;      // first frame of this piece
  long frame_count;      // frames in this piece
  int n_tiles;
  const float* ct;       // constant table (scalar loads): DFT constants, mel weights
  const float* lds_table;  // window and twiddle tables, copied into LDS by every workgroup
  float* out;            // [frame_count][NMELS], row 0 = frame_begin
  float* piece_max;      // written by the block that finishes last
  float* seg_min;        // [n_tiles * 4]
  unsigned* ticket;      // plan-owned, zero between launches
  float* blk_max;        // plan-owned, one slot per block
  float* dbg;            // development builds (LM_DEBUG): per-lane dump of the first tile
};

// =============================================================================================
// Two lanes per frame (lane l and lane l + 32 of a wave): the state a lane carries through the
// transform is 200 floats, the kernel fits 256 registers and two waves share a SIMD, so one
// wave's scalar instructions, waits and hazards issue under the other's arithmetic.  A 256-lane
// workgroup owns a tile of 118 frames (80 KB of LDS: two workgroups per CU); the dataflow inside
// a lane pair is described in gen_logmel.py.
// =============================================================================================
__device__ __forceinline__ void lm_swap(float& vdst, float& src0) {
  // v_permlane32_swap: vdst of lanes 32..63 <-> src0 of lanes 0..31.  Written as an asm statement
  // because __builtin_amdgcn_permlane32_swap loses its second result in this compiler (both
  // elements of the returned pair read the first one); the leading s_nop covers the two wait
  // states a permlane swap needs after a VALU write of its operands, which the compiler does not
  // insert in front of an asm statement.
  asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(vdst), "+v"(src0));
}

// four swaps behind one leading s_nop (the operands of the later ones are older still)
__device__ __forceinline__ void lm_swap4(float& v0, float& s0, float& v1, float& s1, float& v2, float& s2,
                                          float& v3, float& s3) {
  asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\tv_permlane32_swap_b32 %2, %3\n\t"
      "v_permlane32_swap_b32 %4, %5\n\tv_permlane32_swap_b32 %6, %7"
      : "+v"(v0), "+v"(s0), "+v"(v1), "+v"(s1), "+v"(v2), "+v"(s2), "+v"(v3), "+v"(s3));
}

// (|A + W B|^2, |A - W B|^2) for two independent (A, B, W): r / i hold (A, B) real / imaginary
// parts in their two halves, cs = (cos, sin) with W = cos - i sin.  The two sequences are
// interleaved: a packed result may not be consumed by the very next instruction, and the compiler
// does not pad hazards inside (or right after) an asm statement.
__device__ __forceinline__ void lm_combine2(v2f csa, v2f ra, v2f ia, v2f& ma, v2f csb, v2f rb, v2f ib,
                                             v2f& mb) {
  v2f ta, tb, ja, jb;
  asm("s_nop 0\n\t"
      "v_pk_mul_f32 %[ta], %[csa], %[ra] op_sel:[0,1] op_sel_hi:[1,1]\n\t"
      "v_pk_mul_f32 %[tb], %[csb], %[rb] op_sel:[0,1] op_sel_hi:[1,1]\n\t"
      "v_pk_fma_f32 %[ta], %[csa], %[ia], %[ta] op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[0,0,0] neg_hi:[0,0,1]\n\t"
      "v_pk_fma_f32 %[tb], %[csb], %[ib], %[tb] op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[0,0,0] neg_hi:[0,0,1]\n\t"
      "v_pk_add_f32 %[ma], %[ra], %[ta] op_sel:[0,0] op_sel_hi:[0,0] neg_lo:[0,0] neg_hi:[0,1]\n\t"
      "v_pk_add_f32 %[mb], %[rb], %[tb] op_sel:[0,0] op_sel_hi:[0,0] neg_lo:[0,0] neg_hi:[0,1]\n\t"
      "v_pk_add_f32 %[ja], %[ia], %[ta] op_sel:[0,1] op_sel_hi:[0,1] neg_lo:[0,0] neg_hi:[0,1]\n\t"
      "v_pk_add_f32 %[jb], %[ib], %[tb] op_sel:[0,1] op_sel_hi:[0,1] neg_lo:[0,0] neg_hi:[0,1]\n\t"
      "v_pk_mul_f32 %[ma], %[ma], %[ma]\n\t"
      "v_pk_mul_f32 %[mb], %[mb], %[mb]\n\t"
      "v_pk_fma_f32 %[ma], %[ja], %[ja], %[ma]\n\t"
      "v_pk_fma_f32 %[mb], %[jb], %[jb], %[mb]\n\t"
      "s_nop 0"
      : [ma] "=&v"(ma), [mb] "=&v"(mb), [ta] "=&v"(ta), [tb] "=&v"(tb), [ja] "=&v"(ja), [jb] "=&v"(jb)
      : [csa] "v"(csa), [ra] "v"(ra), [ia] "v"(ia), [csb] "v"(csb), [rb] "v"(rb), [ib] "v"(ib));
}

#include "logmel_codelet.inc"

#define LM_TILE_FRAMES 118
#define LM_ROWS (LM_TILE_FRAMES + 2)
#define LM_ROW_BYTES (LM_PITCH * 4)
#define LM_TABLE_OFF (LM_ROWS * LM_ROW_BYTES)                      // 78720
#define LM_LDS_TABLE_BYTES (LOGMEL_WIN_BYTES + LOGMEL_TW_BYTES)    // 2912
#define LM_PIECES (LM_ROWS * LM_CHUNKS_PER_ROW)
#define LM_DMA_PER_WAVE ((LM_PIECES + 255) / 256)
#define LM_ROW_DMA_PER_WAVE ((LM_ROWS + 3) / 4)

// Lane-clamped transfer of one tile's PCM span into LDS: instruction inst = wave + 4 i writes
// 1 KiB = 64 pieces of 16 bytes; LDS piece q = row q / 41, piece q % 41 of that row (the pad piece
// of a row re-reads the row's last data piece).  g0r = index, relative to pcm[0], of the tile's
// first sample (may be negative); source indices are clamped into [0, hi_r - 4] (hi_r a multiple
// of 4, so every piece is a 16-byte aligned load); what lies outside is repaired by
// patch_reflect().  Pieces past the last row are not written.
__device__ __forceinline__ void issue_tile_dma(const float* __restrict__ pcm, int g0r, int hi_r,
                                                unsigned char* lds, int wave, int lane) {
#pragma unroll 4
  for (int i = 0; i < LM_DMA_PER_WAVE; ++i) {
    const int inst = wave + 4 * i;
    const int q = inst * 64 + lane;
    const int r = q / LM_CHUNKS_PER_ROW;
    int cc = q - r * LM_CHUNKS_PER_ROW;
    cc = cc > 39 ? 39 : cc;
    int s = g0r + r * LM_HOP + 4 * cc;
    s = s > hi_r - 4 ? hi_r - 4 : s;
    s = s < 0 ? 0 : s;
    if (q < LM_PIECES)
      __builtin_amdgcn_global_load_lds((glb_void*)((const char*)pcm + (unsigned)(s * 4)),
                                       (lds_void*)(lds + inst * 1024), 16, 0, 0);
  }
}

// The same transfer for a tile whose span, plus one row of slack, lies inside [0, hi_r): one
// instruction per LDS row, rows wave, wave + 4, ...  An instruction covers 1024 LDS bytes, a
// row 656, so lanes 41..63 land on pieces 0..22 of the following row and fetch exactly the
// samples that belong there (lane 40 fills the pad piece); whichever instruction writes a
// byte last writes the same value.  Source and destination are uniform per instruction, so
// issuing costs scalar work only.  The last instalments clamp to the last row, whose
// instruction keeps lanes 41..63 switched off (they would write past the rows).
__device__ __forceinline__ void issue_rows_dma(const RowDma& d, int wave, int ln, int i0, int i1) {
#pragma unroll
  for (int i = i0; i < i1; ++i) {
    int di = 4 * i;
    if (4 * i + 3 >= LM_ROWS) di = min(di, LM_ROWS - 1 - wave);
    const bool last = wave + di == LM_ROWS - 1;             // wave-uniform
    if (!last || ln < 41)
      __builtin_amdgcn_global_load_lds((glb_void*)(d.base + (long)di * (LM_HOP * 4) + d.lane_off),
                                       (lds_void*)(size_t)(d.lds0 + di * LM_ROW_BYTES), 16, 0, 0);
  }
}

// Reflect padding (np.pad mode='reflect', stft.py:251-255): the samples before
// the clip start and after its end, as far as a valid frame can reach.
__device__ __forceinline__ void patch_reflect(const float* __restrict__ pcm, long origin, long n_total,
                                               long hi_al, long g0, float* lds, int tid) {
  if (tid < 200) {
    const long g = -200 + tid;
    const long i = g - g0;
    if (i >= 0 && i < (long)LM_ROWS * LM_HOP) {
      const int r = (int)(i / LM_HOP), c = (int)(i - (long)r * LM_HOP);
      lds[r * LM_PITCH + c] = pcm[-g - origin];
    }
  }
  if (tid < 204) {
    const long g = hi_al + tid;
    const long i = g - g0;
    if (i >= 0 && i < (long)LM_ROWS * LM_HOP) {
      const int r = (int)(i / LM_HOP), c = (int)(i - (long)r * LM_HOP);
      long src = g < n_total ? g : 2 * (n_total - 1) - g;
      src = src < origin ? origin : src;
      lds[r * LM_PITCH + c] = pcm[src - origin];
    }
  }
}

template <int NMELS>
struct LmShape;
template <>
struct LmShape<80> {
  static constexpr int kSplit = LOGMEL_SPLIT_80, kAcc = LOGMEL_NACC_80;
  static __device__ __forceinline__ void project(float (&mg)[104], lm_const_ptr ct, bool upper, float (&acc)[kAcc]) {
    logmel_project_80(mg, ct, upper, acc);
  }
};
template <>
struct LmShape<128> {
  static constexpr int kSplit = LOGMEL_SPLIT_128, kAcc = LOGMEL_NACC_128;
  static __device__ __forceinline__ void project(float (&mg)[104], lm_const_ptr ct, bool upper, float (&acc)[kAcc]) {
    logmel_project_128(mg, ct, upper, acc);
  }
};

// -DLM_TIMING: per-phase cycle stamps, printed by one block (development builds only)
#ifdef LM_TIMING
#define LM_TICK(i) { __builtin_amdgcn_sched_barrier(0); const unsigned now_ = (unsigned)__builtin_amdgcn_s_memtime(); tacc[i] += now_ - tlast; tlast = now_; __builtin_amdgcn_sched_barrier(0); }
#else
#define LM_TICK(i)
#endif

template <int NMELS, bool BULK>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2)))
void logmel_pair(const LogmelArgs a) {
  // two objects, so that the compiler does not order the table reads of the second pass behind
  // the transfers that refill the rows (it would wait for every instalment to land)
  __shared__ __attribute__((aligned(1024))) unsigned char lds[LM_TABLE_OFF];
  __shared__ __attribute__((aligned(16))) unsigned char tabs[LM_LDS_TABLE_BYTES];
  constexpr int kSplit = LmShape<NMELS>::kSplit, kAcc = LmShape<NMELS>::kAcc;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool upper = lane >= 32;
  const int slot = 32 * wave + (lane & 31);             // frame of the tile this lane pair owns
  const lm_const_ptr ct = (lm_const_ptr)(unsigned long long)a.ct;

  int tile = blockIdx.x;
  if (tile >= a.n_tiles) return;
  const int hi_r = (int)(a.hi_al - a.origin);
  issue_tile_dma(a.pcm, (int)((a.frame_begin + (long)tile * LM_TILE_FRAMES) * LM_HOP - 200 - a.origin),
                  hi_r, lds, wave, lane);
  if (tid < LM_LDS_TABLE_BYTES / 16)
    *(v4f*)(tabs + 16 * tid) = *(const v4f*)(a.lds_table + 4 * tid);
  __builtin_amdgcn_s_waitcnt(LM_WAIT_VM0);
  __builtin_amdgcn_s_waitcnt(LM_WAIT_LGKM0);

  const int fslot = slot < LM_TILE_FRAMES ? slot : LM_TILE_FRAMES - 1;
  const unsigned char* base = lds + fslot * LM_ROW_BYTES + (upper ? 8 : 0);
  const unsigned char* wt = tabs + (upper ? LOGMEL_WIN_BYTES / 2 : 0);
  const unsigned char* tt = tabs + LOGMEL_WIN_BYTES + (upper ? LOGMEL_TW_BYTES / 2 : 0);

#ifdef LM_TIMING
  unsigned tacc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  unsigned tlast = (unsigned)__builtin_amdgcn_s_memtime();
#endif
  float run_max = -__builtin_inff();
  while (tile < a.n_tiles) {
    const long t0 = a.frame_begin + (long)tile * LM_TILE_FRAMES;
    const long g0 = t0 * LM_HOP - 200;
    __builtin_amdgcn_s_waitcnt(LM_WAIT_VM0);
    block_barrier();                        // every wave's DMA pieces of this tile have landed
    LM_TICK(0)
    if (g0 < 0 || g0 + (long)(LM_ROWS + 1) * LM_HOP > a.hi_al) {   // block-uniform: an end tile
      if (BULK) {
        issue_tile_dma(a.pcm, (int)(g0 - a.origin), hi_r, lds, wave, lane);
        __builtin_amdgcn_s_waitcnt(LM_WAIT_VM0);
        block_barrier();
      }
      patch_reflect(a.pcm, a.origin, a.n_total, a.hi_al, g0, (float*)lds, tid);
      __builtin_amdgcn_s_waitcnt(LM_WAIT_LGKM0);
      block_barrier();
    }

    v2f sa[LOGMEL_N_SA];
    LM_TICK(1)
    logmel_pass_a(base, wt, ct, sa);
    __builtin_amdgcn_s_waitcnt(LM_WAIT_LGKM0);
    LM_TICK(2)
    block_barrier();                        // the rows may now be refilled
    LM_TICK(3)
    const int next = tile + gridDim.x;
    int g0r_next = (int)((a.frame_begin + (long)next * LM_TILE_FRAMES) * LM_HOP - 200 - a.origin);
    if (!BULK && next < a.n_tiles) issue_tile_dma(a.pcm, g0r_next, hi_r, lds, wave, lane);
    g0r_next = min(max(g0r_next, 0), hi_r - (LM_ROWS + 1) * LM_HOP);
    float mg[104];
    // The next tile's rows are requested in four instalments between the families (scalar code
    // only, rebuilt each time so that nothing stays live across the register-starved sections).
    logmel_pass_b(ct, tt, sa, mg, [&](int f) __attribute__((always_inline)) {
      constexpr int kEvery = 13 / LM_DMA_PARTS;
      if (BULK && f % kEvery == 0 && f / kEvery < LM_DMA_PARTS) {
        const int part = f / kEvery;
        RowDma rd;
        rd.base = (const char*)a.pcm + ((long)g0r_next + wave * LM_HOP) * 4;
        rd.lds0 = (unsigned)(size_t)(lds_void*)lds + wave * LM_ROW_BYTES;
        const int ln = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0));
        rd.lane_off = ln * 16 - (ln >= 40 ? 16 : 0);
        issue_rows_dma(rd, wave, ln, (LM_ROW_DMA_PER_WAVE * part) / LM_DMA_PARTS,
                        (LM_ROW_DMA_PER_WAVE * (part + 1)) / LM_DMA_PARTS);
      }
    });
#ifdef LM_DEBUG       // development build only: keeps all of sa alive through the second pass
    if (a.dbg && tile == 0) {
      float* d = a.dbg + tid * 512;
#pragma unroll
      for (int i = 0; i < LOGMEL_N_SA; ++i) { d[2 * i] = sa[i].x; d[2 * i + 1] = sa[i].y; }
#pragma unroll
      for (int i = 0; i < 104; ++i) d[200 + i] = mg[i];
    }
#endif
    LM_TICK(4)
    float mel[kAcc];
    LmShape<NMELS>::project(mg, ct, upper, mel);
#ifdef LM_TIMING
    asm volatile("" :: "v"(mel[0]), "v"(mel[kAcc - 1]), "v"(mel[7]));
#endif
    LM_TICK(5)
#ifdef LM_DEBUG
    if (a.dbg && tile == 0) {
      float* d = a.dbg + tid * 512;
#pragma unroll
      for (int i = 0; i < kAcc; ++i) d[304 + i] = mel[i];
    }
#endif

    const long row = (long)tile * LM_TILE_FRAMES + slot;
    const bool valid = slot < LM_TILE_FRAMES && row < a.frame_count;
    // log10, scaling, then the store, sixteen columns at a time.  The lower lane of a pair holds
    // columns [0, kSplit) of its frame's row, the upper lane the rest.  Four neighbouring lanes
    // (four consecutive frames, same half) transpose 4 x 4 16-byte pieces so that a store
    // instruction writes 64 contiguous bytes per quad; a group's four stores are issued as soon
    // as it is transposed, so the stores are spread over the whole section instead of meeting
    // every other workgroup's stores at the memory side in one burst.
    // extremes of the row piece: log10 and the scaling are monotone in float arithmetic, so they are
    // taken on the projection sums and pushed through the same two expressions once
    float pmax = mel[0], pmin = mel[0];
#pragma unroll
    for (int m = 1; m < kAcc; ++m) {
      pmax = fmaxf(pmax, mel[m]);
      pmin = fminf(pmin, mel[m]);
    }
    float ymax = __builtin_amdgcn_logf(fmaxf(pmax, 1e-10f)) * 0.3010299956639812f;
    float zmin = (__builtin_amdgcn_logf(fmaxf(pmin, 1e-10f)) * 0.3010299956639812f + 4.0f) * 0.25f;
    {
      const bool hi1 = lane & 1, hi2 = lane & 2;
      const long qrow = row - (lane & 3);
      const int qslot = slot & ~3;
      const int my_cols = upper ? NMELS - kSplit : kSplit;
      float* dst0 = a.out + qrow * NMELS + (upper ? kSplit : 0) + 4 * (lane & 3);
#pragma unroll
      for (int g = 0; g < kAcc; g += 16) {
#pragma unroll
        for (int m = g; m < g + 16; ++m) {
          // log10 = log2 x log10(2), as __log10f computes it; its denormal pre-scaling is left out
          // (the argument is at least 1e-10)
          const float y = __builtin_amdgcn_logf(fmaxf(mel[m], 1e-10f)) * 0.3010299956639812f;
          mel[m] = (y + 4.0f) * 0.25f;
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          float* r0 = &mel[g + c];
#pragma unroll
          for (int j = 0; j < 4; j += 2) {
            const float send = hi1 ? r0[4 * j] : r0[4 * (j + 1)];
            const float recv = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(
                __builtin_bit_cast(int, send), 0xB1, 0xF, 0xF, true));
            if (hi1) r0[4 * j] = recv; else r0[4 * (j + 1)] = recv;
          }
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            const float send = hi2 ? r0[4 * j] : r0[4 * (j + 2)];
            const float recv = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(
                __builtin_bit_cast(int, send), 0x4E, 0xF, 0xF, true));
            if (hi2) r0[4 * j] = recv; else r0[4 * (j + 2)] = recv;
          }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (g < my_cols && qslot + j < LM_TILE_FRAMES && qrow + j < a.frame_count)
            *(v4f*)(dst0 + (long)j * NMELS + g) =
                v4f{mel[g + 4 * j], mel[g + 4 * j + 1], mel[g + 4 * j + 2], mel[g + 4 * j + 3]};
      }
    }
    if (!valid) {
      ymax = -__builtin_inff();
      zmin = __builtin_inff();
    }
    LM_TICK(6)
    run_max = fmaxf(run_max, ymax);
    zmin = wave_min(zmin);
    if (lane == 0) a.seg_min[tile * 4 + wave] = zmin;
    tile = next;
    LM_TICK(7)
  }
#ifdef LM_TIMING
  if (blockIdx.x == 7 && lane == 0)
    printf("wave %d: top wait+barrier %u  edge %u  passA %u  barrier %u  families %u  project %u  log %u  dma issue %u  transposes %u  stores %u\n",
           wave, tacc[0], tacc[1], tacc[2], tacc[3], tacc[4], tacc[5], tacc[6], tacc[8], tacc[9], tacc[7]);
#endif
  run_max = wave_max(run_max);
  block_barrier();
  float* red = (float*)lds;
  if (lane == 0) red[wave] = run_max;
  __syncthreads();
  if (tid == 0) {
    const float m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __hip_atomic_store(a.blk_max + blockIdx.x, m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    const unsigned t = __hip_atomic_fetch_add(a.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    red[4] = t == gridDim.x - 1 ? 1.0f : 0.0f;
  }
  __syncthreads();
  if (red[4] != 0.0f && wave == 0) {
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    float m = -__builtin_inff();
    for (int i = lane; i < (int)gridDim.x; i += 64)
      m = fmaxf(m, __hip_atomic_load(a.blk_max + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    m = wave_max(m);
    if (lane == 0) {
      *a.piece_max = m;
      __hip_atomic_store(a.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// Segment s = (tile, wave) covers rows [tile * 118 + 32 * wave, + 32) of that tile.
__global__ __launch_bounds__(256)
void logmel_finalize(float* __restrict__ out, long n_frames, int n_mels, int n_segments,
                      const float* __restrict__ gmax, const float* __restrict__ seg_min) {
  const float zc = ((*gmax - 8.0f) + 4.0f) * 0.25f;
  for (int s = blockIdx.x; s < n_segments; s += gridDim.x) {
    if (!(seg_min[s] < zc)) continue;
    const int tile = s >> 2, wave = s & 3;
    long r0 = (long)tile * LM_TILE_FRAMES + 32 * wave;
    long r1 = r0 + 32;
    const long tile_end = (long)(tile + 1) * LM_TILE_FRAMES;
    r1 = r1 > tile_end ? tile_end : r1;
    r1 = r1 > n_frames ? n_frames : r1;
    const long e0 = r0 * n_mels, e1 = r1 * n_mels;
    for (long e = e0 + 4 * threadIdx.x; e < e1; e += 4 * 256) {
      v4f v = *(v4f*)(out + e);
      v.x = fmaxf(v.x, zc); v.y = fmaxf(v.y, zc); v.z = fmaxf(v.z, zc); v.w = fmaxf(v.w, zc);
      *(v4f*)(out + e) = v;
    }
  }
}

__global__ void logmel_init(float* piece_max) { *piece_max = -__builtin_inff(); }

// ------------------------------------------------------------------ host side
#include "../../include/stt_mi355x.h"
#include "logmel_tables.h"
#include "stt_common.h"
#include <vector>

struct stt_logmel_plan {
  int n_mels;
  int device;
  int n_cu;
  float* d_table;        // DFT constants + mel weights (scalar loads), then the LDS tables
  float* d_lds_table;    // window / twiddle tables every workgroup copies into LDS (inside d_table)
  unsigned* d_sync;      // [0] ticket counter, [64..] one float per block; zeroed at creation
  void* d_static_ws;     // this device's copy of lm_static_ws (address looked up at creation)
};

// Workspace of stt_logmel_run_static: the same bytes stt_logmel_run takes as `d_workspace`
// (piece maximum, per-segment minima, padded copy of a short clip), held in the code object so
// that the call needs no allocation.  Every GPU the library runs on holds its own copy at its
// own address; host code reaches it only through the pointer hipGetSymbolAddress returned for
// the current device (the array's name is a host-side stand-in, not a device address).
#define LM_STATIC_WS_BYTES 131072
static __device__ __align__(256) unsigned char lm_static_ws[LM_STATIC_WS_BYTES];

static const size_t kWsHeader = 256;      // piece maximum lives in the first float
static const size_t kWsPadPcm = 512 * sizeof(float);

static int tiles_for(int64_t frames) {
  return (int)((frames + LM_TILE_FRAMES - 1) / LM_TILE_FRAMES);
}

static float* g_lm_debug = nullptr;

extern "C" {

void stt_logmel_set_debug(float* d_buffer) { g_lm_debug = d_buffer; }    // development aid, not in the header

int64_t stt_logmel_num_frames(int64_t n_samples) {
  if (n_samples < LM_NFFT) n_samples = LM_NFFT;
  return n_samples / LM_HOP;
}

size_t stt_logmel_workspace_bytes(int64_t n_samples) {
  const int64_t frames = stt_logmel_num_frames(n_samples);
  size_t seg = (size_t)tiles_for(frames) * 4 * sizeof(float);
  seg = (seg + 255) & ~(size_t)255;
  return kWsHeader + seg + kWsPadPcm;
}

int stt_logmel_plan_create(int n_mels, const float* window, const float* mel_basis,
                           stt_logmel_plan** plan) {
  if (!window || !mel_basis || !plan) return stt_fail(STT_ERR_INVALID_ARGUMENT, "null argument");
  if (n_mels != 80 && n_mels != 128)
    return stt_fail(STT_ERR_UNSUPPORTED, "n_mels must be 80 or 128 (got %d)", n_mels);
  const int nnz = n_mels == 80 ? LOGMEL_NNZ_80 : LOGMEL_NNZ_128;
  const unsigned short (*nz)[2] = n_mels == 80 ? logmel_nz_80 : logmel_nz_128;
  const int off_mel = n_mels == 80 ? LOGMEL_OFF_MEL80 : LOGMEL_OFF_MEL128;

  // every non-zero of the basis must have a slot in the generated projection
  std::vector<char> covered((size_t)201 * n_mels, 0);
  for (int i = 0; i < nnz; ++i) covered[(size_t)nz[i][0] * n_mels + nz[i][1]] = 1;
  for (int k = 0; k < 201; ++k)
    for (int m = 0; m < n_mels; ++m)
      if (mel_basis[(size_t)k * n_mels + m] != 0.0f && !covered[(size_t)k * n_mels + m])
        return stt_fail(STT_ERR_UNSUPPORTED,
                        "mel basis has a non-zero at (bin %d, filter %d) outside the Slaney "
                        "16 kHz / 400-point / 0-8000 Hz structure this kernel is built for", k, m);

  // scalar-load table, rounded up so that the LDS tables behind it stay 16-byte aligned
  const size_t n_scalar = (LOGMEL_TABLE_LEN + 3) & ~(size_t)3;
  std::vector<float> table(n_scalar + LM_LDS_TABLE_BYTES / 4, 0.0f);
  for (int i = 0; i < LOGMEL_N_FFT_CONSTS; ++i) table[LOGMEL_OFF_FFT + i] = logmel_fft_consts[i];
  for (int i = 0; i < nnz; ++i) table[off_mel + i] = mel_basis[(size_t)nz[i][0] * n_mels + nz[i][1]];
  // window: per lane half and sub-DFT j, the pairs (w[4m + 2 half], w[4m + 2 half + 1]) of the
  // inputs m = j + 4 m2 in the order the sub-DFT consumes them (26 slots, the last one padding)
  float* lds_table = table.data() + n_scalar;
  for (int half = 0; half < 2; ++half)
    for (int j = 0; j < 4; ++j)
      for (int pos = 0; pos < 25; ++pos) {
        const int m = j + 4 * logmel_win_pos[j][pos], at = 2 * (half * 104 + 26 * j + pos);
        lds_table[at] = window[4 * m + 2 * half];
        lds_table[at + 1] = window[4 * m + 2 * half + 1];
      }
  for (int i = 0; i < LOGMEL_TW_BYTES / 4; ++i) lds_table[LOGMEL_WIN_BYTES / 4 + i] = logmel_twiddles[i];

  stt_logmel_plan* p = new stt_logmel_plan();
  p->n_mels = n_mels;
  STT_HIP_OR_FAIL(hipGetDevice(&p->device), delete p);
  hipDeviceProp_t prop;
  STT_HIP_OR_FAIL(hipGetDeviceProperties(&prop, p->device), delete p);
  p->n_cu = prop.multiProcessorCount;
  STT_HIP_OR_FAIL(hipMalloc((void**)&p->d_table, table.size() * sizeof(float)), delete p);
  p->d_lds_table = p->d_table + n_scalar;
  STT_HIP_OR_FAIL(hipMemcpy(p->d_table, table.data(), table.size() * sizeof(float),
                            hipMemcpyHostToDevice), ((void)hipFree(p->d_table), delete p));
  STT_HIP_OR_FAIL(hipMalloc((void**)&p->d_sync, (64 + 1024) * 4), ((void)hipFree(p->d_table), delete p));
  STT_HIP_OR_FAIL(hipMemset(p->d_sync, 0, (64 + 1024) * 4),
                  ((void)hipFree(p->d_sync), (void)hipFree(p->d_table), delete p));
  // the plan belongs to the current device: look up that device's copy of the static workspace
  STT_HIP_OR_FAIL(hipGetSymbolAddress(&p->d_static_ws, HIP_SYMBOL(lm_static_ws)),
                  ((void)hipFree(p->d_sync), (void)hipFree(p->d_table), delete p));
  *plan = p;
  return STT_OK;
}

void stt_logmel_plan_destroy(stt_logmel_plan* plan) {
  if (!plan) return;
  (void)hipFree(plan->d_table);
  (void)hipFree(plan->d_sync);
  delete plan;
}

int stt_logmel_stage1(const stt_logmel_plan* plan, const float* d_pcm, int64_t sample_origin,
                      int64_t n_local, int64_t n_total, int64_t frame_begin, int64_t frame_count,
                      float* d_mel, float* d_piece_max, void* d_workspace, size_t workspace_bytes,
                      void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!plan || !d_piece_max) return stt_fail(STT_ERR_INVALID_ARGUMENT, "null argument");
  if (frame_count == 0) {      // a piece without frames (more GPUs than tiles): only its maximum, -inf
    hipLaunchKernelGGL(logmel_init, dim3(1), dim3(1), 0, stream, d_piece_max);
    return STT_OK;
  }
  if (!d_pcm || !d_mel || !d_workspace) return stt_fail(STT_ERR_INVALID_ARGUMENT, "null argument");
  if (n_total < LM_NFFT)
    return stt_fail(STT_ERR_INVALID_ARGUMENT, "clip shorter than 400 samples (%lld): pad it first",
                    (long long)n_total);
  if (frame_begin < 0 || frame_count < 0 || frame_begin + frame_count > n_total / LM_HOP)
    return stt_fail(STT_ERR_INVALID_ARGUMENT, "frame range [%lld, +%lld) outside the clip's %lld frames",
                    (long long)frame_begin, (long long)frame_count, (long long)(n_total / LM_HOP));
  if (((uintptr_t)d_pcm & 15) || ((uintptr_t)d_mel & 15))
    return stt_fail(STT_ERR_INVALID_ARGUMENT, "d_pcm and d_mel must be 16-byte aligned");
  // samples the frames touch inside the clip must be readable
  long need_lo = (long)frame_begin * LM_HOP - 200, need_hi = (long)(frame_begin + frame_count - 1) * LM_HOP + 200;
  if (need_lo < 0) need_lo = 0;
  if (need_hi > n_total) need_hi = n_total;
  // reflected samples come from [1, 200] and [n_total - 201, n_total - 2]
  if (frame_begin * LM_HOP - 200 < 0 && need_hi < 201) need_hi = 201;
  const long lo = sample_origin, hi = sample_origin + n_local < n_total ? sample_origin + n_local : n_total;
  if (sample_origin & 3)
    return stt_fail(STT_ERR_INVALID_ARGUMENT, "sample_origin must be a multiple of 4");
  if (n_local > (int64_t)2000000000)
    return stt_fail(STT_ERR_INVALID_ARGUMENT, "pcm piece of %lld samples: at most 2e9 per call",
                    (long long)n_local);
  if (sample_origin < 0 || lo > need_lo || hi < need_hi || hi - lo < 8)
    return stt_fail(STT_ERR_INVALID_ARGUMENT,
                    "pcm piece [%ld, %ld) does not cover the samples [%ld, %ld) its frames read", lo, hi,
                    need_lo, need_hi);
  const int n_tiles = tiles_for(frame_count);
  const size_t seg_bytes = (size_t)n_tiles * 4 * sizeof(float);
  if (workspace_bytes < kWsHeader + seg_bytes)
    return stt_fail(STT_ERR_WORKSPACE, "workspace too small: %zu < %zu", workspace_bytes,
                    kWsHeader + seg_bytes);

  LogmelArgs a;
  a.pcm = d_pcm; a.origin = sample_origin; a.lo = lo; a.hi_al = lo + ((hi - lo) & ~3L);
  a.n_total = n_total;
  a.frame_begin = frame_begin; a.frame_count = frame_count; a.n_tiles = n_tiles;
  a.ct = plan->d_table; a.lds_table = plan->d_lds_table; a.out = d_mel; a.piece_max = d_piece_max;
  a.seg_min = (float*)((char*)d_workspace + kWsHeader);
  a.ticket = plan->d_sync; a.blk_max = (float*)(plan->d_sync + 64);
  a.dbg = g_lm_debug;
  int grid = n_tiles < 2 * plan->n_cu ? n_tiles : 2 * plan->n_cu;      // two workgroups per CU
  grid = grid < 1024 ? grid : 1024;                                    // one blk_max slot per block
  const bool bulk = a.hi_al - a.origin >= (long)(LM_ROWS + 1) * LM_HOP;
  if (plan->n_mels == 80) {
    if (bulk) hipLaunchKernelGGL((logmel_pair<80, true>), dim3(grid), dim3(256), 0, stream, a);
    else hipLaunchKernelGGL((logmel_pair<80, false>), dim3(grid), dim3(256), 0, stream, a);
  } else {
    if (bulk) hipLaunchKernelGGL((logmel_pair<128, true>), dim3(grid), dim3(256), 0, stream, a);
    else hipLaunchKernelGGL((logmel_pair<128, false>), dim3(grid), dim3(256), 0, stream, a);
  }
  STT_HIP_OR_FAIL(hipGetLastError(), (void)0);
  return STT_OK;
}

int stt_logmel_stage2(const stt_logmel_plan* plan, float* d_mel, int64_t frame_count,
                      const float* d_clip_max, const void* d_workspace, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (frame_count <= 0) return STT_OK;
  if (!plan || !d_mel || !d_clip_max || !d_workspace)
    return stt_fail(STT_ERR_INVALID_ARGUMENT, "null argument");
  const int n_seg = tiles_for(frame_count) * 4;
  const int grid = n_seg < 2048 ? n_seg : 2048;
  hipLaunchKernelGGL(logmel_finalize, dim3(grid), dim3(256), 0, stream, d_mel, (long)frame_count,
                     plan->n_mels, n_seg, d_clip_max,
                     (const float*)((const char*)d_workspace + kWsHeader));
  STT_HIP_OR_FAIL(hipGetLastError(), (void)0);
  return STT_OK;
}

int stt_logmel_run(const stt_logmel_plan* plan, const float* d_pcm, int64_t n_samples,
                   float* d_mel, void* d_workspace, size_t workspace_bytes, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!plan || !d_pcm || !d_mel || !d_workspace)
    return stt_fail(STT_ERR_INVALID_ARGUMENT, "null argument");
  if (n_samples <= 0) return stt_fail(STT_ERR_INVALID_ARGUMENT, "empty clip");
  if (workspace_bytes < stt_logmel_workspace_bytes(n_samples))
    return stt_fail(STT_ERR_WORKSPACE, "workspace too small: %zu < %zu", workspace_bytes,
                    stt_logmel_workspace_bytes(n_samples));
  const float* pcm = d_pcm;
  int64_t n = n_samples;
  if (n < LM_NFFT) {   // MelSTFT.__call__ zero-pads short clips to one window (stft.py:111-112)
    float* pad = (float*)((char*)d_workspace + workspace_bytes - kWsPadPcm);
    STT_HIP_OR_FAIL(hipMemsetAsync(pad, 0, kWsPadPcm, stream), (void)0);
    STT_HIP_OR_FAIL(hipMemcpyAsync(pad, d_pcm, (size_t)n * sizeof(float), hipMemcpyDeviceToDevice,
                                   stream), (void)0);
    pcm = pad;
    n = LM_NFFT;
  }
  float* piece_max = (float*)d_workspace;
  const int64_t frames = n / LM_HOP;
  int rc = stt_logmel_stage1(plan, pcm, 0, n, n, 0, frames, d_mel, piece_max, d_workspace,
                             workspace_bytes, stream);
  if (rc != STT_OK) return rc;
  return stt_logmel_stage2(plan, d_mel, frames, piece_max, d_workspace, stream);
}

int64_t stt_logmel_static_max_samples(void) {
  // largest n with stt_logmel_workspace_bytes(n) <= LM_STATIC_WS_BYTES
  const int64_t tiles = (int64_t)(LM_STATIC_WS_BYTES - kWsHeader - kWsPadPcm) / (4 * sizeof(float)) / 64 * 64;
  return tiles * LM_TILE_FRAMES * LM_HOP;
}

int stt_logmel_run_static(const stt_logmel_plan* plan, const float* d_pcm, int64_t n_samples,
                          float* d_mel, void* stream) {
  if (!plan || !d_pcm || !d_mel) return stt_fail(STT_ERR_INVALID_ARGUMENT, "null argument");
  if (n_samples <= 0) return stt_fail(STT_ERR_INVALID_ARGUMENT, "empty clip");
  if (n_samples > stt_logmel_static_max_samples())
    return stt_fail(STT_ERR_WORKSPACE,
                    "clip of %lld samples exceeds the %lld samples the static workspace (%d bytes) "
                    "serves: use stt_logmel_run with a caller workspace",
                    (long long)n_samples, (long long)stt_logmel_static_max_samples(), LM_STATIC_WS_BYTES);
  int dev = -1;
  STT_HIP_OR_FAIL(hipGetDevice(&dev), (void)0);
  if (dev != plan->device)
    return stt_fail(STT_ERR_INVALID_ARGUMENT, "plan was created on device %d, current device is %d",
                    plan->device, dev);
  // Nothing in the workspace has to start at zero: the piece maximum and every segment minimum
  // the second kernel reads are written by the first kernel of the same call, and the padded
  // copy of a short clip is cleared by the call that uses it.
  return stt_logmel_run(plan, d_pcm, n_samples, d_mel, plan->d_static_ws,
                        stt_logmel_workspace_bytes(n_samples), stream);
}

}  // extern "C"
