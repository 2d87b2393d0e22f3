// Device kernels of the MI355X Whisper engine: launch wrappers shared between the .hip files.
// Conventions: activations that feed a GEMM are bf16; the residual stream is fp32; weights are
// bf16 [N][K] (output feature major, K contiguous); every wrapper only enqueues on `stream`.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef uint16_t bf16_t;

// ---------------------------------------------------------------- GEMM epilogues
enum GemmEpilogue {
  EPI_BF16 = 0,        // out_bf16[m][n] = acc + bias[n]
  EPI_GELU_BF16 = 1,   // out_bf16[m][n] = gelu(acc + bias[n])
  EPI_RESID_F32 = 2,   // out_f32[m][n] += acc + bias[n]
  EPI_GELU_POS_F32 = 3,// out_f32[m][n] = gelu(acc + bias[n]) + pos[m][n]     (conv2 + sinusoids)
  EPI_F32 = 4,         // out_f32[m][n] = acc (+ bias if given)
  EPI_KV_HEADS = 5,    // bf16, head-major: row m = (b, t) with t < kv_t, column n < N/2 -> out[b][h][t][c],
                       // n >= N/2 -> out2[b][h][t][c] with h = (n mod N/2) / 64, c = n mod 64 (cross K | V)
  EPI_QKV_ENC = 6,     // encoder q | k | v projection (N = 3 d): q columns scaled by q_scale before the
                       // rounding to bf16, q and k to out[m][n]; v columns (n >= 2 d) transposed into
                       // out2[n - 2 d][vt_pitch] at position m
};

struct GemmArgs {
  const bf16_t* a;     // [batch][M rows, row stride lda elements]
  long lda;
  long a_batch_stride; // elements
  const bf16_t* w;     // [N][K]
  const float* bias;   // [N] or null
  void* out;           // bf16 or f32, row stride ldc elements
  long ldc;
  long out_batch_stride;
  const float* pos;    // [M][ldc-compatible: row stride pos_ld] for EPI_GELU_POS_F32
  long pos_ld;
  int m, n, k;         // k % 64 == 0
  int batch;
  int gelu_tanh;       // 1 = tanh approximation (Keras default), 0 = exact erf
  void* out2;          // EPI_KV_HEADS: the V image
  int kv_t;            // EPI_KV_HEADS: rows per batch item (1500)
  int k_split;         // > 1 (256 x 128 kernel only): blockIdx.y takes K slice y of k_split equal slices
  long split_stride;   // ... and writes to out + y * split_stride elements (raw partial sums: EPI_F32, no bias)
  float q_scale;       // EPI_QKV_ENC
  long vt_pitch;       // EPI_QKV_ENC: row pitch of the V^T image in elements (multiple of 64)
};

// Tiled MFMA GEMM for M >= a few hundred rows (encoder, prompt pass).
void launch_gemm_big(const GemmArgs& g, int epilogue, hipStream_t stream);

// Weight-streaming GEMM for M <= 640 rows (decode steps).  split_k > 1 writes raw fp32 partial
// sums to slabs[split_k][M][N] (no bias / activation); split_k == 1 applies the epilogue
// (EPI_BF16, EPI_GELU_BF16 or EPI_F32).
void launch_gemm_skinny(const GemmArgs& g, int epilogue, int split_k, float* slabs,
                        hipStream_t stream);
int pick_split_k(int m, int n, int k, int n_cu);

// The LayerNorm that produces a decode-step GEMM's input, done by the GEMM itself (rows <= 4):
// every block repeats, for its own use, x_new = x_in + sum of pending slabs + add_bias and
// h = LN(x_new) * gamma + beta (the arithmetic of launch_add_layernorm, operation for operation,
// so the result does not depend on which of the two paths ran); block (0, 0) also stores x_new to
// x_out, which must not be x_in (other blocks are still reading it).  With n_slabs == 0 nothing
// is added and nothing is stored.  Input row of output row r: r * row_mul + row_off.
struct LnFuse {
  const float* x_in;
  float* x_out;
  const float* slabs;
  int n_slabs;
  long slab_stride;
  const float* add_bias;
  const float* gamma;
  const float* beta;
  float eps;
  int row_mul, row_off;
};
// g.a is ignored; g.m <= 4, g.k == the width d of x (<= 2048, a multiple of 64).
void launch_gemm_skinny_ln(const GemmArgs& g, const LnFuse& f, int epilogue, int split_k, float* slabs_out,
                           hipStream_t stream);

// ---------------------------------------------------------------- normalisation / elementwise
// x[m][d] += sum_s slabs[s][m][d] + add_bias[d] (both optional), then out = LN(x) * gamma + beta.
// Output row r is computed from input row r * row_mul + row_off (row_mul = 1, row_off = 0 for a
// plain pass; used to normalise only the last position of every sequence).
void launch_add_layernorm(float* x, const float* slabs, int n_slabs, long slab_stride,
                          const float* add_bias, const float* gamma, const float* beta,
                          bf16_t* out_bf16, float* out_f32, int m, int d, float eps,
                          int row_mul, int row_off, hipStream_t stream);
// mel fp32 [B][frames<=3000][n_mels] -> bf16 [B][3002][n_mels] with one zero row before and after
// and rows >= frames zero (pad_or_trim with 0.0).
void launch_pack_mel(const float* mel, int batch, int frames, int n_mels, bf16_t* out,
                     hipStream_t stream);
// x[row][d] = E[token[row]][d] + P[pos[row]][d]
void launch_embed(const int* tokens, const int* positions, const bf16_t* emb, const float* pos_emb,
                  float* x, int rows, int d, int n_vocab, int n_ctx, hipStream_t stream);

// ---------------------------------------------------------------- attention
// Encoder self-attention, no mask.  q,k: [b][t][h][64] with row stride ld, q already multiplied
// by log2(e) / 8 (EPI_QKV_ENC); vt: [h * 64][vt_pitch] with window b's positions at b * t ..,
// vt_pitch >= batch * t + 32 and every element finite; out bf16 [b][t][h*64].  lift_at: how far (in
// base-2 units) a tile may raise a query's maximum above its softmax reference before the
// reference is moved; 8 in production, anything in (0, 100] gives the same result up to rounding.
void launch_attn_encoder(const bf16_t* q, const bf16_t* k, long ld, long batch_stride,
                         const bf16_t* vt, long vt_pitch, bf16_t* out, int batch, int t, int heads,
                         float lift_at, hipStream_t stream);

struct SelfAttnArgs {
  const bf16_t* qkv;   // [rows][3*d]: q | k | v of the new tokens; rows = n_seq * s_new
  bf16_t* k_cache;     // [n_seq][h][max_len][64]   (this layer)
  bf16_t* v_cache;
  const int* slot_of;  // [n_seq][max_len]: cache sequence that holds position p of sequence i
  const int* base_len; // [n_seq]: positions already in the cache before this call
  bf16_t* out;         // [rows][d]
  int n_seq, s_new, heads, max_len;
  const int* finished; // null, or [n_seq]: a sequence whose flag is set is skipped (nothing appended, its
                       // output rows left as they are): its next token is the pad token whatever the
                       // decoder computes (generation_utils.py:234-236)
  float* probs;        // null, or [n_seq][heads][s_new][probs_len] fp32: the attention weights of this pass
  int probs_len;       // (zero past a token's own position), what the reference returns as 'attn_<layer>'
                       // (architectures/transformers/transformer_arch.py:697-703)
};
// Appends the new k,v at positions base_len[i] .. +s_new-1 of sequence i (writing into slot i)
// and attends causally over positions 0 .. base_len[i] + j for new token j.
void launch_attn_decoder_self(const SelfAttnArgs& a, hipStream_t stream);

struct CrossAttnArgs {
  const bf16_t* q;     // [n_seq * s_new][d]; ignored when q_slabs is set
  const float* q_slabs;   // optional: the query projection as split-K partial sums [n][rows][d] ...
  int n_q_slabs;
  long q_slab_stride;
  const float* q_bias;    // ... plus its bias [d]: q = sum of slabs + bias, never rounded to bf16
  const bf16_t* k;     // [n_win][h][1500][64]   (this layer)
  const bf16_t* v;
  bf16_t* out;         // [n_seq * s_new][d]
  int n_win, seq_per_win, s_new, heads, t_enc;
  const int* finished; // null, or [n_win * seq_per_win]: a window all of whose sequences are finished is
                       // skipped (its K/V are not read, its output rows left as they are)
  float* probs;        // null, or [n_seq][heads][s_new][t_enc] fp32: the attention weights ('enc_attn_<layer>')
};
void launch_attn_decoder_cross(const CrossAttnArgs& a, hipStream_t stream);

// ---------------------------------------------------------------- generation
struct GenState {          // all device pointers; n_seq = batch * beams
  float* logits;           // [n_seq][V] fp32 (last position of each sequence)
  int* generated;          // [n_seq][max_steps]
  int* lengths;            // [n_seq]   (prompt + generated, EOS not counted)
  float* scores;           // [n_seq]
  int* finished;           // [n_seq]
  int* next_token;         // [n_seq]
  int* base_len;           // [n_seq]   cache length
  int* slot_of;            // [n_seq][max_len]
  int* step;               // [1]
  int* all_finished;       // [1]
  float* logits_out;       // null, or [n_seq][max_steps][V]: the processed scores of every step (greedy)
  const int* step_cap;     // null, or [batch]: new tokens window b may produce; it counts as finished after them
  // beam search scratch
  float* cand_val;         // [batch][beams]
  int* cand_idx;           // [batch][beams]
  float* row_val;          // [n_seq][16]: the `beams` best entries of each sequence's row ...
  int* row_idx;            // ... and their token ids (beam search, before the per-window merge)
  int* generated_tmp; int* lengths_tmp; int* finished_tmp; int* slot_of_tmp; float* scores_tmp;
};

struct FilterArgs {        // device pointers
  const int* suppress;     // tokens set to -inf at every step (or at steps > 0 with timestamps)
  int n_suppress;
  const int* suppress_first; // timestamp filter: tokens removed at the first step
  int n_suppress_first;
  int timestamps;          // 1 = apply the Whisper timestamp rules
  int timestamp_begin, eos, max_initial_timestamp;
};

struct LogitParams {
  float temperature;       // <= 0: unused
  float length_temperature;// 0: unused
  int has_length_temperature;
  int sample;              // 1: draw the next token(s) from the categorical distribution
  unsigned long long seed; // of the counter-based random numbers of the draw
};

// Per sequence: filter, log_softmax; greedy: argmax + state update.
// s_new = tokens the decoder consumed in this step (prompt length at the first step, then 1).
void launch_greedy_step(const GenState& s, const FilterArgs& f, const LogitParams& p, int n_seq,
                        int n_vocab, int max_steps, int s_new, int eos, int pad,
                        hipStream_t stream);
// tokens_in[i] = next_token[i], positions[i] = base_len[i]
void launch_prepare_step(const GenState& s, int* tokens_in, int* positions, int n_seq,
                         hipStream_t stream);
// Beam: per sequence log-probs (in place in s.logits, plus running score / finished handling),
// then per batch element top-`beams` over beams*V and the reorder of the per-sequence state.
void launch_beam_step(const GenState& s, const FilterArgs& f, const LogitParams& p, int batch,
                      int beams, int num_sentences, int n_vocab, int max_steps, int max_len,
                      int s_new, int eos, int pad, float length_power, int has_length_power,
                      int early_stopping, hipStream_t stream);
