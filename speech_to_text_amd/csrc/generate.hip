// Token selection kernels: logits processing, greedy argmax, beam top-k and state reorder.
//
// Restates architectures/generation_utils.py:195-278 (greedy body), :333-438 (beam body),
// :484-548 (process_logits, select_next_token) and the opt-in Whisper logits filter
// models/stt/whisper.py:354-416, fused so that the [n_seq, V] logits are read, not rewritten,
// wherever possible.  All per-step scalars (step counter, lengths, finished flags) live in
// device memory so that one captured graph can be replayed for every token.
#include "kernels.h"

#define NEG_INF (-__builtin_inff())

struct RowMask {           // per sequence: scores outside [lo, hi) are -inf
  int lo, hi;
};

__device__ __forceinline__ float block_max(float v, float* sh) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  return fmaxf(fmaxf(sh[0], sh[1]), fmaxf(sh[2], sh[3]));
}
__device__ __forceinline__ float block_sum(float v, float* sh) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  return (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

// f(j, row[j]) for j = lo + tid, lo + tid + 256, ... < hi with eight loads in flight per thread: a
// row is a few hundred KB read by one block, so a loop with one load per iteration is bound by
// the latency of that load.
template <class F>
__device__ __forceinline__ void for_row(const float* row, int lo, int hi, F f) {
  int j = lo + threadIdx.x;
  for (; j + 7 * 256 < hi; j += 8 * 256) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = row[j + u * 256];
#pragma unroll
    for (int u = 0; u < 8; ++u) f(j + u * 256, v[u]);
  }
  for (; j < hi; j += 256) f(j, row[j]);
}

// Scales the row in place (temperature, length temperature), writes -inf at suppressed tokens
// and returns the range mask of the timestamp rules.  Every thread of the block calls it.
__device__ RowMask prepare_row(float* __restrict__ row, int n_vocab, const FilterArgs& f,
                               const LogitParams& p, int length, const int* __restrict__ gen_row,
                               int t, float* sh) {
  const int tid = threadIdx.x;
  float mul = 1.0f;
  if (p.temperature > 0.f) mul = 1.0f / p.temperature;
  float lmul = 1.0f;
  if (p.has_length_temperature) lmul = powf((float)(length + 1), p.length_temperature);
  if (p.temperature > 0.f || p.has_length_temperature) {
    for (int i = tid; i < n_vocab; i += 256) {
      float v = row[i];
      if (p.temperature > 0.f) v = v / p.temperature;
      if (p.has_length_temperature) v = v * lmul;
      row[i] = v;
    }
  }
  (void)mul;
  RowMask m{0, n_vocab};
  if (!f.timestamps) {
    for (int i = tid; i < f.n_suppress; i += 256) {
      const int tok = f.suppress[i];
      if (tok >= 0 && tok < n_vocab) row[tok] = NEG_INF;
    }
    __syncthreads();
    return m;
  }
  if (t == 0) {          // only the first timestamp may start a window (whisper.py:355-367)
    for (int i = tid; i < f.n_suppress_first; i += 256) {
      const int tok = f.suppress_first[i];
      if (tok >= 0 && tok < n_vocab) row[tok] = NEG_INF;
    }
    m.lo = f.timestamp_begin;
    if (f.max_initial_timestamp > 0) m.hi = min(n_vocab, f.timestamp_begin + f.max_initial_timestamp);
    __syncthreads();
    return m;
  }
  for (int i = tid; i < f.n_suppress; i += 256) {
    const int tok = f.suppress[i];
    if (tok >= 0 && tok < n_vocab) row[tok] = NEG_INF;
  }
  const bool last_ts = gen_row[t - 1] >= f.timestamp_begin;
  const bool pen_ts = t < 2 ? true : gen_row[t - 2] >= f.timestamp_begin;
  if (last_ts && !pen_ts) m.lo = f.eos;                 // must close the pair (or stop)
  if (last_ts && pen_ts) m.hi = f.timestamp_begin;      // no third timestamp in a row
  __syncthreads();
  // if the timestamps together outweigh every single text token, sample a timestamp:
  // logsumexp(logp[ts:]) > max(logp[:ts]); the log-softmax constant cancels.
  float mt = NEG_INF, ma = NEG_INF;
  for_row(row, m.lo, m.hi, [&](int i, float v) {
    if (i < f.timestamp_begin) mt = fmaxf(mt, v); else ma = fmaxf(ma, v);
  });
  mt = block_max(mt, sh);
  ma = block_max(ma, sh);
  float se = 0.f;
  if (ma > NEG_INF)
    for_row(row, max(f.timestamp_begin, m.lo), m.hi, [&](int, float v) { se += __expf(v - ma); });
  se = block_sum(se, sh);
  const float ts_lse = ma > NEG_INF ? ma + __logf(se) : NEG_INF;
  if (ts_lse > mt) m.lo = max(m.lo, f.timestamp_begin);
  return m;
}

// Random numbers of the categorical draw (`keras.random.categorical`, generation_utils.py:546),
// counter based so that a recorded step can be replayed: element `(stream << 32) | j` of the
// SplitMix64 sequence started at `seed`, its top 23 bits as u = (k + 1/2) / 2^23 in (0, 1), and
// the standard Gumbel variate -log(-log u).  argmax_j (x_j + g_j) is a draw from softmax(x).
__device__ __forceinline__ float gumbel_noise(unsigned long long seed, unsigned stream, unsigned j) {
  unsigned long long z = seed + (((unsigned long long)stream << 32) | j) * 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z ^= z >> 31;
  const float u = ((float)(unsigned)(z >> 41) + 0.5f) * (1.0f / 8388608.0f);
  return -logf(-logf(u));
}

// first argmax over the block of (value, index) pairs; every thread gets the result
__device__ void block_argmax(float& mx, int& arg, float* sh, int* shi) {
  const int tid = threadIdx.x;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(mx, o, 64);
    const int oa = __shfl_xor(arg, o, 64);
    if (ov > mx || (ov == mx && oa < arg)) { mx = ov; arg = oa; }
  }
  __syncthreads();
  if ((tid & 63) == 0) { sh[tid >> 6] = mx; shi[tid >> 6] = arg; }
  __syncthreads();
  mx = sh[0]; arg = shi[0];
  for (int w = 1; w < 4; ++w)
    if (sh[w] > mx || (sh[w] == mx && shi[w] < arg)) { mx = sh[w]; arg = shi[w]; }
}

// max / first argmax / logsumexp of the masked row
__device__ void row_stats(const float* __restrict__ row, RowMask m, float* sh, int* shi,
                          float& out_max, int& out_arg, float& out_lse) {
  const int tid = threadIdx.x;
  float mx = NEG_INF;
  int arg = 0x7fffffff;
  for_row(row, m.lo, m.hi, [&](int i, float v) { if (v > mx) { mx = v; arg = i; } });
  block_argmax(mx, arg, sh, shi);
  float se = 0.f;
  for_row(row, m.lo, m.hi, [&](int, float v) { se += __expf(v - mx); });
  se = block_sum(se, sh);
  out_max = mx; out_arg = arg; out_lse = mx + __logf(se);
}

__global__ __launch_bounds__(256) void greedy_step_kernel(const GenState s, const FilterArgs f,
                                                          const LogitParams p, int n_vocab,
                                                          int max_steps, int s_new, int eos,
                                                          int pad) {
  __shared__ float sh[4];
  __shared__ int shi[4];
  const int i = blockIdx.x;
  const int t = *s.step;
  float* row = s.logits + (long)i * n_vocab;
  const RowMask m = prepare_row(row, n_vocab, f, p, s.lengths[i], s.generated + (long)i * max_steps, t, sh);
  float mx, lse; int arg;
  row_stats(row, m, sh, shi, mx, arg, lse);
  if (s.logits_out && t < max_steps) {    // return_logits: the log-probabilities selection works on
    float* dst = s.logits_out + ((long)i * max_steps + t) * n_vocab;
    for (int j = threadIdx.x; j < n_vocab; j += blockDim.x)
      dst[j] = (j < m.lo || j >= m.hi) ? NEG_INF : row[j] - lse;
  }
  if (p.sample) {          // a draw from softmax(row) instead of its mode
    float key = NEG_INF;
    int karg = 0x7fffffff;
    const unsigned stream = (unsigned)t * gridDim.x + i;
    for_row(row, m.lo, m.hi, [&](int j, float v) {
      if (v == NEG_INF) return;
      const float k = (v - lse) + gumbel_noise(p.seed, stream, j);
      if (k > key) { key = k; karg = j; }
    });
    block_argmax(key, karg, sh, shi);
    if (karg != 0x7fffffff) { arg = karg; mx = row[karg]; }
  }
  if (threadIdx.x == 0) {
    const int was_finished = s.finished[i];
    const int tok = was_finished ? pad : arg;
    if (!was_finished) s.scores[i] += mx - lse;
    const int fin = was_finished || tok == eos;
    // a window's own limit of new tokens: it ends after them as a call with that max_length would
    // (the token counts, no EOS is added)
    s.finished[i] = fin || (s.step_cap && t + 1 >= s.step_cap[i]);
    s.lengths[i] += fin ? 0 : 1;
    if (t < max_steps) s.generated[(long)i * max_steps + t] = tok;
    s.next_token[i] = tok;
    s.base_len[i] += s_new;
  }
}

__global__ void finish_step_kernel(const GenState s, int n_seq, int group, int first_n) {
  // all_finished = every group's first `first_n` sequences are finished; step += 1
  __shared__ int ok;
  if (threadIdx.x == 0) ok = 1;
  __syncthreads();
  for (int i = threadIdx.x; i < n_seq; i += blockDim.x)
    if ((i % group) < first_n && !s.finished[i]) ok = 0;
  __syncthreads();
  if (threadIdx.x == 0) { *s.all_finished = ok; *s.step += 1; }
}

__global__ void prepare_step_kernel(const GenState s, int* tokens_in, int* positions, int n_seq) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n_seq) { tokens_in[i] = s.next_token[i]; positions[i] = s.base_len[i]; }
}
void launch_prepare_step(const GenState& s, int* tokens_in, int* positions, int n_seq,
                         hipStream_t stream) {
  hipLaunchKernelGGL(prepare_step_kernel, dim3((n_seq + 255) / 256), dim3(256), 0, stream, s,
                     tokens_in, positions, n_seq);
}

void launch_greedy_step(const GenState& s, const FilterArgs& f, const LogitParams& p, int n_seq,
                        int n_vocab, int max_steps, int s_new, int eos, int pad,
                        hipStream_t stream) {
  hipLaunchKernelGGL(greedy_step_kernel, dim3(n_seq), dim3(256), 0, stream, s, f, p, n_vocab,
                     max_steps, s_new, eos, pad);
  hipLaunchKernelGGL(finish_step_kernel, dim3(1), dim3(256), 0, stream, s, n_seq, 1, 1);
}

// ---------------------------------------------------------------------------------------------
// beam search
// ---------------------------------------------------------------------------------------------
// Stage A: row <- running score + (finished ? {pad: 0, else -inf} : log_softmax(filtered row)),
// and the `beams` best entries of the row (ties to the lowest token id), which is all the
// per-window selection needs: within a row the length penalty is one positive factor.
#define BEAM_MAX 16
#define BEAM_KEEP 8       // candidates a thread keeps in registers during the one-pass selection
__global__ __launch_bounds__(256) void beam_logprob_kernel(const GenState s, const FilterArgs f,
                                                           const LogitParams p, int n_vocab,
                                                           int max_steps, int pad, int beams) {
  __shared__ float sh[4];
  __shared__ int shi[4];
  const int i = blockIdx.x, tid = threadIdx.x;
  const int t = *s.step;
  float* row = s.logits + (long)i * n_vocab;
  const RowMask m = prepare_row(row, n_vocab, f, p, s.lengths[i], s.generated + (long)i * max_steps, t, sh);
  float mx, lse; int arg;
  row_stats(row, m, sh, shi, mx, arg, lse);
  const float base = s.scores[i];
  const bool fin = s.finished[i];
  if (!p.sample && beams <= BEAM_KEEP) {
    // One pass writes the row and keeps, per thread, the BEAM_KEEP best of the entries it wrote
    // (sorted: value descending, token id ascending - a thread meets its ids in ascending order,
    // so an equal later value never moves ahead of an earlier one).  The block then takes the best
    // head `beams` times; the thread that owned it moves its list up.  Same order as the
    // pass-per-winner loop below, one read of the row instead of `beams`.
    float tv[BEAM_KEEP];
    int ti[BEAM_KEEP];
#pragma unroll
    for (int k = 0; k < BEAM_KEEP; ++k) { tv[k] = NEG_INF; ti[k] = 0x7fffffff; }
    for_row(row, 0, n_vocab, [&](int j, float x) {
      float v;
      if (fin) v = j == pad ? 0.f : NEG_INF;
      else v = (j < m.lo || j >= m.hi) ? NEG_INF : x - lse;
      v = base + v;
      row[j] = v;
      if (v > tv[BEAM_KEEP - 1]) {
        tv[BEAM_KEEP - 1] = v; ti[BEAM_KEEP - 1] = j;
#pragma unroll
        for (int k = BEAM_KEEP - 1; k > 0; --k)
          if (tv[k] > tv[k - 1]) {
            const float fv = tv[k]; tv[k] = tv[k - 1]; tv[k - 1] = fv;
            const int fi = ti[k]; ti[k] = ti[k - 1]; ti[k - 1] = fi;
          }
      }
    });
    for (int sel = 0; sel < beams; ++sel) {
      float bm = tv[0]; int ba = ti[0];
      block_argmax(bm, ba, sh, shi);
      if (tid == 0) { s.row_val[i * BEAM_MAX + sel] = bm; s.row_idx[i * BEAM_MAX + sel] = ba; }
      if (ba != 0x7fffffff && ti[0] == ba) {
#pragma unroll
        for (int k = 0; k + 1 < BEAM_KEEP; ++k) { tv[k] = tv[k + 1]; ti[k] = ti[k + 1]; }
        tv[BEAM_KEEP - 1] = NEG_INF; ti[BEAM_KEEP - 1] = 0x7fffffff;
      }
    }
    return;
  }
  for_row(row, 0, n_vocab, [&](int j, float x) {
    float v;
    if (fin) v = j == pad ? 0.f : NEG_INF;
    else v = (j < m.lo || j >= m.hi) ? NEG_INF : x - lse;
    row[j] = base + v;
  });
  if (p.sample) return;                 // the draw reads the whole rows
  // `beams` passes of "best entry after the previous winner" in the order (value descending,
  // token id ascending); a thread re-reads only the entries it wrote itself
  float prev_v = __builtin_inff();
  int prev_i = -1;
  for (int sel = 0; sel < beams; ++sel) {
    float bm = NEG_INF; int ba = 0x7fffffff;
    for_row(row, 0, n_vocab, [&](int j, float v) {
      const bool after = v < prev_v || (v == prev_v && j > prev_i);
      if (after && v > bm) { bm = v; ba = j; }
    });
    block_argmax(bm, ba, sh, shi);
    if (tid == 0) { s.row_val[i * BEAM_MAX + sel] = bm; s.row_idx[i * BEAM_MAX + sel] = ba; }
    prev_v = bm; prev_i = ba;
    if (ba == 0x7fffffff) prev_v = NEG_INF;     // nothing finite is left: the remaining slots stay empty
  }
}

// Stage B: per batch element, the `beams` best of the rows' candidates (first step: the first
// row only), ranked by value / (length + 1)^length_power, ties to the lowest flat index
// row * V + token.  One wave per batch element.
__global__ __launch_bounds__(64) void beam_topk_kernel(const GenState s, int beams, int n_vocab,
                                                       float length_power, int has_length_power) {
  const int b = blockIdx.x, lane = threadIdx.x;
  const int t = *s.step;
  const int rows = t == 0 ? 1 : beams;
  // lane -> candidates lane, lane + 64, ... of the rows * beams (<= 256)
  float cvv[4]; int cii[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int c = lane + 64 * q;
    cvv[q] = NEG_INF; cii[q] = 0x7fffffff;
    if (c < rows * beams) {
      const int r = c / beams, k = c % beams;
      const int seq = b * beams + r;
      const int tok = s.row_idx[seq * BEAM_MAX + k];
      float v = s.row_val[seq * BEAM_MAX + k];
      if (tok != 0x7fffffff) {
        if (has_length_power) v = v * (1.0f / powf((float)(s.lengths[seq] + 1), length_power));
        cvv[q] = v; cii[q] = r * n_vocab + tok;
      }
    }
  }
  for (int sel = 0; sel < beams; ++sel) {
    float bm = NEG_INF; int ba = 0x7fffffff;
#pragma unroll
    for (int q = 0; q < 4; ++q)
      if (cvv[q] > bm || (cvv[q] == bm && cii[q] < ba)) { bm = cvv[q]; ba = cii[q]; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float ov = __shfl_xor(bm, o, 64); const int oa = __shfl_xor(ba, o, 64);
      if (ov > bm || (ov == bm && oa < ba)) { bm = ov; ba = oa; }
    }
    if (lane == 0) { s.cand_idx[b * beams + sel] = ba; s.cand_val[b * beams + sel] = bm; }
#pragma unroll
    for (int q = 0; q < 4; ++q)
      if (cii[q] == ba) { cvv[q] = NEG_INF; cii[q] = 0x7fffffff; }
  }
}

// Stage B when sampling: draw d of batch element b is one categorical sample (with replacement,
// like `keras.random.categorical(logits, beams)`) over the same (beams * V) candidates, with
// weights exp(value / (length + 1)^length_power).
__global__ __launch_bounds__(256) void beam_sample_kernel(const GenState s, int beams, int n_vocab,
                                                          float length_power, int has_length_power,
                                                          unsigned long long seed) {
  __shared__ float sh[4];
  __shared__ int shi[4];
  const int b = blockIdx.x / beams, tid = threadIdx.x;
  const int t = *s.step;
  const int rows = t == 0 ? 1 : beams;
  const unsigned stream = (unsigned)t * gridDim.x + blockIdx.x;
  float key = NEG_INF;
  int karg = 0x7fffffff;
  for (int r = 0; r < rows; ++r) {
    const int seq = b * beams + r;
    const float* row = s.logits + (long)seq * n_vocab;
    float inv = 1.0f;
    if (has_length_power) inv = 1.0f / powf((float)(s.lengths[seq] + 1), length_power);
    for (int j = tid; j < n_vocab; j += 256) {
      float v = row[j];
      if (v == NEG_INF) continue;
      if (has_length_power) v *= inv;
      const float k = v + gumbel_noise(seed, stream, (unsigned)(r * n_vocab + j));
      if (k > key) { key = k; karg = r * n_vocab + j; }
    }
  }
  block_argmax(key, karg, sh, shi);
  if (tid == 0) { s.cand_idx[blockIdx.x] = karg; s.cand_val[blockIdx.x] = key; }
}

// Stage C: rebuild every sequence from its parent beam (generation_utils.py:385-408, 658-676).
__global__ __launch_bounds__(256) void beam_reorder_kernel(const GenState s, int beams, int n_vocab,
                                                           int max_steps, int max_len, int s_new,
                                                           int eos, int pad) {
  const int i = blockIdx.x;                 // new sequence
  const int b = i / beams;
  const int t = *s.step;
  int idx = s.cand_idx[i];
  if (idx == 0x7fffffff || idx < 0) idx = 0;
  const int parent = b * beams + idx / n_vocab;
  int tok = idx % n_vocab;
  const int old_len = s.base_len[parent];
  for (int j = threadIdx.x; j < max_steps; j += 256)
    s.generated_tmp[(long)i * max_steps + j] = s.generated[(long)parent * max_steps + j];
  for (int p = threadIdx.x; p < max_len; p += 256) {
    int v = s.slot_of[(long)parent * max_len + p];
    if (p >= old_len && p < old_len + s_new) v = parent;   // written by the parent during this step
    s.slot_of_tmp[(long)i * max_len + p] = v;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    int fin = s.finished[parent];
    int len = s.lengths[parent];
    const float score = s.logits[(long)parent * n_vocab + tok];
    tok = fin ? pad : tok;
    fin = fin || tok == eos;
    len += fin ? 0 : 1;
    if (s.step_cap && t + 1 >= s.step_cap[b]) fin = 1;
    s.scores_tmp[i] = score;
    s.finished_tmp[i] = fin;
    s.lengths_tmp[i] = len;
    if (t < max_steps) s.generated_tmp[(long)i * max_steps + t] = tok;
    s.next_token[i] = tok;
  }
}

// `freeze`: early stopping is on.  The host polls "all finished" only every few steps, so steps
// may still be replayed after the one that finished the first `num_sentences` beams of every
// window; the reference stops right there (generation_utils.py:326-330), so from then on nothing
// is committed (unfinished lower beams must not overtake the finished ones afterwards).
__global__ __launch_bounds__(256) void beam_commit_kernel(const GenState s, int max_steps,
                                                          int max_len, int s_new, int freeze) {
  const int i = blockIdx.x;
  if (freeze && *s.all_finished) return;
  for (int j = threadIdx.x; j < max_steps; j += 256)
    s.generated[(long)i * max_steps + j] = s.generated_tmp[(long)i * max_steps + j];
  for (int p = threadIdx.x; p < max_len; p += 256)
    s.slot_of[(long)i * max_len + p] = s.slot_of_tmp[(long)i * max_len + p];
  if (threadIdx.x == 0) {
    s.scores[i] = s.scores_tmp[i];
    s.finished[i] = s.finished_tmp[i];
    s.lengths[i] = s.lengths_tmp[i];
    s.base_len[i] += s_new;
  }
}

void launch_beam_step(const GenState& s, const FilterArgs& f, const LogitParams& p, int batch,
                      int beams, int num_sentences, int n_vocab, int max_steps, int max_len,
                      int s_new, int eos, int pad, float length_power, int has_length_power,
                      int early_stopping, hipStream_t stream) {
  const int n_seq = batch * beams;
  hipLaunchKernelGGL(beam_logprob_kernel, dim3(n_seq), dim3(256), 0, stream, s, f, p, n_vocab,
                     max_steps, pad, beams);
  if (p.sample)
    hipLaunchKernelGGL(beam_sample_kernel, dim3(n_seq), dim3(256), 0, stream, s, beams, n_vocab,
                       length_power, has_length_power, p.seed);
  else
    hipLaunchKernelGGL(beam_topk_kernel, dim3(batch), dim3(64), 0, stream, s, beams, n_vocab,
                       length_power, has_length_power);
  hipLaunchKernelGGL(beam_reorder_kernel, dim3(n_seq), dim3(256), 0, stream, s, beams, n_vocab,
                     max_steps, max_len, s_new, eos, pad);
  hipLaunchKernelGGL(beam_commit_kernel, dim3(n_seq), dim3(256), 0, stream, s, max_steps, max_len,
                     s_new, early_stopping);
  hipLaunchKernelGGL(finish_step_kernel, dim3(1), dim3(256), 0, stream, s, n_seq, beams,
                     num_sentences);
}
