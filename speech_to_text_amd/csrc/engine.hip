// Whisper engine: weights, activation buffers and the encoder / decoder / generation drivers
// behind the C ABI of include/stt_mi355x.h.  Host code only enqueues kernels (kernels.h); the
// generation loop synchronises every `check_every` tokens to read the "all finished" flag.
//
// Reference call path replaced: Transformer.infer (architectures/transformers/
// transformer_arch.py:1029-1065) = encoder once + generation_utils.infer (greedy / beam).
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "kernels.h"
#include "stt_common.h"

namespace {

struct DevBuf {
  void* p = nullptr;
  size_t bytes = 0;
};

struct LayerW {
  bf16_t *wqkv = nullptr, *wo = nullptr, *w1 = nullptr, *w2 = nullptr;
  float *bqkv = nullptr, *bo = nullptr, *b1 = nullptr, *b2 = nullptr;
  float *ln1g = nullptr, *ln1b = nullptr, *ln2g = nullptr, *ln2b = nullptr;
  // decoder cross-attention
  bf16_t *wq_c = nullptr, *wkv_c = nullptr, *wo_c = nullptr;
  float *bq_c = nullptr, *bkv_c = nullptr, *bo_c = nullptr, *lncg = nullptr, *lncb = nullptr;
};

}  // namespace

struct stt_engine {
  stt_model_config cfg;
  int max_batch = 0, max_beams = 0, max_len = 0, max_prompt = 0;
  int device = 0, n_cu = 256;
  bool finalized = false;
  bool buffers_ok = false;             // false after a failed (re-)allocation: every entry point then refuses
  hipStream_t gen_stream = nullptr;    // generation runs here: a captured token step needs a real stream
  hipEvent_t gen_event = nullptr;
  std::map<std::string, std::vector<float>> host_w;
  std::vector<void*> allocs;           // weights: live as long as the engine
  std::vector<void*> buf_allocs;       // activation / cache / state buffers: replaced by stt_engine_reserve
  int* pinned = nullptr;               // page-locked host staging for prompts and results
  size_t pinned_ints = 0;
  // instantiated token-step graphs, keyed by everything the recorded launches bake in
  std::map<std::string, hipGraphExec_t> step_graphs;

  // weights
  bf16_t *conv1_w = nullptr, *conv2_w = nullptr, *tok_emb = nullptr;
  float *conv1_b = nullptr, *conv2_b = nullptr, *enc_pos = nullptr, *pos_emb = nullptr;
  float *enc_lng = nullptr, *enc_lnb = nullptr, *dec_lng = nullptr, *dec_lnb = nullptr;
  int conv1_k = 0;
  std::vector<LayerW> enc, dec;

  // encoder buffers
  bf16_t *mel_pad = nullptr, *h1_pad = nullptr, *eh = nullptr, *eqkv = nullptr, *evt = nullptr,
         *eatt = nullptr, *ef = nullptr, *enc_out = nullptr;
  float* ex = nullptr;
  float* eslabs = nullptr;  // split-K partial sums of the encoder's residual projections (a few windows only)
  std::vector<bf16_t*> cross_k, cross_v;
  int enc_batch = 0;       // windows held by the last stt_encode
  int step_len = 0, step_seqs = 0;   // stt_decode_step: positions in the cache, sequences of the history
  int step_per_window = 1;
  // stt_decode_step_attention: where the pass in progress leaves its attention weights (null otherwise)
  float* attn_self_out = nullptr;
  float* attn_cross_out = nullptr;
  int attn_first_layer = 0, attn_len = 0;

  // decoder buffers
  int max_seq = 0, max_rows = 0;
  float *dx = nullptr, *slabs = nullptr;
  float *dx2 = nullptr, *slabs2 = nullptr;   // rows <= kLnFuseRows: second residual buffer, query slabs
  bf16_t *dh = nullptr, *dqkv = nullptr, *datt = nullptr, *dq = nullptr, *df = nullptr;
  std::vector<bf16_t*> self_k, self_v;
  int *tokens_in = nullptr, *positions = nullptr;
  GenState gs{};
  int *d_suppress = nullptr, *d_suppress_first = nullptr, *d_step_cap = nullptr;
  int suppress_cap = 0;
};

namespace {

template <typename T>
int dev_alloc_in(std::vector<void*>& owner, T** out, size_t count) {
  void* p = nullptr;
  const size_t bytes = (count * sizeof(T) + 255) & ~(size_t)255;
  STT_HIP_OR_FAIL(hipMalloc(&p, bytes), (void)0);
  owner.push_back(p);
  *out = (T*)p;
  return STT_OK;
}
template <typename T>
int dev_alloc(stt_engine* e, T** out, size_t count) { return dev_alloc_in(e->allocs, out, count); }
template <typename T>
int buf_alloc(stt_engine* e, T** out, size_t count) { return dev_alloc_in(e->buf_allocs, out, count); }

void drop_step_graphs(stt_engine* e) {
  for (auto& kv : e->step_graphs) (void)hipGraphExecDestroy(kv.second);
  e->step_graphs.clear();
}
void free_buffers(stt_engine* e) {
  drop_step_graphs(e);
  for (void* p : e->buf_allocs) (void)hipFree(p);
  e->buf_allocs.clear();
  if (e->pinned) (void)hipHostFree(e->pinned);
  e->pinned = nullptr;
  e->pinned_ints = 0;
}

bf16_t host_f2bf(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  u += 0x7FFFu + ((u >> 16) & 1u);
  return (bf16_t)(u >> 16);
}

int upload_bf16(stt_engine* e, bf16_t** out, const std::vector<float>& v) {
  std::vector<bf16_t> h(v.size());
  for (size_t i = 0; i < v.size(); ++i) h[i] = host_f2bf(v[i]);
  int rc = dev_alloc(e, out, v.size());
  if (rc) return rc;
  STT_HIP_OR_FAIL(hipMemcpy(*out, h.data(), h.size() * sizeof(bf16_t), hipMemcpyHostToDevice), (void)0);
  return STT_OK;
}
int upload_f32(stt_engine* e, float** out, const std::vector<float>& v) {
  int rc = dev_alloc(e, out, v.size());
  if (rc) return rc;
  STT_HIP_OR_FAIL(hipMemcpy(*out, v.data(), v.size() * sizeof(float), hipMemcpyHostToDevice), (void)0);
  return STT_OK;
}

const std::vector<float>* find_w(stt_engine* e, const std::string& name, size_t numel) {
  auto it = e->host_w.find(name);
  if (it == e->host_w.end()) {
    stt_fail(STT_ERR_INVALID_ARGUMENT, "weight '%s' was never set", name.c_str());
    return nullptr;
  }
  if (it->second.size() != numel) {
    stt_fail(STT_ERR_INVALID_ARGUMENT, "weight '%s' has %zu elements, expected %zu", name.c_str(),
             it->second.size(), numel);
    return nullptr;
  }
  return &it->second;
}

#define GETW(var, name, numel)                      \
  const std::vector<float>* var = find_w(e, name, numel); \
  if (!var) return STT_ERR_INVALID_ARGUMENT;

// conv weight [out][in][3] -> GEMM weight [out][kpad], k = tap * in + channel
std::vector<float> conv_as_gemm(const std::vector<float>& w, int out, int in, int kpad) {
  std::vector<float> g((size_t)out * kpad, 0.f);
  for (int o = 0; o < out; ++o)
    for (int c = 0; c < in; ++c)
      for (int t = 0; t < 3; ++t) g[(size_t)o * kpad + t * in + c] = w[((size_t)o * in + c) * 3 + t];
  return g;
}

std::vector<float> concat(std::initializer_list<const std::vector<float>*> parts) {
  std::vector<float> r;
  for (auto p : parts) r.insert(r.end(), p->begin(), p->end());
  return r;
}

int load_layer(stt_engine* e, const std::string& p, int d, bool decoder, LayerW& L) {
  const size_t dd = (size_t)d * d;
  const std::vector<float> zeros(d, 0.f);
  {
    GETW(q, p + "attn.query.weight", dd) GETW(k, p + "attn.key.weight", dd)
    GETW(v, p + "attn.value.weight", dd) GETW(o, p + "attn.out.weight", dd)
    GETW(qb, p + "attn.query.bias", d) GETW(vb, p + "attn.value.bias", d)
    GETW(ob, p + "attn.out.bias", d)
    int rc;
    if ((rc = upload_bf16(e, &L.wqkv, concat({q, k, v})))) return rc;
    if ((rc = upload_f32(e, &L.bqkv, concat({qb, &zeros, vb})))) return rc;   // key has no bias
    if ((rc = upload_bf16(e, &L.wo, *o))) return rc;
    if ((rc = upload_f32(e, &L.bo, *ob))) return rc;
    GETW(g1, p + "attn_ln.weight", d) GETW(b1, p + "attn_ln.bias", d)
    GETW(g2, p + "mlp_ln.weight", d) GETW(b2, p + "mlp_ln.bias", d)
    if ((rc = upload_f32(e, &L.ln1g, *g1)) || (rc = upload_f32(e, &L.ln1b, *b1)) ||
        (rc = upload_f32(e, &L.ln2g, *g2)) || (rc = upload_f32(e, &L.ln2b, *b2))) return rc;
    GETW(w1, p + "mlp.0.weight", 4 * dd) GETW(bb1, p + "mlp.0.bias", 4 * (size_t)d)
    GETW(w2, p + "mlp.2.weight", 4 * dd) GETW(bb2, p + "mlp.2.bias", d)
    if ((rc = upload_bf16(e, &L.w1, *w1)) || (rc = upload_f32(e, &L.b1, *bb1)) ||
        (rc = upload_bf16(e, &L.w2, *w2)) || (rc = upload_f32(e, &L.b2, *bb2))) return rc;
  }
  if (decoder) {
    GETW(q, p + "cross_attn.query.weight", dd) GETW(k, p + "cross_attn.key.weight", dd)
    GETW(v, p + "cross_attn.value.weight", dd) GETW(o, p + "cross_attn.out.weight", dd)
    GETW(qb, p + "cross_attn.query.bias", d) GETW(vb, p + "cross_attn.value.bias", d)
    GETW(ob, p + "cross_attn.out.bias", d)
    GETW(g, p + "cross_attn_ln.weight", d) GETW(b, p + "cross_attn_ln.bias", d)
    int rc;
    if ((rc = upload_bf16(e, &L.wq_c, *q)) || (rc = upload_f32(e, &L.bq_c, *qb)) ||
        (rc = upload_bf16(e, &L.wkv_c, concat({k, v}))) ||
        (rc = upload_f32(e, &L.bkv_c, concat({&zeros, vb}))) ||
        (rc = upload_bf16(e, &L.wo_c, *o)) || (rc = upload_f32(e, &L.bo_c, *ob)) ||
        (rc = upload_f32(e, &L.lncg, *g)) || (rc = upload_f32(e, &L.lncb, *b))) return rc;
  }
  return STT_OK;
}

GemmArgs gemm_args(const bf16_t* a, long lda, const bf16_t* w, const float* bias, void* out, long ldc,
                   int m, int n, int k, int gelu_tanh) {
  GemmArgs g{};
  g.a = a; g.lda = lda; g.a_batch_stride = 0; g.w = w; g.bias = bias; g.out = out; g.ldc = ldc;
  g.out_batch_stride = 0; g.pos = nullptr; g.pos_ld = 0; g.m = m; g.n = n; g.k = k; g.batch = 1;
  g.gelu_tanh = gelu_tanh;
  return g;
}

// y = a @ w^T (+bias) on `rows` rows: skinny kernel for decode-sized M, tiled kernel otherwise.
// resid != null: the product is added to the fp32 residual stream; for the skinny path that
// happens later (the partial sums stay pending in e->slabs and are consumed by the next
// add_layernorm), *pending_slabs returns how many.
// Up to this many rows a layer goes through the decode-step kernels (weights streamed once, split-K
// partial sums consumed by the next kernel); above it through the tiled large-M kernels.
constexpr int kDecodeRows = 640;       // (1 200 windows split evenly over two calls or two ranks: 600 rows per token step)
// With a single row (one clip, greedy: the way the reference calls) the three LayerNorms of a
// decoder layer and the final one run as the prologue of the projection that consumes them
// (launch_gemm_skinny_ln): 8 instead of 11 launches per layer, same bits, 1.96 -> 1.80 ms per token
// at large-v3.  The prologue handles its rows one after the other, so from two rows on the separate
// one-block-per-row kernel is faster (2.02 against 2.11 ms per token at two rows, 2.09 against 2.70 at four).
constexpr int kLnFuseRows = 1;

void dense(stt_engine* e, const bf16_t* a, long lda, const bf16_t* w, const float* bias, int rows,
           int n, int k, int epilogue, void* out, long ldc, int* pending_slabs, hipStream_t st) {
  GemmArgs g = gemm_args(a, lda, w, bias, out, ldc, rows, n, k, e->cfg.gelu_tanh);
  if (pending_slabs) *pending_slabs = 0;
  if (rows <= kDecodeRows) {
    if (epilogue == EPI_RESID_F32) {
      const int s = pick_split_k(rows, n, k, e->n_cu);
      launch_gemm_skinny(g, -1, s, e->slabs, st);
      *pending_slabs = s;
    } else {
      launch_gemm_skinny(g, epilogue, 1, nullptr, st);
    }
  } else {
    launch_gemm_big(g, epilogue, st);
  }
}

// Row pitch of the encoder's V^T image [head * 64 + feature][row of the batch]: every row of the
// image starts on a 128-byte boundary and ends with at least 64 elements of padding.
long vt_pitch(const stt_engine* e) { return ((long)e->max_batch * 1500 + 64 + 63) / 64 * 64; }

// Encoder residual projections (N = d: 10 column tiles at large-v3) of a few windows leave most of
// the chip idle - one window is 6 x 10 = 60 tiles of 256 x 128 for 256 CUs - so up to
// kEncSplitWindows windows they split K over 4 blocks per tile; the partial sums go to slabs that
// the following LayerNorm adds up with the bias and the residual (as the decode steps do).  The
// split does not depend on the number of windows below that limit, so a window's result does not
// depend on how many others (up to the limit) were encoded with it.  1 = no split.
constexpr int kEncSplitWindows = 3;      // measured: 12.2 -> 8.7 ms at one window, 11.6 -> 9.4 at two, even at three, slower from four
constexpr int kEncSplit = 4;
int enc_split(const stt_engine* e, int batch, int k) {
  static const int forced = getenv("STT_MI355X_ENC_SPLIT") ? atoi(getenv("STT_MI355X_ENC_SPLIT")) : 0;
  const int s = forced > 0 ? forced : kEncSplit;
  if (batch > kEncSplitWindows || s < 2 || s > kEncSplit || (k / 64) % s) return 1;
  return s;
}

// Every buffer whose size depends on the capacity (max_batch, max_beams, max_len, max_prompt).
int alloc_buffers(stt_engine* e) {
  const stt_model_config& c = e->cfg;
  const int d = c.n_audio_state, V = c.n_vocab;
  int rc;
  const size_t B = e->max_batch, T = 1500, H = c.n_audio_head;
  const size_t R = B * T;
  if ((rc = buf_alloc(e, &e->mel_pad, B * 3002 * c.n_mels + 4096)) ||
      (rc = buf_alloc(e, &e->h1_pad, B * 3002 * d + 4096)) || (rc = buf_alloc(e, &e->ex, R * d)) ||
      (rc = buf_alloc(e, &e->eh, R * d)) || (rc = buf_alloc(e, &e->eqkv, R * 3 * d)) ||
      (rc = buf_alloc(e, &e->evt, H * 64 * (size_t)vt_pitch(e))) || (rc = buf_alloc(e, &e->eatt, R * d)) ||
      (rc = buf_alloc(e, &e->ef, R * 4 * d)) || (rc = buf_alloc(e, &e->enc_out, R * d)) ||
      (rc = buf_alloc(e, &e->eslabs, (size_t)kEncSplit * (B < kEncSplitWindows ? B : kEncSplitWindows) * T * d))) return rc;
  STT_HIP_OR_FAIL(hipMemset(e->mel_pad, 0, (B * 3002 * c.n_mels + 4096) * sizeof(bf16_t)), (void)0);
  STT_HIP_OR_FAIL(hipMemset(e->h1_pad, 0, (B * 3002 * d + 4096) * sizeof(bf16_t)), (void)0);
  // the padding of the V^T image is read by the attention kernel (under zero weights) and never written
  STT_HIP_OR_FAIL(hipMemset(e->evt, 0, H * 64 * (size_t)vt_pitch(e) * sizeof(bf16_t)), (void)0);
  e->cross_k.resize(c.n_text_layer); e->cross_v.resize(c.n_text_layer);
  for (int l = 0; l < c.n_text_layer; ++l)
    if ((rc = buf_alloc(e, &e->cross_k[l], B * H * T * 64)) ||
        (rc = buf_alloc(e, &e->cross_v[l], B * H * T * 64))) return rc;

  e->max_seq = e->max_batch * e->max_beams;
  e->max_rows = e->max_seq * e->max_prompt;
  const size_t S = e->max_seq, MR = e->max_rows, L = e->max_len;
  const size_t slab_rows = MR < kDecodeRows ? MR : kDecodeRows;
  if ((rc = buf_alloc(e, &e->dx, MR * d)) || (rc = buf_alloc(e, &e->dh, MR * d)) ||
      (rc = buf_alloc(e, &e->dqkv, MR * 3 * d)) || (rc = buf_alloc(e, &e->datt, MR * d)) ||
      (rc = buf_alloc(e, &e->dq, MR * d)) || (rc = buf_alloc(e, &e->df, MR * 4 * d)) ||
      (rc = buf_alloc(e, &e->slabs, (size_t)8 * slab_rows * d)) ||
      (rc = buf_alloc(e, &e->dx2, (size_t)kLnFuseRows * d)) ||
      (rc = buf_alloc(e, &e->slabs2, (size_t)8 * kLnFuseRows * d)) ||
      (rc = buf_alloc(e, &e->tokens_in, MR)) || (rc = buf_alloc(e, &e->positions, MR))) return rc;
  e->self_k.resize(c.n_text_layer); e->self_v.resize(c.n_text_layer);
  for (int l = 0; l < c.n_text_layer; ++l)
    if ((rc = buf_alloc(e, &e->self_k[l], S * H * L * 64)) ||
        (rc = buf_alloc(e, &e->self_v[l], S * H * L * 64))) return rc;
  GenState& g = e->gs;
  if ((rc = buf_alloc(e, &g.logits, S * V)) || (rc = buf_alloc(e, &g.generated, S * L)) ||
      (rc = buf_alloc(e, &g.lengths, S)) || (rc = buf_alloc(e, &g.scores, S)) ||
      (rc = buf_alloc(e, &g.finished, S)) || (rc = buf_alloc(e, &g.next_token, S)) ||
      (rc = buf_alloc(e, &g.base_len, S)) || (rc = buf_alloc(e, &g.slot_of, S * L)) ||
      (rc = buf_alloc(e, &g.step, 64)) || (rc = buf_alloc(e, &g.cand_val, S)) ||
      (rc = buf_alloc(e, &g.cand_idx, S)) || (rc = buf_alloc(e, &g.row_val, S * 16)) ||
      (rc = buf_alloc(e, &g.row_idx, S * 16)) || (rc = buf_alloc(e, &g.generated_tmp, S * L)) ||
      (rc = buf_alloc(e, &g.lengths_tmp, S)) || (rc = buf_alloc(e, &g.finished_tmp, S)) ||
      (rc = buf_alloc(e, &g.slot_of_tmp, S * L)) || (rc = buf_alloc(e, &g.scores_tmp, S))) return rc;
  g.all_finished = g.step + 16;
  g.logits_out = nullptr;
  e->suppress_cap = 4096;
  if ((rc = buf_alloc(e, &e->d_suppress, e->suppress_cap)) ||
      (rc = buf_alloc(e, &e->d_suppress_first, e->suppress_cap)) ||
      (rc = buf_alloc(e, &e->d_step_cap, B))) return rc;
  // page-locked staging: prompt tokens + positions, suppress lists, results, the "finished" flag
  e->pinned_ints = 2 * MR + 2 * (size_t)e->suppress_cap + S * L + 2 * S + 64 + B;
  STT_HIP_OR_FAIL(hipHostMalloc((void**)&e->pinned, e->pinned_ints * sizeof(int), hipHostMallocDefault), (void)0);
  return STT_OK;
}


}  // namespace

extern "C" {

int stt_engine_create(const stt_model_config* cfg, int max_batch, int max_beams, int max_text_len,
                      int max_prompt_len, stt_engine** out) {
  if (!cfg || !out) return stt_fail(STT_ERR_INVALID_ARGUMENT, "null argument");
  if (cfg->n_audio_state % 64 || cfg->n_text_state % 64 ||
      cfg->n_audio_state != cfg->n_audio_head * 64 || cfg->n_text_state != cfg->n_text_head * 64)
    return stt_fail(STT_ERR_UNSUPPORTED, "head size must be 64 (state %d/%d, heads %d/%d)",
                    cfg->n_audio_state, cfg->n_text_state, cfg->n_audio_head, cfg->n_text_head);
  if (cfg->n_audio_state != cfg->n_text_state)
    return stt_fail(STT_ERR_UNSUPPORTED, "encoder and decoder widths must match");
  if (cfg->n_audio_ctx != 1500) return stt_fail(STT_ERR_UNSUPPORTED, "n_audio_ctx must be 1500");
  if (cfg->n_mels % 8 || cfg->n_audio_state > 2048)
    return stt_fail(STT_ERR_UNSUPPORTED, "n_mels must be a multiple of 8 and the width <= 2048");
  if (max_batch < 1 || max_beams < 1 || max_beams > 16 || max_text_len < 2 ||
      max_text_len > cfg->n_text_ctx || max_text_len > 512 || max_prompt_len < 1 ||
      max_prompt_len >= max_text_len)
    return stt_fail(STT_ERR_INVALID_ARGUMENT, "bad capacity (batch %d, beams %d, text %d, prompt %d)",
                    max_batch, max_beams, max_text_len, max_prompt_len);
  stt_engine* e = new stt_engine();
  e->cfg = *cfg;
  e->max_batch = max_batch; e->max_beams = max_beams; e->max_len = max_text_len;
  e->max_prompt = max_prompt_len;
  if (hipGetDevice(&e->device) != hipSuccess) {
    delete e;
    return stt_fail(STT_ERR_HIP, "no current HIP device");
  }
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, e->device) == hipSuccess) e->n_cu = prop.multiProcessorCount;
  *out = e;
  return STT_OK;
}

void stt_engine_destroy(stt_engine* e) {
  if (!e) return;
  free_buffers(e);
  for (void* p : e->allocs) (void)hipFree(p);
  if (e->gen_stream) (void)hipStreamDestroy(e->gen_stream);
  if (e->gen_event) (void)hipEventDestroy(e->gen_event);
  delete e;
}

int stt_engine_set_weight(stt_engine* e, const char* name, const float* data, int64_t numel) {
  if (!e || !name || !data || numel <= 0) return stt_fail(STT_ERR_INVALID_ARGUMENT, "null argument");
  if (e->finalized) return stt_fail(STT_ERR_INVALID_ARGUMENT, "engine already finalized");
  e->host_w[name].assign(data, data + numel);
  return STT_OK;
}

int stt_engine_finalize(stt_engine* e) {
  if (!e) return stt_fail(STT_ERR_INVALID_ARGUMENT, "null argument");
  if (e->finalized) return STT_OK;
  const stt_model_config& c = e->cfg;
  const int d = c.n_audio_state, V = c.n_vocab;
  int rc;
  {
    e->conv1_k = ((3 * c.n_mels + 63) / 64) * 64;
    GETW(w1, "encoder.conv1.weight", (size_t)d * c.n_mels * 3) GETW(b1, "encoder.conv1.bias", d)
    GETW(w2, "encoder.conv2.weight", (size_t)d * d * 3) GETW(b2, "encoder.conv2.bias", d)
    if ((rc = upload_bf16(e, &e->conv1_w, conv_as_gemm(*w1, d, c.n_mels, e->conv1_k))) ||
        (rc = upload_f32(e, &e->conv1_b, *b1)) ||
        (rc = upload_bf16(e, &e->conv2_w, conv_as_gemm(*w2, d, d, 3 * d))) ||
        (rc = upload_f32(e, &e->conv2_b, *b2))) return rc;
    // sinusoids (whisper_arch.py:93-104), computed in float32 like the reference
    std::vector<float> pos((size_t)1500 * d);
    const float inc = logf(10000.f) / (float)(d / 2 - 1);
    for (int t = 0; t < 1500; ++t)
      for (int i = 0; i < d / 2; ++i) {
        const float inv = expf(-inc * (float)i);
        const float st = (float)t * inv;
        pos[(size_t)t * d + i] = sinf(st);
        pos[(size_t)t * d + d / 2 + i] = cosf(st);
      }
    auto it = e->host_w.find("encoder.positional_embedding");   // optional override
    if (it != e->host_w.end() && it->second.size() == pos.size()) pos = it->second;
    if ((rc = upload_f32(e, &e->enc_pos, pos))) return rc;
    GETW(g, "encoder.ln_post.weight", d) GETW(b, "encoder.ln_post.bias", d)
    if ((rc = upload_f32(e, &e->enc_lng, *g)) || (rc = upload_f32(e, &e->enc_lnb, *b))) return rc;
    GETW(te, "decoder.token_embedding.weight", (size_t)V * d)
    GETW(pe, "decoder.positional_embedding", (size_t)c.n_text_ctx * d)
    GETW(dg, "decoder.ln.weight", d) GETW(db, "decoder.ln.bias", d)
    if ((rc = upload_bf16(e, &e->tok_emb, *te)) || (rc = upload_f32(e, &e->pos_emb, *pe)) ||
        (rc = upload_f32(e, &e->dec_lng, *dg)) || (rc = upload_f32(e, &e->dec_lnb, *db))) return rc;
  }
  e->enc.resize(c.n_audio_layer);
  for (int l = 0; l < c.n_audio_layer; ++l)
    if ((rc = load_layer(e, "encoder.blocks." + std::to_string(l) + ".", d, false, e->enc[l]))) return rc;
  e->dec.resize(c.n_text_layer);
  for (int l = 0; l < c.n_text_layer; ++l)
    if ((rc = load_layer(e, "decoder.blocks." + std::to_string(l) + ".", d, true, e->dec[l]))) return rc;
  e->host_w.clear();

  e->finalized = true;
  rc = alloc_buffers(e);
  e->buffers_ok = rc == STT_OK;
  if (rc) { free_buffers(e); (void)hipGetLastError(); }   // weights stay: stt_engine_reserve with a capacity that fits may retry
  return rc;
}

int stt_engine_reserve(stt_engine* e, int max_batch, int max_beams, int max_text_len, int max_prompt_len) {
  if (!e) return stt_fail(STT_ERR_INVALID_ARGUMENT, "null argument");
  if (!e->finalized) return stt_fail(STT_ERR_INVALID_ARGUMENT, "engine not finalized");
  const int nb = max_batch > e->max_batch ? max_batch : e->max_batch;
  const int nm = max_beams > e->max_beams ? max_beams : e->max_beams;
  const int nl = max_text_len > e->max_len ? max_text_len : e->max_len;
  const int np = max_prompt_len > e->max_prompt ? max_prompt_len : e->max_prompt;
  if (e->buffers_ok && nb == e->max_batch && nm == e->max_beams && nl == e->max_len && np == e->max_prompt)
    return STT_OK;
  if (nm > 16 || nl > e->cfg.n_text_ctx || nl > 512 || np >= nl || nb < 1)
    return stt_fail(STT_ERR_INVALID_ARGUMENT, "bad capacity (batch %d, beams %d, text %d, prompt %d)", nb, nm, nl, np);
  STT_HIP_OR_FAIL(hipDeviceSynchronize(), (void)0);        // nothing may still read the old buffers
  free_buffers(e);
  const int ob = e->max_batch, om = e->max_beams, ol = e->max_len, op = e->max_prompt;
  const bool had = e->buffers_ok;
  e->max_batch = nb; e->max_beams = nm; e->max_len = nl; e->max_prompt = np;
  e->enc_batch = 0;
  e->buffers_ok = false;
  int rc = alloc_buffers(e);
  if (rc == STT_OK) { e->buffers_ok = true; return STT_OK; }
  // Out of memory part-way: nothing half-allocated may stay reachable.  Go back to the capacity
  // that did fit (its buffers are gone, so they are allocated afresh); if even that fails the
  // engine keeps its weights but refuses encode / generate until a reserve succeeds.
  const std::string why = stt_last_error();
  free_buffers(e);
  (void)hipGetLastError();               // the failed hipMalloc must not surface in a later call's check
  e->max_batch = ob; e->max_beams = om; e->max_len = ol; e->max_prompt = op;
  if (had && alloc_buffers(e) == STT_OK) e->buffers_ok = true;
  else free_buffers(e);
  return stt_fail(rc, "stt_engine_reserve(batch %d, beams %d, text %d, prompt %d) failed: %s; the engine %s",
                  nb, nm, nl, np, why.c_str(),
                  e->buffers_ok ? "keeps its previous capacity (encode again before decoding)"
                                : "holds no buffers until a reserve succeeds");
}

int stt_engine_usable(const stt_engine* e) { return e && e->finalized && e->buffers_ok ? 1 : 0; }

}  // extern "C"

namespace {
// ex += a @ w^T + bias for the encoder.  Returns 0 when the sum is already in ex, or the number of
// split-K slabs in e->eslabs that the next LayerNorm has to add (with the bias).
int residual_projection(stt_engine* e, const bf16_t* a, long lda, const bf16_t* w, const float* bias, int rows,
                        int batch, int k, hipStream_t st) {
  const int d = e->cfg.n_audio_state;
  const int s = enc_split(e, batch, k);
  if (s == 1) {
    launch_gemm_big(gemm_args(a, lda, w, bias, e->ex, d, rows, d, k, e->cfg.gelu_tanh), EPI_RESID_F32, st);
    return 0;
  }
  GemmArgs g = gemm_args(a, lda, w, nullptr, e->eslabs, d, rows, d, k, e->cfg.gelu_tanh);
  g.k_split = s;
  g.split_stride = (long)rows * d;
  launch_gemm_big(g, EPI_F32, st);
  return s;
}
}  // namespace

extern "C" {

// ------------------------------------------------------------------------------ encoder
int stt_encode(stt_engine* e, const float* d_mel, int batch, int frames, float* d_enc_out,
               void* stream_) {
  hipStream_t st = (hipStream_t)stream_;
  if (!e || !d_mel) return stt_fail(STT_ERR_INVALID_ARGUMENT, "null argument");
  if (!e->finalized) return stt_fail(STT_ERR_INVALID_ARGUMENT, "engine not finalized");
  if (!e->buffers_ok) return stt_fail(STT_ERR_INVALID_ARGUMENT, "engine holds no buffers (an allocation failed)");
  if (batch < 1 || batch > e->max_batch || frames < 1 || frames > 3000)
    return stt_fail(STT_ERR_INVALID_ARGUMENT, "batch %d (max %d) / frames %d (max 3000) out of range",
                    batch, e->max_batch, frames);
  const stt_model_config& c = e->cfg;
  const int d = c.n_audio_state, H = c.n_audio_head, T = 1500;
  const int rows = batch * T;
  const int gt = c.gelu_tanh;
  // Test hook: STT_MI355X_ATTN_LIFT in (0, 100] makes the attention kernel move its softmax
  // reference more (or less) eagerly than the production setting, so that the rarely taken branch
  // can be exercised on ordinary inputs.  The result does not depend on it beyond rounding.
  float lift_at = 8.0f;
  if (const char* v = getenv("STT_MI355X_ATTN_LIFT")) {
    const float f = (float)atof(v);
    if (f > 0.f && f <= 100.f) lift_at = f;
  }

  // The attention kernel's last key tile of the last window reads up to 64 positions past the
  // batch in every row of the V^T image (under zero weights).  In a smaller batch than an earlier
  // one those positions hold that call's data: clear them, so that a non-finite value left there
  // cannot reach this call through 0 * NaN.
  STT_HIP_OR_FAIL(hipMemset2DAsync(e->evt + (size_t)batch * T, (size_t)vt_pitch(e) * sizeof(bf16_t), 0,
                                   64 * sizeof(bf16_t), (size_t)H * 64, st), (void)0);
  launch_pack_mel(d_mel, batch, frames, c.n_mels, e->mel_pad, st);
  {   // conv1: rows (b, t) start at mel_pad[b][t], K = 3 * n_mels (zero-padded weights beyond)
    GemmArgs g = gemm_args(e->mel_pad, c.n_mels, e->conv1_w, e->conv1_b, e->h1_pad + d, d, 3000, d,
                           e->conv1_k, gt);
    g.batch = batch; g.a_batch_stride = 3002L * c.n_mels; g.out_batch_stride = 3002L * d;
    launch_gemm_big(g, EPI_GELU_BF16, st);
  }
  {   // conv2, stride 2: row (b, t) starts at h1_pad[b][2 t]; + sinusoids
    GemmArgs g = gemm_args(e->h1_pad, 2L * d, e->conv2_w, e->conv2_b, e->ex, d, T, d, 3 * d, gt);
    g.batch = batch; g.a_batch_stride = 3002L * d; g.out_batch_stride = (long)T * d;
    g.pos = e->enc_pos; g.pos_ld = d;
    launch_gemm_big(g, EPI_GELU_POS_F32, st);
  }
  int pend = 0;                              // slabs of the last residual projection not yet added to ex
  const float* pend_bias = nullptr;
  const long slab_stride = (long)rows * d;
  for (int l = 0; l < c.n_audio_layer; ++l) {
    const LayerW& L = e->enc[l];
    launch_add_layernorm(e->ex, e->eslabs, pend, slab_stride, pend ? pend_bias : nullptr, L.ln1g, L.ln1b, e->eh,
                         nullptr, rows, d, 1e-5f, 1, 0, st);
    {   // q (scaled for the base-2 softmax) | k into eqkv, v transposed into evt
      GemmArgs gq = gemm_args(e->eh, d, L.wqkv, L.bqkv, e->eqkv, 3 * d, rows, 3 * d, d, gt);
      gq.out2 = e->evt; gq.vt_pitch = vt_pitch(e);
      gq.q_scale = 0.125f * 1.4426950408889634f;
      launch_gemm_big(gq, EPI_QKV_ENC, st);
    }
    launch_attn_encoder(e->eqkv, e->eqkv + d, 3L * d, (long)T * 3 * d, e->evt, vt_pitch(e), e->eatt, batch,
                        T, H, lift_at, st);
    pend = residual_projection(e, e->eatt, d, L.wo, L.bo, rows, batch, d, st);
    pend_bias = L.bo;
    launch_add_layernorm(e->ex, e->eslabs, pend, slab_stride, pend ? pend_bias : nullptr, L.ln2g, L.ln2b, e->eh,
                         nullptr, rows, d, 1e-5f, 1, 0, st);
    launch_gemm_big(gemm_args(e->eh, d, L.w1, L.b1, e->ef, 4 * d, rows, 4 * d, d, gt), EPI_GELU_BF16, st);
    pend = residual_projection(e, e->ef, 4 * d, L.w2, L.b2, rows, batch, 4 * d, st);
    pend_bias = L.b2;
  }
  launch_add_layernorm(e->ex, e->eslabs, pend, slab_stride, pend ? pend_bias : nullptr, e->enc_lng, e->enc_lnb,
                       e->enc_out, d_enc_out, rows, d, 1e-5f, 1, 0, st);
  // cross-attention K/V of every decoder layer, once per window
  for (int l = 0; l < c.n_text_layer; ++l) {
    const LayerW& L = e->dec[l];
    // K | V projection written straight into the head-major images the decoder reads
    GemmArgs gkv = gemm_args(e->enc_out, d, L.wkv_c, L.bkv_c, e->cross_k[l], 2 * d, rows, 2 * d, d, gt);
    gkv.out2 = e->cross_v[l];
    gkv.kv_t = T;
    launch_gemm_big(gkv, EPI_KV_HEADS, st);
  }
  e->enc_batch = batch;
  e->step_len = e->step_seqs = 0;
  STT_HIP_OR_FAIL(hipGetLastError(), (void)0);
  return STT_OK;
}

}  // extern "C"

namespace {

// One decoder pass over n_seq sequences x s_new new tokens (tokens_in / positions already on the
// device).  last_only: logits of the last new token of every sequence -> gs.logits [n_seq][V];
// otherwise logits of every row -> d_all_logits [rows][V].
// finished: null, or the per-sequence flags of a generation loop: the attention kernels skip
// sequences (self-attention) and windows (cross-attention) that have ended.
int decoder_pass(stt_engine* e, int n_seq, int seq_per_win, int s_new, bool last_only,
                 float* d_all_logits, hipStream_t st, float* d_last_logits = nullptr,
                 const int* finished = nullptr) {
  float* last_dst = d_last_logits ? d_last_logits : e->gs.logits;
  const stt_model_config& c = e->cfg;
  const int d = c.n_text_state, H = c.n_text_head, V = c.n_vocab;
  const int rows = n_seq * s_new;
  const long slab_stride = (long)rows * d;
  launch_embed(e->tokens_in, e->positions, e->tok_emb, e->pos_emb, e->dx, rows, d, V, c.n_text_ctx, st);
  int pend = 0;
  const float* pend_bias = nullptr;
  static const bool no_fuse = getenv("STT_MI355X_NO_LNFUSE") != nullptr;
  const bool fuse = rows <= kLnFuseRows && d <= 2048 && !no_fuse;
  float *xcur = e->dx, *xalt = e->dx2;       // fused path: the residual stream alternates between the two
  const int gt = e->cfg.gelu_tanh;
  // LayerNorm (with the pending partial sums) + projection in one launch
  auto ln_dense = [&](const float* gamma, const float* beta, const bf16_t* w, const float* bias, int n,
                      int epilogue, void* out, long ldc, int split, float* slabs_out) {
    LnFuse f{xcur, xalt, e->slabs, pend, slab_stride, pend ? pend_bias : nullptr, gamma, beta, 1e-5f, 1, 0};
    GemmArgs g = gemm_args(nullptr, d, w, bias, out, ldc, rows, n, d, gt);
    launch_gemm_skinny_ln(g, f, epilogue, split, slabs_out, st);
    if (pend) { float* t = xcur; xcur = xalt; xalt = t; }
    pend = 0;
  };
  for (int l = 0; l < c.n_text_layer; ++l) {
    const LayerW& L = e->dec[l];
    if (fuse) {
      ln_dense(L.ln1g, L.ln1b, L.wqkv, L.bqkv, 3 * d, EPI_BF16, e->dqkv, 3 * d, 1, nullptr);
    } else {
      launch_add_layernorm(e->dx, e->slabs, pend, slab_stride, pend_bias, L.ln1g, L.ln1b, e->dh, nullptr,
                           rows, d, 1e-5f, 1, 0, st);
      dense(e, e->dh, d, L.wqkv, L.bqkv, rows, 3 * d, d, EPI_BF16, e->dqkv, 3 * d, nullptr, st);
    }
    SelfAttnArgs sa{e->dqkv, e->self_k[l], e->self_v[l], e->gs.slot_of, e->gs.base_len, e->datt,
                    n_seq, s_new, H, e->max_len, finished, nullptr, 0};
    if (e->attn_self_out && l >= e->attn_first_layer) {
      sa.probs = e->attn_self_out + (long)(l - e->attn_first_layer) * rows * H * e->attn_len;
      sa.probs_len = e->attn_len;
    }
    launch_attn_decoder_self(sa, st);
    dense(e, e->datt, d, L.wo, L.bo, rows, d, d, EPI_RESID_F32, e->dx, d, &pend, st);
    pend_bias = L.bo;
    // decode steps: the query projection is split over K like the residual projections (twice
    // the blocks of an unsplit launch) and the attention kernel sums the slabs as it loads q
    CrossAttnArgs ca{e->dq, nullptr, 0, 0, nullptr, e->cross_k[l], e->cross_v[l], e->datt,
                     n_seq / seq_per_win, seq_per_win, s_new, H, 1500, finished, nullptr};
    if (e->attn_cross_out && l >= e->attn_first_layer)
      ca.probs = e->attn_cross_out + (long)(l - e->attn_first_layer) * rows * H * 1500;
    if (fuse) {
      const int sq = pick_split_k(rows, d, d, e->n_cu);
      ln_dense(L.lncg, L.lncb, L.wq_c, nullptr, d, -1, nullptr, d, sq, e->slabs2);
      ca.q_slabs = e->slabs2; ca.n_q_slabs = sq; ca.q_slab_stride = slab_stride; ca.q_bias = L.bq_c;
    } else {
      launch_add_layernorm(e->dx, e->slabs, pend, slab_stride, pend_bias, L.lncg, L.lncb, e->dh, nullptr,
                           rows, d, 1e-5f, 1, 0, st);
      if (rows <= kDecodeRows) {
        GemmArgs gq = gemm_args(e->dh, d, L.wq_c, nullptr, nullptr, d, rows, d, d, e->cfg.gelu_tanh);
        const int sq = pick_split_k(rows, d, d, e->n_cu);
        launch_gemm_skinny(gq, -1, sq, e->slabs, st);
        ca.q_slabs = e->slabs; ca.n_q_slabs = sq; ca.q_slab_stride = slab_stride; ca.q_bias = L.bq_c;
      } else {
        dense(e, e->dh, d, L.wq_c, L.bq_c, rows, d, d, EPI_BF16, e->dq, d, nullptr, st);
      }
    }
    launch_attn_decoder_cross(ca, st);
    dense(e, e->datt, d, L.wo_c, L.bo_c, rows, d, d, EPI_RESID_F32, e->dx, d, &pend, st);
    pend_bias = L.bo_c;
    if (fuse) {
      ln_dense(L.ln2g, L.ln2b, L.w1, L.b1, 4 * d, EPI_GELU_BF16, e->df, 4 * d, 1, nullptr);
    } else {
      launch_add_layernorm(e->dx, e->slabs, pend, slab_stride, pend_bias, L.ln2g, L.ln2b, e->dh, nullptr,
                           rows, d, 1e-5f, 1, 0, st);
      dense(e, e->dh, d, L.w1, L.b1, rows, 4 * d, d, EPI_GELU_BF16, e->df, 4 * d, nullptr, st);
    }
    dense(e, e->df, 4 * d, L.w2, L.b2, rows, d, 4 * d, EPI_RESID_F32, e->dx, d, &pend, st);
    pend_bias = L.b2;
  }
  if (fuse) {
    // final LayerNorm + tied vocabulary projection; the residual stream is not needed afterwards
    const int out_rows = last_only ? n_seq : rows;
    LnFuse f{xcur, nullptr, e->slabs, pend, slab_stride, pend ? pend_bias : nullptr, e->dec_lng, e->dec_lnb, 1e-5f,
             last_only ? s_new : 1, last_only ? s_new - 1 : 0};
    GemmArgs g = gemm_args(nullptr, d, e->tok_emb, nullptr, last_only ? (void*)last_dst : (void*)d_all_logits, V,
                           out_rows, V, d, gt);
    launch_gemm_skinny_ln(g, f, EPI_F32, 1, nullptr, st);
  } else if (last_only) {
    launch_add_layernorm(e->dx, e->slabs, pend, slab_stride, pend_bias, e->dec_lng, e->dec_lnb, e->dh,
                         nullptr, n_seq, d, 1e-5f, s_new, s_new - 1, st);
    dense(e, e->dh, d, e->tok_emb, nullptr, n_seq, V, d, EPI_F32, last_dst, V, nullptr, st);
  } else {
    launch_add_layernorm(e->dx, e->slabs, pend, slab_stride, pend_bias, e->dec_lng, e->dec_lnb, e->dh,
                         nullptr, rows, d, 1e-5f, 1, 0, st);
    dense(e, e->dh, d, e->tok_emb, nullptr, rows, V, d, EPI_F32, d_all_logits, V, nullptr, st);
  }
  return STT_OK;
}

__global__ void init_state_kernel(GenState s, int n_seq, int prompt_len, int max_len) {
  const int i = blockIdx.x;
  if (threadIdx.x == 0) {
    s.lengths[i] = prompt_len; s.scores[i] = 0.f; s.finished[i] = 0; s.next_token[i] = 0;
    s.base_len[i] = 0;
    if (i == 0) { *s.step = 0; *s.all_finished = 0; }
  }
  for (int p = threadIdx.x; p < max_len; p += blockDim.x) s.slot_of[(long)i * max_len + p] = i;
}
// stt_decode_step: positions of the new tokens, and the cache bookkeeping around the pass
__global__ void step_positions_kernel(const int* base_len, int* positions, int n_seq, int n_tokens) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r < n_seq * n_tokens) positions[r] = base_len[r / n_tokens] + r % n_tokens;
}
__global__ void advance_len_kernel(int* base_len, int n_seq, int n_tokens) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n_seq) base_len[i] += n_tokens;
}
// Sequence i continues the history of sequence parents[i]: positions the parent's history covers
// keep the cache slots the parent read them from, later ones will be written by i itself.
__global__ void relink_kernel(GenState s, const int* parents, int n_seq, int max_len) {
  const int i = blockIdx.x;
  int p = parents[i];
  if (p < 0 || p >= n_seq) p = i;
  const int len = s.base_len[p];
  for (int q = threadIdx.x; q < max_len; q += blockDim.x)
    s.slot_of_tmp[(long)i * max_len + q] = q < len ? s.slot_of[(long)p * max_len + q] : i;
  if (threadIdx.x == 0) s.lengths_tmp[i] = len;
}
__global__ void relink_commit_kernel(GenState s, int max_len) {
  const int i = blockIdx.x;
  for (int q = threadIdx.x; q < max_len; q += blockDim.x)
    s.slot_of[(long)i * max_len + q] = s.slot_of_tmp[(long)i * max_len + q];
  if (threadIdx.x == 0) s.base_len[i] = s.lengths_tmp[i];
}
// [n][heads][length][64] fp32 out of a cache [slot][heads][pitch][64] bf16; slot_of (or null: slot = n)
__global__ void export_cache_kernel(const bf16_t* cache, const int* slot_of, int slot_of_pitch, float* out,
                                    int heads, int pitch, int length) {
  const int i = blockIdx.y, h = blockIdx.x;
  for (long e = threadIdx.x; e < (long)length * 64; e += blockDim.x) {
    const int p = (int)(e >> 6);
    const int slot = slot_of ? slot_of[(long)i * slot_of_pitch + p] : i;
    out[(((long)i * heads + h) * length) * 64 + e] =
        __uint_as_float((uint32_t)cache[(((long)slot * heads + h) * pitch) * 64 + e] << 16);
  }
}

__global__ void fill_int_kernel(int* p, long n, int v) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) p[i] = v;
}

}  // namespace

extern "C" {

int stt_decoder_forward(stt_engine* e, const int32_t* tokens, int batch, int n_tokens,
                        float* d_logits, void* stream_) {
  hipStream_t st = (hipStream_t)stream_;
  if (!e || !tokens || !d_logits) return stt_fail(STT_ERR_INVALID_ARGUMENT, "null argument");
  if (!e->finalized) return stt_fail(STT_ERR_INVALID_ARGUMENT, "engine not finalized");
  if (!e->buffers_ok) return stt_fail(STT_ERR_INVALID_ARGUMENT, "engine holds no buffers (an allocation failed)");
  if (batch != e->enc_batch)
    return stt_fail(STT_ERR_INVALID_ARGUMENT, "batch %d differs from the encoded batch %d", batch, e->enc_batch);
  if (n_tokens < 1 || n_tokens > e->max_len || batch * n_tokens > e->max_rows)
    return stt_fail(STT_ERR_INVALID_ARGUMENT, "%d tokens x %d sequences exceed the engine capacity", n_tokens, batch);
  std::vector<int> pos((size_t)batch * n_tokens);
  for (int b = 0; b < batch; ++b) for (int j = 0; j < n_tokens; ++j) pos[(size_t)b * n_tokens + j] = j;
  STT_HIP_OR_FAIL(hipMemcpyAsync(e->tokens_in, tokens, pos.size() * 4, hipMemcpyHostToDevice, st), (void)0);
  STT_HIP_OR_FAIL(hipMemcpyAsync(e->positions, pos.data(), pos.size() * 4, hipMemcpyHostToDevice, st), (void)0);
  hipLaunchKernelGGL(init_state_kernel, dim3(batch), dim3(64), 0, st, e->gs, batch, n_tokens, e->max_len);
  int rc = decoder_pass(e, batch, 1, n_tokens, false, d_logits, st);
  if (rc) return rc;
  STT_HIP_OR_FAIL(hipStreamSynchronize(st), (void)0);     // `pos` is a host temporary
  return STT_OK;
}

int stt_decode_step_attention(stt_engine* e, const int32_t* d_tokens, int n_seq, int seq_per_window, int n_tokens,
                              int reset, const int32_t* d_parents, float* d_logits, int first_layer,
                              float* d_self_attention, float* d_cross_attention, void* stream_) {
  if (!e) return stt_fail(STT_ERR_INVALID_ARGUMENT, "null argument");
  if (first_layer < 0 || first_layer >= e->cfg.n_text_layer)
    return stt_fail(STT_ERR_INVALID_ARGUMENT, "first_layer %d of %d decoder layers", first_layer, e->cfg.n_text_layer);
  e->attn_self_out = d_self_attention;
  e->attn_cross_out = d_cross_attention;
  e->attn_first_layer = first_layer;
  e->attn_len = (reset ? 0 : e->step_len) + n_tokens;
  const int rc = stt_decode_step(e, d_tokens, n_seq, seq_per_window, n_tokens, reset, d_parents, d_logits, stream_);
  e->attn_self_out = e->attn_cross_out = nullptr;
  return rc;
}

int stt_engine_export_state(stt_engine* e, int layer, int cross, float* d_k, float* d_v, void* stream_) {
  hipStream_t st = (hipStream_t)stream_;
  if (!e || !d_k || !d_v) return stt_fail(STT_ERR_INVALID_ARGUMENT, "null argument");
  if (!e->finalized || !e->buffers_ok) return stt_fail(STT_ERR_INVALID_ARGUMENT, "engine not finalized");
  if (layer < 0 || layer >= e->cfg.n_text_layer) return stt_fail(STT_ERR_INVALID_ARGUMENT, "no such layer");
  const int H = e->cfg.n_text_head;
  if (cross) {
    if (e->enc_batch < 1) return stt_fail(STT_ERR_INVALID_ARGUMENT, "nothing encoded");
    const dim3 grid(H, e->enc_batch);
    hipLaunchKernelGGL(export_cache_kernel, grid, dim3(256), 0, st, e->cross_k[layer], (const int*)nullptr, 0, d_k,
                       H, 1500, 1500);
    hipLaunchKernelGGL(export_cache_kernel, grid, dim3(256), 0, st, e->cross_v[layer], (const int*)nullptr, 0, d_v,
                       H, 1500, 1500);
  } else {
    if (e->step_len < 1 || e->step_seqs < 1)
      return stt_fail(STT_ERR_INVALID_ARGUMENT, "no stt_decode_step history to export");
    const dim3 grid(H, e->step_seqs);
    hipLaunchKernelGGL(export_cache_kernel, grid, dim3(256), 0, st, e->self_k[layer], e->gs.slot_of, e->max_len, d_k,
                       H, e->max_len, e->step_len);
    hipLaunchKernelGGL(export_cache_kernel, grid, dim3(256), 0, st, e->self_v[layer], e->gs.slot_of, e->max_len, d_v,
                       H, e->max_len, e->step_len);
  }
  STT_HIP_OR_FAIL(hipGetLastError(), (void)0);
  return STT_OK;
}

int stt_engine_step_state(const stt_engine* e, int* n_seq, int* seq_per_window, int* length) {
  if (!e) return stt_fail(STT_ERR_INVALID_ARGUMENT, "null argument");
  if (n_seq) *n_seq = e->step_seqs;
  if (seq_per_window) *seq_per_window = e->step_per_window;
  if (length) *length = e->step_len;
  return STT_OK;
}

int stt_decode_step(stt_engine* e, const int32_t* d_tokens, int n_seq, int seq_per_window, int n_tokens,
                    int reset, const int32_t* d_parents, float* d_logits, void* stream_) {
  hipStream_t st = (hipStream_t)stream_;
  if (!e || !d_tokens || !d_logits) return stt_fail(STT_ERR_INVALID_ARGUMENT, "null argument");
  if (!e->finalized) return stt_fail(STT_ERR_INVALID_ARGUMENT, "engine not finalized");
  if (!e->buffers_ok) return stt_fail(STT_ERR_INVALID_ARGUMENT, "engine holds no buffers (an allocation failed)");
  if (seq_per_window < 1 || seq_per_window > e->max_beams || n_seq < 1 || n_seq % seq_per_window ||
      n_seq / seq_per_window != e->enc_batch)
    return stt_fail(STT_ERR_INVALID_ARGUMENT, "%d sequences at %d per window do not match the encoded batch %d "
                    "(at most %d per window)", n_seq, seq_per_window, e->enc_batch, e->max_beams);
  if (reset) e->step_len = 0;
  else if (e->step_seqs != n_seq)
    return stt_fail(STT_ERR_INVALID_ARGUMENT, "%d sequences, but the history was started with %d (reset first)",
                    n_seq, e->step_seqs);
  if (n_tokens < 1 || e->step_len + n_tokens > e->max_len || (long)n_seq * n_tokens > e->max_rows)
    return stt_fail(STT_ERR_INVALID_ARGUMENT, "%d new tokens after %d exceed the engine capacity (%d positions, "
                    "%d rows per pass)", n_tokens, e->step_len, e->max_len, e->max_rows);
  if (reset) {
    hipLaunchKernelGGL(init_state_kernel, dim3(n_seq), dim3(64), 0, st, e->gs, n_seq, 0, e->max_len);
  } else if (d_parents) {
    hipLaunchKernelGGL(relink_kernel, dim3(n_seq), dim3(256), 0, st, e->gs, d_parents, n_seq, e->max_len);
    hipLaunchKernelGGL(relink_commit_kernel, dim3(n_seq), dim3(256), 0, st, e->gs, e->max_len);
  }
  const int rows = n_seq * n_tokens;
  STT_HIP_OR_FAIL(hipMemcpyAsync(e->tokens_in, d_tokens, (size_t)rows * 4, hipMemcpyDeviceToDevice, st), (void)0);
  hipLaunchKernelGGL(step_positions_kernel, dim3((rows + 255) / 256), dim3(256), 0, st, e->gs.base_len,
                     e->positions, n_seq, n_tokens);
  int rc = decoder_pass(e, n_seq, seq_per_window, n_tokens, true, nullptr, st, d_logits);
  if (rc) return rc;
  hipLaunchKernelGGL(advance_len_kernel, dim3((n_seq + 255) / 256), dim3(256), 0, st, e->gs.base_len, n_seq,
                     n_tokens);
  e->step_len += n_tokens;
  e->step_seqs = n_seq;
  e->step_per_window = seq_per_window;
  STT_HIP_OR_FAIL(hipGetLastError(), (void)0);
  return STT_OK;
}

int stt_engine_probe_kernel(stt_engine* e, int kernel, int layer, void* stream_) {
  hipStream_t st = (hipStream_t)stream_;
  if (!e || !e->finalized || !e->buffers_ok) return stt_fail(STT_ERR_INVALID_ARGUMENT, "engine not finalized");
  if (e->enc_batch < 1) return stt_fail(STT_ERR_INVALID_ARGUMENT, "nothing encoded");
  if (layer < 0 || layer >= e->cfg.n_text_layer) return stt_fail(STT_ERR_INVALID_ARGUMENT, "no such layer");
  const int d = e->cfg.n_text_state, H = e->cfg.n_text_head;
  if (kernel == STT_PROBE_CROSS_ATTENTION || (kernel & ~0xFF) == 0x100) {
    // exactly the launch of a token step: one new token per sequence, `per` sequences per window
    // (1 = greedy, more = beam search), the query read as split-K partial sums (whatever the last
    // step left in them: timing only)
    const int per = kernel == STT_PROBE_CROSS_ATTENTION ? 1 : (kernel & 0xFF);
    if (per < 1 || per > e->max_beams) return stt_fail(STT_ERR_INVALID_ARGUMENT, "%d sequences per window (max %d)", per, e->max_beams);
    const int rows = e->enc_batch * per;
    const LayerW& L = e->dec[layer];
    CrossAttnArgs ca{e->dq, nullptr, 0, 0, nullptr, e->cross_k[layer], e->cross_v[layer], e->datt,
                     e->enc_batch, per, 1, H, 1500, nullptr};
    if (rows <= kDecodeRows) {
      ca.q_slabs = e->slabs; ca.n_q_slabs = pick_split_k(rows, d, d, e->n_cu);
      ca.q_slab_stride = (long)rows * d; ca.q_bias = L.bq_c;
    }
    launch_attn_decoder_cross(ca, st);
  } else {
    return stt_fail(STT_ERR_UNSUPPORTED, "unknown probe %d", kernel);
  }
  STT_HIP_OR_FAIL(hipGetLastError(), (void)0);
  return STT_OK;
}

int stt_generate(stt_engine* e, const stt_generate_params* p, const int32_t* prompt, int batch,
                 int prompt_len, int32_t* out_tokens, int32_t* out_lengths, float* out_scores,
                 void* stream_) {
  hipStream_t st = (hipStream_t)stream_;
  if (!e || !p || !prompt || !out_tokens || !out_lengths || !out_scores)
    return stt_fail(STT_ERR_INVALID_ARGUMENT, "null argument");
  if (!e->finalized) return stt_fail(STT_ERR_INVALID_ARGUMENT, "engine not finalized");
  if (!e->buffers_ok) return stt_fail(STT_ERR_INVALID_ARGUMENT, "engine holds no buffers (an allocation failed)");
  if (batch != e->enc_batch)
    return stt_fail(STT_ERR_INVALID_ARGUMENT, "batch %d differs from the encoded batch %d", batch, e->enc_batch);
  const int beams = p->method == STT_METHOD_BEAM ? p->num_beams : 1;
  const int nsent = p->method == STT_METHOD_BEAM ? p->num_sentences : 1;
  if (p->method != STT_METHOD_GREEDY && p->method != STT_METHOD_BEAM)
    return stt_fail(STT_ERR_UNSUPPORTED, "method %d: only greedy (0) and beam search (1) are implemented", p->method);
  if (beams < 1 || beams > e->max_beams || nsent < 1 || nsent > beams)
    return stt_fail(STT_ERR_INVALID_ARGUMENT, "beams %d (max %d) / sentences %d out of range", beams, e->max_beams, nsent);
  if (prompt_len < 1 || prompt_len > e->max_prompt || p->max_steps < 1 ||
      prompt_len + p->max_steps > e->max_len)
    return stt_fail(STT_ERR_INVALID_ARGUMENT, "prompt %d + %d new tokens exceed the engine capacity (%d / %d)",
                    prompt_len, p->max_steps, e->max_prompt, e->max_len);
  if (p->n_suppress > e->suppress_cap || p->n_suppress_first > e->suppress_cap)
    return stt_fail(STT_ERR_INVALID_ARGUMENT, "too many suppressed tokens");
  const int n_seq = batch * beams, V = e->cfg.n_vocab, max_steps = p->max_steps;
  e->step_len = e->step_seqs = 0;          // the cache is taken over: a stt_decode_step history ends here
  if (p->d_logits_out && p->method != STT_METHOD_GREEDY)
    return stt_fail(STT_ERR_UNSUPPORTED, "d_logits_out (return_logits) is implemented for greedy search only");
  GenState gs = e->gs;
  gs.logits_out = p->d_logits_out;
  gs.step_cap = nullptr;
  if (p->max_steps_per_window) {
    for (int b = 0; b < batch; ++b)
      if (p->max_steps_per_window[b] < 1)
        return stt_fail(STT_ERR_INVALID_ARGUMENT, "max_steps_per_window[%d] = %d: must be at least 1", b,
                        p->max_steps_per_window[b]);
    gs.step_cap = e->d_step_cap;
  }
  // Sequences that have ended are skipped by the attention kernels (their cross K/V are the bulk
  // of a token step's traffic).  Only with early stopping: without it the caller asked for every
  // step to be computed as if nothing had ended.
  const int* skip = p->early_stopping ? gs.finished : nullptr;

  // The token step is recorded into a graph (every per-token quantity lives in device memory, so
  // the recorded launches are the same for every token) and replayed.  A stream the caller
  // created can be captured directly, so generation runs there.  Only the legacy default stream
  // cannot: then the work moves to a stream the engine owns, ordered after the caller's by an
  // event.  Worker threads that bring their own streams (two engines, one encoding while the
  // other decodes) therefore add no stream and no cross-stream wait of their own: two workers
  // use two streams, which stays within the four hardware queues a process gets here.  (An
  // earlier version always switched streams, at high priority: with two workers that made five
  // software streams in two priority classes for four hardware queues, the waits between them
  // hung in the queue scheduler, and it was dropped.)
  if (st == nullptr) {
    if (!e->gen_stream) {
      STT_HIP_OR_FAIL(hipStreamCreateWithFlags(&e->gen_stream, hipStreamNonBlocking), (void)0);
      STT_HIP_OR_FAIL(hipEventCreateWithFlags(&e->gen_event, hipEventDisableTiming), (void)0);
    }
    STT_HIP_OR_FAIL(hipEventRecord(e->gen_event, nullptr), (void)0);
    st = e->gen_stream;
    STT_HIP_OR_FAIL(hipStreamWaitEvent(st, e->gen_event, 0), (void)0);
  }

  // prompt repeated for every beam (generation_utils.py:448-451); staged in page-locked memory so
  // the copies are asynchronous and nothing has to be waited for before the first launch (the
  // staging area is free again when the call returns: it ends with a stream synchronisation)
  const size_t n_prompt = (size_t)n_seq * prompt_len;
  int* h_tok = e->pinned;
  int* h_pos = h_tok + e->max_rows;
  int* h_sup = h_pos + e->max_rows;
  int* h_sup1 = h_sup + e->suppress_cap;
  int* h_fin = h_sup1 + e->suppress_cap;         // 64 ints
  int* h_cap = h_fin + 64;                       // [max_batch]
  int* h_gen = h_cap + e->max_batch;             // results: [max_seq * max_len], [max_seq], [max_seq]
  int* h_len = h_gen + (size_t)e->max_seq * e->max_len;
  float* h_sc = (float*)(h_len + e->max_seq);
  for (int i = 0; i < n_seq; ++i)
    for (int j = 0; j < prompt_len; ++j) {
      h_tok[(size_t)i * prompt_len + j] = prompt[(size_t)(i / beams) * prompt_len + j];
      h_pos[(size_t)i * prompt_len + j] = j;
    }
  STT_HIP_OR_FAIL(hipMemcpyAsync(e->tokens_in, h_tok, n_prompt * 4, hipMemcpyHostToDevice, st), (void)0);
  STT_HIP_OR_FAIL(hipMemcpyAsync(e->positions, h_pos, n_prompt * 4, hipMemcpyHostToDevice, st), (void)0);
  if (p->n_suppress) {
    memcpy(h_sup, p->suppress, (size_t)p->n_suppress * 4);
    STT_HIP_OR_FAIL(hipMemcpyAsync(e->d_suppress, h_sup, p->n_suppress * 4, hipMemcpyHostToDevice, st), (void)0);
  }
  if (p->n_suppress_first) {
    memcpy(h_sup1, p->suppress_first, (size_t)p->n_suppress_first * 4);
    STT_HIP_OR_FAIL(hipMemcpyAsync(e->d_suppress_first, h_sup1, p->n_suppress_first * 4,
                                   hipMemcpyHostToDevice, st), (void)0);
  }
  if (gs.step_cap) {
    // greedy: one sequence per window, the limit is read per sequence; beam search reads it per window
    memcpy(h_cap, p->max_steps_per_window, (size_t)batch * 4);
    STT_HIP_OR_FAIL(hipMemcpyAsync(e->d_step_cap, h_cap, (size_t)batch * 4, hipMemcpyHostToDevice, st), (void)0);
  }
  hipLaunchKernelGGL(init_state_kernel, dim3(n_seq), dim3(64), 0, st, gs, n_seq, prompt_len, e->max_len);
  hipLaunchKernelGGL(fill_int_kernel, dim3(64), dim3(256), 0, st, gs.generated, (long)n_seq * max_steps,
                     p->pad_token);

  FilterArgs f{e->d_suppress, p->n_suppress, e->d_suppress_first, p->n_suppress_first,
               p->timestamp_filter, p->timestamp_begin, p->eos_token, p->max_initial_timestamp};
  LogitParams lp{p->temperature, p->length_temperature, p->has_length_temperature,
                 p->sample ? 1 : 0, (unsigned long long)p->seed};
  auto select = [&](int s_new) {
    if (p->method == STT_METHOD_BEAM)
      launch_beam_step(gs, f, lp, batch, beams, nsent, V, max_steps, e->max_len, s_new, p->eos_token,
                       p->pad_token, p->length_power, p->has_length_power, p->early_stopping, st);
    else
      launch_greedy_step(gs, f, lp, n_seq, V, max_steps, s_new, p->eos_token, p->pad_token, st);
  };
  int rc = decoder_pass(e, n_seq, beams, prompt_len, true, nullptr, st, nullptr, skip);
  if (rc) return rc;
  select(prompt_len);

  // One token step = prepare inputs, decoder pass over one new token, selection.  Recorded into a
  // graph the first time a configuration is seen and kept on the engine: every quantity that
  // changes from token to token lives in device memory, everything the launches bake in (sizes,
  // filter and sampling parameters) is part of the key, and the buffers live until
  // stt_engine_reserve / destroy, which drop the cache.
  hipGraphExec_t step_exec = nullptr;
  const char* no_graph = getenv("STT_MI355X_NO_GRAPH");
  // (a caller's logits buffer would be baked into the recorded launches: plain launches then)
  if (max_steps > 1 && !(no_graph && no_graph[0] == '1') && !p->d_logits_out) {
    char key[320];
    snprintf(key, sizeof key, "%d/%d/%d/%d/%d/%d|%d/%d/%d|%a/%d/%a/%d/%a|%d/%d/%d/%d/%d|%d/%llu",
             p->method, batch, beams, nsent, max_steps, gs.step_cap ? 1 : 0, p->eos_token, p->pad_token, p->early_stopping,
             p->temperature, p->has_length_temperature, p->length_temperature, p->has_length_power,
             p->length_power, p->n_suppress, p->n_suppress_first, p->timestamp_filter, p->timestamp_begin,
             p->max_initial_timestamp, p->sample ? 1 : 0, (unsigned long long)(p->sample ? p->seed : 0));
    auto hit = e->step_graphs.find(key);
    if (hit != e->step_graphs.end()) {
      step_exec = hit->second;
    } else {
      hipGraph_t graph = nullptr;
      STT_HIP_OR_FAIL(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal), (void)0);
      launch_prepare_step(gs, e->tokens_in, e->positions, n_seq, st);
      rc = decoder_pass(e, n_seq, beams, 1, true, nullptr, st, nullptr, skip);
      select(1);
      const hipError_t cap = hipStreamEndCapture(st, &graph);
      if (rc) { if (graph) (void)hipGraphDestroy(graph); return rc; }
      STT_HIP_OR_FAIL(cap, (void)0);
      const hipError_t inst = hipGraphInstantiate(&step_exec, graph, nullptr, nullptr, 0);
      (void)hipGraphDestroy(graph);
      STT_HIP_OR_FAIL(inst, (void)0);
      if (e->step_graphs.size() >= 16) drop_step_graphs(e);     // e.g. a fresh seed per call
      e->step_graphs[key] = step_exec;
    }
  }
  const int check_every = p->check_every > 0 ? p->check_every : 8;
  for (int t = 1; t < max_steps; ++t) {
    if (p->early_stopping && (t % check_every) == 0) {
      STT_HIP_OR_FAIL(hipMemcpyAsync(h_fin, gs.all_finished, 4, hipMemcpyDeviceToHost, st), (void)0);
      STT_HIP_OR_FAIL(hipStreamSynchronize(st), (void)0);
      if (*h_fin) break;
    }
    if (step_exec) {
      STT_HIP_OR_FAIL(hipGraphLaunch(step_exec, st), (void)0);
    } else {
      launch_prepare_step(gs, e->tokens_in, e->positions, n_seq, st);
      rc = decoder_pass(e, n_seq, beams, 1, true, nullptr, st, nullptr, skip);
      if (rc) return rc;
      select(1);
    }
  }
  // results: [batch][nsent][max_steps]; lengths exclude the prompt (generation_utils.py:293-300)
  STT_HIP_OR_FAIL(hipMemcpyAsync(h_gen, gs.generated, (size_t)n_seq * max_steps * 4, hipMemcpyDeviceToHost, st), (void)0);
  STT_HIP_OR_FAIL(hipMemcpyAsync(h_len, gs.lengths, (size_t)n_seq * 4, hipMemcpyDeviceToHost, st), (void)0);
  STT_HIP_OR_FAIL(hipMemcpyAsync(h_sc, gs.scores, (size_t)n_seq * 4, hipMemcpyDeviceToHost, st), (void)0);
  STT_HIP_OR_FAIL(hipStreamSynchronize(st), (void)0);
  STT_HIP_OR_FAIL(hipGetLastError(), (void)0);
  for (int b = 0; b < batch; ++b)
    for (int s = 0; s < nsent; ++s) {
      const int i = b * beams + s;
      memcpy(out_tokens + ((size_t)b * nsent + s) * max_steps, h_gen + (size_t)i * max_steps,
             (size_t)max_steps * 4);
      out_lengths[b * nsent + s] = h_len[i] - prompt_len;
      out_scores[b * nsent + s] = h_sc[i];
    }
  return STT_OK;
}

}  // extern "C"
