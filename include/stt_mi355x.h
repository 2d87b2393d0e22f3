/* C ABI of libstt_mi355x.so: the MI355X (gfx950) Whisper inference path.
 *
 * Plain pointers and sizes only.  Pointers named d_* are device memory on the
 * current HIP device (e.g. torch.Tensor.data_ptr()); everything else is host
 * memory.  `stream` is a hipStream_t passed as void* (NULL = default stream).
 * Every function returns STT_OK (0) or a negative STT_ERR_* code;
 * stt_last_error() then returns a thread-local message.  Nothing here
 * synchronises the stream unless stated.
 *
 * The reference (yui-mhcp/speech_to_text) has no native code; each entry point
 * cites the Python function whose arithmetic it replaces.
 */
#ifndef STT_MI355X_H
#define STT_MI355X_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define STT_OK 0
#define STT_ERR_INVALID_ARGUMENT (-1)
#define STT_ERR_UNSUPPORTED (-2)
#define STT_ERR_HIP (-3)
#define STT_ERR_WORKSPACE (-4)

const char* stt_last_error(void);
/* ABI version of the loaded library (bumped on any signature change). */
int stt_abi_version(void);

/* ------------------------------------------------------------------ frontend
 * Log-mel spectrogram of one clip, Whisper settings (filter/win length 400,
 * hop 160).  Replaces WhisperSTFT.mel_spectrogram = STFT.transform + mel
 * projection + log10/clamp/scale (utils/audio/stft.py:238-270, 346-360) and the
 * short-clip zero padding of MelSTFT.__call__ (utils/audio/stft.py:111-112).
 */
typedef struct stt_logmel_plan stt_logmel_plan;

/* window: 400 floats (the reference's periodic Hann, stft.py:223).
 * mel_basis: [201][n_mels] floats, row-major (MelSTFT.mel_basis[0], stft.py:61-68).
 * n_mels must be 80 or 128 and the basis must have the non-zero structure of the
 * Slaney filterbank for sr=16000, n_fft=400, fmin=0, fmax=8000 (any weights);
 * anything else is STT_ERR_UNSUPPORTED.  Allocates a few KB on the current device,
 * among them the counters the kernels of one call synchronise through: calls on one
 * plan must not overlap (use one plan per stream). */
int stt_logmel_plan_create(int n_mels, const float* window, const float* mel_basis,
                           stt_logmel_plan** plan);
void stt_logmel_plan_destroy(stt_logmel_plan* plan);

/* Number of output frames for a clip of n_samples (stft.py:111-112, 348):
 * max(n_samples, 400) / 160. */
int64_t stt_logmel_num_frames(int64_t n_samples);
/* Bytes of device workspace stt_logmel_run needs for a clip of n_samples. */
size_t stt_logmel_workspace_bytes(int64_t n_samples);

/* d_pcm: n_samples float32 (already normalised / trimmed by the caller, as the
 * reference's load_audio does on the host).  d_mel: [num_frames][n_mels] float32.
 * d_pcm must be 16-byte aligned. */
int stt_logmel_run(const stt_logmel_plan* plan, const float* d_pcm, int64_t n_samples,
                   float* d_mel, void* d_workspace, size_t workspace_bytes, void* stream);

/* stt_logmel_run without the workspace argument and without any allocation, for callers that
 * cannot call hipMalloc (code under HIP graph capture, tight serving loops, harnesses that
 * forbid it).  No counterpart in the reference.  The workspace is a 128 KiB array inside the
 * library's code object (one copy per GPU, held until the process ends), which limits the clip
 * to stt_logmel_static_max_samples() samples (about 2.6 h at 16 kHz); longer clips get
 * STT_ERR_WORKSPACE.  The plan must have been created on the current device.  All calls on one
 * GPU share that array: calls that overlap on one GPU race (also across plans and streams). */
int64_t stt_logmel_static_max_samples(void);
int stt_logmel_run_static(const stt_logmel_plan* plan, const float* d_pcm, int64_t n_samples,
                          float* d_mel, void* stream);

/* ---- audio pre-steps on the device (what the reference's read_audio does on the host between
 * reading a file and the mel, utils/audio/audio_io.py:249-264) ---- */
#define STT_PCM_F32 0
#define STT_PCM_I16 1
#define STT_PCM_F64 2

/* Bytes of device workspace stt_pcm_normalize needs (8-byte aligned). */
size_t stt_pcm_workspace_bytes(void);

/* normalize_audio(audio, max_val = 1.) of utils/audio/audio_processing.py:48-60: d_out[i] =
 * float32((d_in[i] - mean) * (1 / max|d_in - mean|)), the clip unchanged apart from the mean when
 * it is flat (peak <= 1e-9).  d_in: n samples of `dtype` (STT_PCM_I16: what wavfile.read / ffmpeg
 * s16le give; reduced and scaled in float64 as numpy does, result bit-identical to the host
 * function.  STT_PCM_F32: float32 arithmetic as numpy's, the mean summed in float64.  STT_PCM_F64:
 * what a resampled clip is (scipy.signal.resample returns float64, audio_processing.py:27-32):
 * mean, peak and scaling in float64, one rounding to float32 at the end, as numpy does; only the
 * order of the mean's additions differs from the host's). */
int stt_pcm_normalize(const void* d_in, int dtype, int64_t n, float* d_out, void* d_workspace,
                      size_t workspace_bytes, void* stream);

/* Per-block peak of x^2 for the RMS silence trimming (utils/audio/audio_processing.py:120-127):
 * d_peaks[b] = max over samples [b * block_size, (b + 1) * block_size) of d_pcm[i]^2 (float32
 * product), b < ceil(n / block_size).  Which blocks are silent and what is cut is decided on the
 * host from these ceil(n / block_size) values (:128-198). */
int stt_pcm_block_peaks(const float* d_pcm, int64_t n, int block_size, float* d_peaks, void* stream);

/* The two halves of stt_logmel_run, for a clip whose frames are computed in several
 * pieces (for instance on several GPUs) and share one clip-wide maximum:
 *  stage1 computes frames [frame_begin, frame_begin + frame_count) of a clip of
 *   n_total samples.  d_pcm[0] is sample number sample_origin of the clip and
 *   n_local samples (at most 2e9) are readable (sample_origin a multiple of 4); they must cover every sample those frames touch
 *   inside the clip ([160*frame_begin - 200, 160*(frame_begin+frame_count-1) + 200)
 *   intersected with [0, n_total)).  Writes z = (log10(max(mel,1e-10)) + 4) / 4 to
 *   d_mel (row 0 = frame_begin) and the piece's maximum of log10(mel) to
 *   *d_piece_max (device float).
 *  stage2 clamps a piece to max(z, ((clip_max - 8) + 4) / 4) given the clip-wide
 *   maximum read from *d_clip_max (device float).
 * d_workspace is the one passed to stage1 (stage2 reads what stage1 recorded). */
int stt_logmel_stage1(const stt_logmel_plan* plan, const float* d_pcm, int64_t sample_origin,
                      int64_t n_local, int64_t n_total, int64_t frame_begin, int64_t frame_count,
                      float* d_mel, float* d_piece_max, void* d_workspace, size_t workspace_bytes,
                      void* stream);
int stt_logmel_stage2(const stt_logmel_plan* plan, float* d_mel, int64_t frame_count,
                      const float* d_clip_max, const void* d_workspace, void* stream);

/* -------------------------------------------------------------------- engine
 * Whisper encoder + KV-cached decoder + greedy / beam generation.  Replaces what
 * the reference reaches through `compiled_infer` (models/interfaces/
 * base_model.py:365-374): Transformer.infer (architectures/transformers/
 * transformer_arch.py:1029-1065) = WhisperEncoder (whisper_arch.py:106-155) once,
 * then generation_utils.infer (architectures/generation_utils.py:57-169).
 * bf16 weights and GEMM inputs, fp32 accumulation, residual stream, softmax and
 * log-softmax.
 */
typedef struct stt_engine stt_engine;

typedef struct {
  int32_t n_mels, n_audio_ctx, n_audio_state, n_audio_head, n_audio_layer;
  int32_t n_vocab, n_text_ctx, n_text_state, n_text_head, n_text_layer;
  int32_t gelu_tanh;     /* 1: tanh GELU (Keras default, what the reference runs); 0: exact erf */
} stt_model_config;

/* Capacity: at most max_batch windows per call, max_beams sequences per window,
 * max_text_len positions per sequence (prompt + generated, <= n_text_ctx, <= 512),
 * max_prompt_len prompt tokens.  Nothing is allocated before stt_engine_finalize. */
int stt_engine_create(const stt_model_config* cfg, int max_batch, int max_beams,
                      int max_text_len, int max_prompt_len, stt_engine** engine);
void stt_engine_destroy(stt_engine* engine);

/* Weights by their OpenAI Whisper state-dict names and layouts (the source format of
 * models/weights_converter.py): float32 host data, Linear weights [out][in], conv
 * weights [out][in][3].  `encoder.positional_embedding` is optional (the reference
 * recomputes the sinusoids, whisper_arch.py:93-119).  Data is copied. */
int stt_engine_set_weight(stt_engine* engine, const char* name, const float* data, int64_t numel);
/* Converts to bf16, uploads, allocates every buffer on the current device. */
int stt_engine_finalize(stt_engine* engine);

/* Grows the capacity given to stt_engine_create (each argument: the new minimum; smaller values
 * keep the current one) without touching the weights: waits for the device, frees and
 * re-allocates the activation, cache and generation buffers only.  The encoded batch is lost
 * (encode again before decoding).  The reference sizes its buffers per call from the prompt
 * and max_length (architectures/generation_utils.py:96-169, 693-725); a caller whose prompt grows
 * from window to window (models/stt/whisper.py:238-243) reserves once for the worst case. */
int stt_engine_reserve(stt_engine* engine, int max_batch, int max_beams, int max_text_len,
                       int max_prompt_len);
/* When the new buffers do not fit the device, stt_engine_reserve returns STT_ERR_HIP and the engine
 * is back at the capacity it had (buffers allocated afresh, encoded batch lost); if even those no
 * longer fit it holds its weights and no buffers, every compute entry point returns
 * STT_ERR_INVALID_ARGUMENT, and a later stt_engine_reserve that fits revives it.
 * stt_engine_usable: 1 when the engine is finalized and holds its buffers, else 0.  (The reference
 * leaves allocation failures to its backend: no counterpart.) */
int stt_engine_usable(const stt_engine* engine);

/* d_mel: [batch][frames][n_mels] float32 on the device, frames <= 3000; shorter
 * windows are padded with 0.0 (BaseSTT.pad_or_trim, models/stt/base_stt.py:227-236).
 * Runs the encoder and the per-layer cross-attention K/V of the decoder
 * (residual_multi_head_attention.py:241-245) and keeps them in the engine.
 * d_enc_out: optional [batch][1500][n_audio_state] float32 copy of the encoder output. */
int stt_encode(stt_engine* engine, const float* d_mel, int batch, int frames, float* d_enc_out,
               void* stream);

/* Teacher-forced decoder pass over the last encoded batch (the `.decoder(tokens,
 * encoder_output=...)` call of Whisper.compiled_detect_language, models/stt/
 * whisper.py:174-186).  tokens: host [batch][n_tokens]; d_logits: device
 * [batch][n_tokens][n_vocab] float32.  Synchronises the stream. */
int stt_decoder_forward(stt_engine* engine, const int32_t* tokens, int batch, int n_tokens,
                        float* d_logits, void* stream);

/* One cached decoder step with the token selection left to the caller: what generation_utils.py
 * calls `step_fn` inside its loop bodies (architectures/generation_utils.py:201-223, 338-358) -
 * the decoder over the new tokens with `initial_state` = the K/V cache, logits of the last
 * position out, K/V of the new tokens appended (:274-277, 693-725).  It is the route for what
 * cannot run inside stt_generate's device loop: an arbitrary Python `logits_filter`
 * (:518-520), per-step logits under beam search (:422-428).
 *  n_seq = windows of the last stt_encode x seq_per_window (sequences of one window are
 *   consecutive and share its cross-attention K/V; seq_per_window <= max_beams).
 *  d_tokens: device int32 [n_seq][n_tokens], the new tokens of every sequence (the prompt on the
 *   first call, one token afterwards; every sequence advances by the same n_tokens).
 *  reset != 0: the sequences start empty.  Otherwise the call continues the history of the
 *   previous stt_decode_step calls (same n_seq); stt_encode and stt_generate end a history.
 *  d_parents: NULL, or device int32 [n_seq]: before the step, sequence i takes over the history
 *   of sequence d_parents[i] (beam search's gather of the decoder state by beam index,
 *   generation_utils.py:658-676; here the cache is re-linked, not copied).
 *  d_logits: device float32 [n_seq][n_vocab], raw logits of the last new position.
 * Enqueues on `stream` and returns; nothing is synchronised. */
int stt_decode_step(stt_engine* engine, const int32_t* d_tokens, int n_seq, int seq_per_window,
                    int n_tokens, int reset, const int32_t* d_parents, float* d_logits, void* stream);

/* stt_decode_step that also returns the attention weights of decoder layers first_layer ..
 * n_text_layer - 1, what the reference collects when its loop runs with `return_attention`
 * (first_layer = 0) or `return_last_attention` (first_layer = n_text_layer - 1):
 * architectures/transformers/transformer_arch.py:697-703 ('attn_<layer>' / 'enc_attn_<layer>'),
 * accumulated over the steps by generation_utils.py:576-610.  Either pointer may be NULL.
 *  d_self_attention: device float32 [layers][n_seq][heads][n_tokens][history + n_tokens], where
 *   history = positions in the cache before this call (0 with reset); entries past a token's own
 *   position are 0 (the causal mask, transformer_arch.py:186-195).
 *  d_cross_attention: device float32 [layers][n_seq][heads][n_tokens][1500].
 * The logits are those of stt_decode_step bit for bit when a window has one or two new rows
 * (token steps of a greedy loop); with 3..16 rows per window (a prompt pass, beams) this call runs
 * the vector-unit cross-attention kernel where the plain one runs the matrix-core kernel: the same
 * sums in another order. */
int stt_decode_step_attention(stt_engine* engine, const int32_t* d_tokens, int n_seq, int seq_per_window,
                              int n_tokens, int reset, const int32_t* d_parents, float* d_logits,
                              int first_layer, float* d_self_attention, float* d_cross_attention,
                              void* stream);

/* The decoder state of one layer as the reference's loop returns it with `return_state`
 * (generation_utils.py:298, 478; per layer ((k, v) of the self-attention, (k, v) of the encoder
 * attention): transformer_arch.py:440-475, residual_multi_head_attention.py:237-245, 313), copied
 * out of the engine as float32 [n][heads][length][64].
 *  cross = 0: the self-attention cache of the stt_decode_step history (n = its sequences, length
 *   = its positions: stt_engine_step_state), each sequence's positions gathered from wherever
 *   beam re-linking left them.  cross = 1: the encoder-attention K/V of the last stt_encode
 *   (n = its windows, length = 1500). */
int stt_engine_export_state(stt_engine* engine, int layer, int cross, float* d_k, float* d_v, void* stream);
/* Sequences, sequences per window and cached positions of the current stt_decode_step history
 * (all 0 / 1 / 0 when there is none).  Any pointer may be NULL. */
int stt_engine_step_state(const stt_engine* engine, int* n_seq, int* seq_per_window, int* length);

/* Measurement aid, no counterpart in the reference: enqueues on `stream` one launch of a hot
 * kernel exactly as a greedy token step over the last encoded batch issues it, so that a
 * benchmark can bracket it with events on its own stream (inside stt_generate the kernel runs
 * in a replayed graph on an engine-owned stream).  Results are discarded. */
#define STT_PROBE_CROSS_ATTENTION 0   /* attn_decoder_cross_kernel of decoder layer `layer` */
#define STT_PROBE_CROSS_ATTENTION_BEAMS(n) (0x100 | (n))   /* ... with n <= max_beams sequences per window */
int stt_engine_probe_kernel(stt_engine* engine, int kernel, int layer, void* stream);

#define STT_METHOD_GREEDY 0
#define STT_METHOD_BEAM 1

typedef struct {
  int32_t method;                 /* STT_METHOD_* (generation_utils.py:826-831) */
  int32_t num_beams, num_sentences;
  int32_t max_steps;              /* new tokens at most (= max_length - prompt length) */
  int32_t eos_token, pad_token;
  int32_t early_stopping;
  float temperature;              /* <= 0: not applied; otherwise scores are divided by it */
  int32_t has_length_temperature;
  float length_temperature;
  int32_t has_length_power;       /* beam ranking by score / (length + 1)^length_power */
  float length_power;
  const int32_t* suppress;        /* host: token ids forced to -inf (logits_filter as an index
                                     list, generation_utils.py:521-522); with timestamp_filter = 1
                                     it is `remove_tokens` (every step but the first) */
  int32_t n_suppress;
  int32_t timestamp_filter;       /* 1: the Whisper rules of models/stt/whisper.py:354-416 */
  const int32_t* suppress_first;  /* host: `remove_tokens_with_space` (first step) */
  int32_t n_suppress_first;
  int32_t timestamp_begin;
  int32_t max_initial_timestamp;
  int32_t check_every;            /* tokens between two host checks of "all finished" (0 = 8) */
  int32_t sample;                 /* 1: the next token (greedy) / the `num_beams` candidates (beam,
                                     with replacement) are drawn from the categorical distribution
                                     of the processed scores, as select_next_token does whenever a
                                     temperature is given (generation_utils.py:546-548); 0: argmax /
                                     top-k.  The reported score is the log-probability of the
                                     token drawn */
  uint64_t seed;                  /* of the draws: same seed, same inputs -> same tokens */
  const int32_t* max_steps_per_window; /* NULL, or host [batch]: window b produces at most that many new tokens
                                     (>= 1; values above max_steps change nothing) and counts as finished after
                                     them, as a call of its own with that max_length would end
                                     (generation_utils.py:794-801): the batched form of the reference's one
                                     max_length per (single-window) call.  Tokens after the limit are pad_token */
  float* d_logits_out;            /* NULL, or device [batch][max_steps][n_vocab] float32 that receives the
                                     processed scores (log-probabilities after temperature, filter and
                                     log-softmax) of every step: InferenceOutput.logits with
                                     return_logits = True (generation_utils.py:266-268, 297, 551-561).
                                     Greedy search only; steps after an early stop are left untouched */
} stt_generate_params;

/* Generation over the last encoded batch.  prompt: host [batch][prompt_len].
 * out_tokens: host [batch][num_sentences][max_steps] (pad_token after the end),
 * out_lengths: [batch][num_sentences] generated tokens without the EOS,
 * out_scores: [batch][num_sentences] summed log-probabilities
 * (InferenceOutput of generation_utils.py:293-300, 466-481).  num_sentences is 1 for
 * greedy.  The work is ordered after what `stream` holds at the call and runs on it (the
 * per-token step is recorded into a graph the first time a configuration is seen, kept on the
 * engine and replayed); the legacy default stream (NULL) cannot be captured, so then the work
 * runs on a stream the engine owns.  It has finished, results in the host buffers, when the
 * call returns.  Calls on one engine must not
 * overlap; two engines may run at the same time from two host threads. */
int stt_generate(stt_engine* engine, const stt_generate_params* params, const int32_t* prompt,
                 int batch, int prompt_len, int32_t* out_tokens, int32_t* out_lengths,
                 float* out_scores, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* STT_MI355X_H */
