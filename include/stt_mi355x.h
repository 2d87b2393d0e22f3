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
 * anything else is STT_ERR_UNSUPPORTED.  Allocates a few KB on the current device. */
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

/* The two halves of stt_logmel_run, for a clip whose frames are computed in several
 * pieces (for instance on several GPUs) and share one clip-wide maximum:
 *  stage1 computes frames [frame_begin, frame_begin + frame_count) of a clip of
 *   n_total samples.  d_pcm[0] is sample number sample_origin of the clip and
 *   n_local samples are readable (sample_origin a multiple of 4); they must cover every sample those frames touch
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

#ifdef __cplusplus
}
#endif
#endif /* STT_MI355X_H */
