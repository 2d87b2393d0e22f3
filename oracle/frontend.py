"""CPU oracle for the Whisper log-mel frontend and its host pre-steps.

TEST INFRASTRUCTURE ONLY.  Nothing in the product package may import this
module: it is the checker that the HIP frontend is compared against (tests,
the smoke check and the cpu_baseline leg of the benchmark).

It restates, in numpy, the algorithm of the reference files
  utils/audio/stft.py:205-232   (windowed Fourier basis, ``STFT.__init__``)
  utils/audio/stft.py:238-270   (``STFT.transform``: reflect pad, strided
                                 conv1d with the basis, magnitude)
  utils/audio/stft.py:59-68     (``librosa.filters.mel`` basis, slaney)
  utils/audio/stft.py:96-120    (``MelSTFT.__call__``: min-length zero pad)
  utils/audio/stft.py:346-360   (``WhisperSTFT.mel_spectrogram``)
  utils/audio/audio_processing.py:48-60    (``normalize_audio``)
  utils/audio/audio_processing.py:98-198   (``trim_silence_rms``)
  utils/audio/audio_io.py:185-269          (``read_audio`` processing order)
  models/stt/base_stt.py:227-236           (``pad_or_trim``)

Parity pin: ``whisper_log_mel(load_audio_like_reference(audio_test.wav))``
reproduces the reference's own golden file
``tests/__reproduction/stft-WhisperSTFT.npy`` (tolerance of the reference's
test: 2e-3, ``tests/test_utils_audio.py:109-111``); see
``tests/test_oracle_frontend.py``.

The mel basis comes from ``librosa.filters.mel`` in the reference, a
dependency that is not vendored and not installed here (version unpinned in
the reference's ``requirements.txt``).  Its published algorithm (Slaney
mel scale, Slaney area normalisation) is restated in ``slaney_mel_basis``
and is pinned indirectly by the golden file above.
"""
import itertools

import numpy as np

N_FFT = 400
HOP = 160
SAMPLE_RATE = 16000


# --------------------------------------------------------------------------
# constants (stft.py:205-232, stft.py:59-68)
# --------------------------------------------------------------------------
def hann_periodic(n=N_FFT):
    """scipy.signal.get_window('hann', n, fftbins=True) (stft.py:223)."""
    return 0.5 - 0.5 * np.cos(2.0 * np.pi * np.arange(n) / n)


def stft_forward_basis(n_fft=N_FFT, window=None):
    """Float32 [2*cutoff, n_fft] windowed Fourier basis (stft.py:205-232).

    Rows 0..cutoff-1 are Re(fft(eye))[k], rows cutoff.. are Im(fft(eye))[k];
    the float32 cast happens before the window multiply, as in the reference
    (``forward_basis *= fft_window`` on a float32 array).
    """
    cutoff = n_fft // 2 + 1
    fourier = np.fft.fft(np.eye(n_fft))
    basis = np.vstack([np.real(fourier[:cutoff]), np.imag(fourier[:cutoff])])
    basis = basis.astype(np.float32)
    if window is None:
        window = hann_periodic(n_fft)
    basis *= window          # float32 *= float64 -> rounded once to float32
    return basis


def _hz_to_mel(f):
    f = np.asanyarray(f, dtype=np.float64)
    f_sp = 200.0 / 3
    mels = f / f_sp
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = np.log(6.4) / 27.0
    with np.errstate(divide="ignore", invalid="ignore"):
        log_part = min_log_mel + np.log(np.maximum(f, 1e-300) / min_log_hz) / logstep
    return np.where(f >= min_log_hz, log_part, mels)


def _mel_to_hz(m):
    m = np.asanyarray(m, dtype=np.float64)
    f_sp = 200.0 / 3
    freqs = f_sp * m
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = np.log(6.4) / 27.0
    return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), freqs)


def slaney_mel_basis(sr=SAMPLE_RATE, n_fft=N_FFT, n_mels=80, fmin=0.0, fmax=8000.0):
    """librosa.filters.mel(sr, n_fft, n_mels, fmin, fmax) -> float32 [n_freq, n_mels].

    The reference stores ``mel_basis.T`` (stft.py:61-68), i.e. [201, n_mels].
    """
    n_freq = 1 + n_fft // 2
    fftfreqs = np.linspace(0.0, sr / 2.0, n_freq)
    mel_pts = np.linspace(_hz_to_mel(fmin), _hz_to_mel(fmax), n_mels + 2)
    mel_f = _mel_to_hz(mel_pts)
    fdiff = np.diff(mel_f)
    ramps = np.subtract.outer(mel_f, fftfreqs)
    weights = np.zeros((n_mels, n_freq), dtype=np.float64)
    for i in range(n_mels):
        lower = -ramps[i] / fdiff[i]
        upper = ramps[i + 2] / fdiff[i + 1]
        weights[i] = np.maximum(0.0, np.minimum(lower, upper))
    enorm = 2.0 / (mel_f[2:n_mels + 2] - mel_f[:n_mels])
    weights *= enorm[:, None]
    return np.ascontiguousarray(weights.T.astype(np.float32))


# --------------------------------------------------------------------------
# frontend (stft.py:96-120, 238-270, 346-360)
# --------------------------------------------------------------------------
def stft_magnitude(audio, n_fft=N_FFT, hop=HOP, basis=None, chunk_frames=65536):
    """``STFT.transform`` magnitude for one clip: float32 [T+1, cutoff].

    The reference runs conv1d(audio[..., None], basis[400, 1, 402], stride=hop,
    'valid'); a strided 'valid' conv1d with one input channel is exactly
    frames @ basis.T, which is what is computed here (in float32, chunked so
    that an hour of audio does not materialise 360k x 400 frames at once).
    """
    audio = np.asarray(audio, dtype=np.float32)
    if basis is None:
        basis = stft_forward_basis(n_fft)
    cutoff = n_fft // 2 + 1
    padded = np.pad(audio, (n_fft // 2, n_fft // 2), mode="reflect")
    n_frames = (len(padded) - n_fft) // hop + 1
    frames = np.lib.stride_tricks.as_strided(
        padded, shape=(n_frames, n_fft),
        strides=(padded.strides[0] * hop, padded.strides[0]), writeable=False)
    out = np.empty((n_frames, cutoff), dtype=np.float32)
    bt = np.ascontiguousarray(basis.T)
    for s in range(0, n_frames, chunk_frames):
        ft = frames[s:s + chunk_frames] @ bt                    # [t, 402] f32
        re, im = ft[:, :cutoff], ft[:, cutoff:]
        out[s:s + chunk_frames] = np.sqrt(re * re + im * im)
    return out


def whisper_log_mel(audio, n_mels=80, mel_basis=None, basis=None, return_raw=False):
    """``WhisperSTFT.__call__`` for a 1-D clip -> float32 [T, n_mels].

    audio is what ``load_audio`` hands over (already normalised / trimmed).
    Magnitude (not power) spectrum, global max-8 clamp over the whole clip,
    (x+4)/4 scaling: stft.py:346-360.
    """
    audio = np.asarray(audio, dtype=np.float32).reshape(-1)
    if len(audio) < N_FFT:                                     # stft.py:111-112
        audio = np.pad(audio, (0, N_FFT - len(audio)))
    if mel_basis is None:
        mel_basis = slaney_mel_basis(n_mels=n_mels)
    mag = np.abs(stft_magnitude(audio, basis=basis)[:-1])      # drop last frame
    mel = mag @ mel_basis.astype(np.float32)
    mel = np.log10(np.maximum(mel, np.float32(1e-10)))
    if return_raw:
        return mel
    mel = np.maximum(mel, mel.max() - np.float32(8.0))
    mel = (mel + np.float32(4.0)) / np.float32(4.0)
    return mel.astype(np.float32)


def whisper_log_mel_f64(audio, n_mels=80):
    """Same chain evaluated in float64 with an FFT (a truth reference used to
    show that a float32 result is no further from exact arithmetic than the
    float32 restatement above is)."""
    audio = np.asarray(audio, dtype=np.float64).reshape(-1)
    if len(audio) < N_FFT:
        audio = np.pad(audio, (0, N_FFT - len(audio)))
    padded = np.pad(audio, (N_FFT // 2, N_FFT // 2), mode="reflect")
    n_frames = (len(padded) - N_FFT) // HOP + 1
    frames = np.lib.stride_tricks.as_strided(
        padded, shape=(n_frames, N_FFT),
        strides=(padded.strides[0] * HOP, padded.strides[0]), writeable=False)
    win = hann_periodic()
    out = np.empty((n_frames - 1, n_mels))
    mb = slaney_mel_basis(n_mels=n_mels).astype(np.float64)
    for s in range(0, n_frames - 1, 65536):
        e = min(n_frames - 1, s + 65536)
        spec = np.abs(np.fft.rfft(frames[s:e] * win, axis=1))
        out[s:e] = spec @ mb
    mel = np.log10(np.maximum(out, 1e-10))
    mel = np.maximum(mel, mel.max() - 8.0)
    return (mel + 4.0) / 4.0


def pad_or_trim(mel, n_frames=3000, pad_value=0.0):
    """``BaseSTT.pad_or_trim`` (base_stt.py:227-236) on a [t, n_mels] mel."""
    mel = np.asarray(mel)
    if mel.shape[0] > n_frames:
        return mel[:n_frames]
    if mel.shape[0] < n_frames:
        return np.pad(mel, ((0, n_frames - mel.shape[0]), (0, 0)), constant_values=pad_value)
    return mel


# --------------------------------------------------------------------------
# host pre-steps (audio_processing.py:48-60, 98-198; audio_io.py:249-264)
# --------------------------------------------------------------------------
def normalize_audio(audio, max_val=1.0):
    """audio_processing.py:48-60 with max_val <= 1 (float32 output)."""
    dtype = np.float32 if max_val <= 1.0 else np.int16
    audio = audio - np.mean(audio)
    peak = np.max(np.abs(audio))
    if peak <= 1e-9:
        return audio.astype(dtype)
    return (audio * (max_val / peak)).astype(dtype)


def trim_silence_rms(audio, rate=SAMPLE_RATE, mode="start_end", threshold=-25,
                     min_silence=0.1, block_size=0.01, replace_by=0.5,
                     min_voice_time=0.2):
    """audio_processing.py:98-198, statement by statement."""
    if isinstance(replace_by, float):
        replace_by = int(replace_by * rate)
    if isinstance(block_size, float):
        block_size = int(block_size * rate)
    block_time = block_size / rate
    amplitude_threshold = 10 ** (threshold / 20.0)

    blocks = audio
    if not np.issubdtype(blocks.dtype, np.floating):
        blocks = blocks / np.iinfo(blocks.dtype).max
    pad = len(blocks) % block_size
    if pad != 0:
        blocks = np.pad(blocks, [(0, block_size - pad)])
    blocks = blocks.reshape(-1, block_size)
    rms = np.sqrt(np.max(blocks ** 2, axis=1))
    silent_blocks = rms < amplitude_threshold

    idx = 0
    starts, ends = [], []
    for is_silent, group in itertools.groupby(silent_blocks):
        n = len(list(group))
        if is_silent and n * block_time >= min_silence:
            starts.append(idx * block_time)
            ends.append(min(len(audio) / rate, (idx + n) * block_time))
        idx += n

    if min_voice_time:
        for i in reversed(range(len(starts) - 1)):
            if starts[i + 1] - ends[i] < min_voice_time:
                starts.pop(i + 1)
                ends.pop(i)

    if mode == "remove":
        replace_half = replace_by // 2
        is_silence = np.zeros((len(audio),), dtype=bool)
        for start, end in zip(starts, ends):
            start, end = int(start * rate), int(end * rate)
            if start == 0:
                is_silence[: max(0, end - replace_by)] = True
            elif abs(end - len(audio)) <= 1:
                is_silence[start + replace_by:] = True
            else:
                start, end = start + replace_half, end - replace_half
                if start < end:
                    is_silence[start:end] = True
        audio = audio[~is_silence]
    else:
        if "end" in mode and starts and abs(ends[-1] * rate - len(audio)) <= 1:
            audio = audio[: int(starts[-1] * rate) + replace_by]
        if "start" in mode and starts and starts[0] == 0:
            audio = audio[max(0, int(ends[0] * rate) - replace_by):]
    return audio


def load_audio_like_reference(pcm, normalize=True, trim_silence=False, rate=SAMPLE_RATE,
                              **trim_kwargs):
    """Processing order of ``read_audio`` (audio_io.py:249-264) for raw PCM at
    the target rate: normalise (mean-subtract, peak -> 1) then optional trim."""
    audio = np.asarray(pcm)
    if normalize is True:
        audio = normalize_audio(audio, max_val=1.0)
    if trim_silence:
        audio = trim_silence_rms(audio, rate=rate, **trim_kwargs)
    return audio


# The processing the Whisper wrapper ends up applying (base_audio_model.py:20-28,
# 121-132 push AudioTrainingHParams into trim_kwargs): normalise, then RMS
# silence removal with these settings.
WHISPER_TRIM_KWARGS = dict(mode="remove", threshold=-25, min_silence=0.1,
                           replace_by=0.4, block_size=0.01, min_voice_time=0.2)


def whisper_prepare_audio(pcm, rate=SAMPLE_RATE):
    return load_audio_like_reference(pcm, normalize=True, trim_silence=True, rate=rate,
                                     **WHISPER_TRIM_KWARGS)


def whisper_log_mel_torch(audio, n_mels=80, threads=None):
    """The same chain with torch-CPU float32 ops, in the reference's own formulation
    (reflect pad -> conv1d with the [402, 1, 400] windowed basis, stride 160 ->
    magnitude -> mel matmul -> log10 / max-8 / (x+4)/4; stft.py:238-270, 346-360).
    Multi-threaded; this is the CPU baseline the benchmark times."""
    import torch
    import torch.nn.functional as Fn
    if threads:
        torch.set_num_threads(int(threads))
    x = torch.as_tensor(np.asarray(audio, dtype=np.float32)).reshape(1, 1, -1)
    if x.shape[-1] < N_FFT:
        x = Fn.pad(x, (0, N_FFT - x.shape[-1]))
    basis = torch.from_numpy(stft_forward_basis()).unsqueeze(1)        # [402, 1, 400]
    mel_basis = torch.from_numpy(slaney_mel_basis(n_mels=n_mels))      # [201, n_mels]
    x = Fn.pad(x, (N_FFT // 2, N_FFT // 2), mode="reflect")
    ft = Fn.conv1d(x, basis, stride=HOP)                               # [1, 402, T+1]
    cutoff = N_FFT // 2 + 1
    re, im = ft[:, :cutoff], ft[:, cutoff:]
    mag = torch.sqrt(re * re + im * im)[:, :, :-1].abs()               # drop last frame
    mel = torch.matmul(mag.transpose(1, 2), mel_basis)                 # [1, T, n_mels]
    mel = torch.log10(torch.clamp(mel, min=1e-10))
    mel = torch.maximum(mel, mel.amax(dim=(1, 2), keepdim=True) - 8.0)
    mel = (mel + 4.0) / 4.0
    return mel[0].numpy()
