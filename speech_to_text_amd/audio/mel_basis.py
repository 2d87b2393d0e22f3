"""Slaney mel filterbank and the periodic Hann window, computed on the host.

The reference gets these from ``librosa.filters.mel`` (utils/audio/stft.py:59-68)
and ``scipy.signal.get_window('hann', n, fftbins=True)`` (stft.py:223).  Both
are small closed-form constants; they are built here in float64 and rounded to
float32 once, which is what the reference stores.
"""
import numpy as np


def hann_window(win_length, periodic=True):
    n = np.arange(win_length, dtype=np.float64)
    denom = win_length if periodic else win_length - 1
    return 0.5 - 0.5 * np.cos(2.0 * np.pi * n / denom)


def _hz_to_mel(freq):
    freq = np.asarray(freq, dtype=np.float64)
    lin = freq * 3.0 / 200.0
    log_start_hz, log_start_mel = 1000.0, 15.0
    step = np.log(6.4) / 27.0
    safe = np.maximum(freq, log_start_hz)
    return np.where(freq >= log_start_hz, log_start_mel + np.log(safe / log_start_hz) / step, lin)


def _mel_to_hz(mel):
    mel = np.asarray(mel, dtype=np.float64)
    lin = mel * 200.0 / 3.0
    log_start_hz, log_start_mel = 1000.0, 15.0
    step = np.log(6.4) / 27.0
    return np.where(mel >= log_start_mel, log_start_hz * np.exp(step * (mel - log_start_mel)), lin)


def mel_filterbank(sampling_rate, n_fft, n_mels, fmin=0.0, fmax=None):
    """Triangular, area-normalised (Slaney) filters as float32 [n_fft//2+1, n_mels]."""
    if fmax is None:
        fmax = sampling_rate / 2.0
    n_freq = n_fft // 2 + 1
    bin_hz = np.linspace(0.0, sampling_rate / 2.0, n_freq)
    edges = _mel_to_hz(np.linspace(_hz_to_mel(fmin), _hz_to_mel(fmax), n_mels + 2))
    width = np.diff(edges)
    out = np.zeros((n_freq, n_mels), dtype=np.float64)
    for m in range(n_mels):
        rising = (bin_hz - edges[m]) / width[m]
        falling = (edges[m + 2] - bin_hz) / width[m + 1]
        tri = np.maximum(0.0, np.minimum(rising, falling))
        out[:, m] = tri * (2.0 / (edges[m + 2] - edges[m]))
    return out.astype(np.float32)
