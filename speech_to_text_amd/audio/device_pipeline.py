"""The wrapper's audio pre-steps with the clip on the GPU: what `read_audio` of the reference does
on the host between a file and the mel (utils/audio/audio_io.py:249-264) - resample, normalise,
remove silence - so that a short clip is decoded, sent over the link once (int16, from a reused
page-locked buffer) and never comes back.

  decode      .wav through scipy (memory-mapped); anything else through the `ffmpeg` binary when
              there is one, asked for mono s16le at the target rate like read_ffmpeg
              (audio_io.py:299-320)
  resample    scipy.signal.resample restated on torch.fft (rocFFT, float64): the one library call of
              this module, an arbitrary-length FFT; only taken when the file's rate differs
  normalise   stt_pcm_normalize (HIP, csrc/audio.hip), bit-identical to normalize_audio for int16
  silence     stt_pcm_block_peaks (HIP) for the per-block peaks, the run-length decisions over the
              (clip / 10 ms) flags on the host with the same code as the host path, the kept
              stretches concatenated on the device

Exactness per path (tests/test_device_pipeline_gpu.py): an int16 clip at the target rate comes out
bit-identical to the host's `read_audio`; a resampled clip is float64 up to the final cast on both
sides and differs from the host's by the rounding of the FFT library and of the mean's summation
order (a few float32 ulps; the reference's own `audio_resample.npy` fixture is met to 1e-6).
"""
import ctypes as C
import shutil
import subprocess
import threading

import numpy as np
import torch

from .. import _lib
from . import audio_processing as host


def resample_fft(x, num):
    """scipy.signal.resample(x, num) for a real 1-D signal (Fourier method: the spectrum is cut or
    zero-padded, the Nyquist bin split or joined), on whatever device `x` lives, in float64."""
    x = torch.as_tensor(x)
    if x.dtype != torch.float64:
        x = x.to(torch.float64)
    nx = x.shape[0]
    spec = torch.fft.rfft(x)
    out = torch.zeros(num // 2 + 1, dtype=spec.dtype, device=x.device)
    n = min(num, nx)
    nyq = n // 2 + 1
    out[:nyq] = spec[:nyq]
    if n % 2 == 0:
        if num < nx:
            out[n // 2] *= 2.0
        elif nx < num:
            out[n // 2] *= 0.5
    y = torch.fft.irfft(out, n=num)
    return y * (float(num) / float(nx))


def read_ffmpeg(filename, rate=None):
    """(rate, int16 samples) through the ffmpeg binary: mono s16le, resampled by ffmpeg when `rate`
    is given (audio_io.py:299-320 does the same through ffmpeg-python)."""
    exe = shutil.which('ffmpeg')
    if exe is None:
        raise RuntimeError('no ffmpeg binary on PATH: only .wav files can be read here ({})'.format(filename))
    cmd = [exe, '-nostdin', '-threads', '0', '-i', filename, '-f', 's16le', '-acodec', 'pcm_s16le', '-ac', '1']
    if rate:
        cmd += ['-ar', str(int(rate))]
    run = subprocess.run(cmd + ['-'], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    if run.returncode != 0:
        raise RuntimeError('Failed to load audio : {}'.format(run.stderr.decode(errors='replace')[-400:]))
    if not rate:
        probe = shutil.which('ffprobe')
        if probe is None:
            raise RuntimeError('ffprobe is needed to learn the sampling rate of {}'.format(filename))
        got = subprocess.run([probe, '-v', 'error', '-select_streams', 'a:0', '-show_entries',
                              'stream=sample_rate', '-of', 'default=nw=1:nk=1', filename],
                             stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        rate = int(got.stdout.decode().strip())
    return int(rate), np.frombuffer(run.stdout, np.int16)


class DeviceAudioPipeline(object):
    """One per device; calls on one pipeline are serialised (it owns the staging buffer)."""

    def __init__(self, device=None, rate=16000):
        self.device = torch.device('cuda', torch.cuda.current_device()) if device is None else torch.device(device)
        self.rate = rate
        self._lib = _lib.load()
        self._ws = torch.empty(self._lib.stt_pcm_workspace_bytes(), dtype=torch.uint8, device=self.device)
        self._pinned = None
        self._lock = threading.Lock()

    # -- host -> device --------------------------------------------------------------------
    def _upload(self, samples):
        """int16 / float32 numpy samples -> device tensor, staged through page-locked memory that
        is allocated once and grows by doubling."""
        samples = np.ascontiguousarray(samples)
        nbytes = samples.nbytes
        if self._pinned is None or self._pinned.numel() < nbytes:
            size = max(nbytes, 2 * (self._pinned.numel() if self._pinned is not None else 1 << 20))
            self._pinned = torch.empty(size, dtype=torch.uint8).pin_memory()
        stage = self._pinned[:nbytes]
        stage.numpy()[:] = samples.view(np.uint8).reshape(-1)
        dev = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
        dev.copy_(stage, non_blocking=True)
        torch.cuda.current_stream(self.device).synchronize()       # the staging buffer is reused
        return dev.view(torch.int16 if samples.dtype == np.int16 else torch.float32)

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    # -- steps ------------------------------------------------------------------------------
    def normalize(self, x):
        """normalize_audio(x, max_val = 1.) for an int16, float32 or float64 device tensor, each in the
        arithmetic numpy uses for that type (a resampled clip is float64 and stays so up to the final
        cast, utils/audio/audio_io.py:249-256)."""
        if x.dtype not in (torch.int16, torch.float32, torch.float64):
            x = x.to(torch.float32)
        x = x.contiguous()
        out = torch.empty(x.shape[0], dtype=torch.float32, device=self.device)
        code = {torch.float32: 0, torch.int16: 1, torch.float64: 2}[x.dtype]
        _lib.check(self._lib.stt_pcm_normalize(
            C.c_void_p(x.data_ptr()), code, x.shape[0], C.c_void_p(out.data_ptr()),
            C.c_void_p(self._ws.data_ptr()), self._ws.numel(), self._stream()))
        return out

    def trim_silence(self, x, *, mode='start_end', threshold=-25, min_silence=0.1, block_size=0.01,
                     replace_by=0.5, min_voice_time=0.2, **_):
        """trim_silence_rms on a float32 device tensor: block peaks on the device, the decisions on
        the host from one float per block, the kept stretches concatenated on the device."""
        rate = self.rate
        if isinstance(replace_by, float):
            replace_by = int(replace_by * rate)
        if isinstance(block_size, float):
            block_size = int(block_size * rate)
        n = x.shape[0]
        n_blocks = -(-n // block_size)
        peaks = torch.empty(n_blocks, dtype=torch.float32, device=self.device)
        _lib.check(self._lib.stt_pcm_block_peaks(C.c_void_p(x.data_ptr()), n, block_size,
                                                 C.c_void_p(peaks.data_ptr()), self._stream()))
        keep = host.kept_stretches(peaks.cpu().numpy(), n, rate=rate, mode=mode, threshold=threshold,
                                   min_silence=min_silence, block_size=block_size, replace_by=replace_by,
                                   min_voice_time=min_voice_time)
        if len(keep) == 1 and keep[0] == (0, n):
            return x
        if not keep:
            return x[:0]
        return torch.cat([x[a:b] for a, b in keep]) if len(keep) > 1 else x[keep[0][0]: keep[0][1]]

    # -- the whole chain ------------------------------------------------------------------------
    def __call__(self, data, *, rate=None, normalize=True, trim_silence=False, **kwargs):
        """`.wav` / media path, or raw samples with their `rate` -> (rate, float32 CUDA tensor)."""
        with self._lock, torch.cuda.device(self.device):
            if isinstance(data, str):
                if data.lower().endswith('.wav'):
                    from scipy.io import wavfile
                    file_rate, samples = wavfile.read(data, mmap=True)
                    if samples.ndim > 1:
                        samples = samples[:, 0]
                else:
                    file_rate, samples = read_ffmpeg(data, rate=self.rate)
            else:
                samples, file_rate = np.asarray(data), rate or self.rate
            if samples.dtype not in (np.int16, np.float32):
                samples = samples.astype(np.float32)
            x = self._upload(samples)
            if file_rate != self.rate:
                x = resample_fft(x, int(len(samples) / file_rate * self.rate))
                if normalize is not True:
                    x = x.to(torch.float32)
            if normalize is True:
                x = self.normalize(x)
            elif normalize and normalize > 1 and x.dtype == torch.int16:
                x = (x.to(torch.float64) / normalize).to(torch.float32)
            elif x.dtype != torch.float32:
                x = x.to(torch.float32)
            if trim_silence:
                x = self.trim_silence(x, **kwargs)
            return self.rate, x
