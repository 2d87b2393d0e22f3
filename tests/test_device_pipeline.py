"""Host-checkable parts of the device audio pipeline: the Fourier resampling restated on torch.fft
against scipy.signal.resample (which is what the reference calls, utils/audio/audio_processing.py:
27-32), the ffmpeg reader's command line, and the kept-stretch form of the silence removal."""
import os
import stat

import numpy as np
import pytest
import scipy.signal
import torch

from oracle import frontend as oracle


@pytest.mark.parametrize("n,num", [(1000, 363), (1001, 363), (1000, 362), (44100, 16000), (999, 2000),
                                   (1000, 2000), (1000, 2001), (16000, 16000), (7, 3)])
def test_resample_fft_equals_scipy(n, num):
    from speech_to_text_amd.audio.device_pipeline import resample_fft
    rng = np.random.default_rng(n + num)
    x = rng.integers(-30000, 30000, n).astype(np.int16)
    want = scipy.signal.resample(x, num)
    got = resample_fft(torch.from_numpy(x), num).numpy()
    assert got.shape == want.shape and got.dtype == np.float64
    assert np.abs(got - want).max() < 1e-8 * 30000


def test_read_ffmpeg_command_line(tmp_path, monkeypatch):
    """A stand-in `ffmpeg` that records its arguments and emits raw s16le: the reader must ask for
    mono pcm_s16le at the target rate on stdout and return those samples."""
    from speech_to_text_amd.audio import device_pipeline as dp
    raw = np.arange(-100, 100, dtype=np.int16)
    (tmp_path / "samples.raw").write_bytes(raw.tobytes())
    exe = tmp_path / "ffmpeg"
    exe.write_text("#!/bin/sh\necho \"$@\" > %s/args.txt\ncat %s/samples.raw\n" % (tmp_path, tmp_path))
    exe.chmod(exe.stat().st_mode | stat.S_IEXEC)
    monkeypatch.setenv("PATH", str(tmp_path) + os.pathsep + os.environ["PATH"])
    rate, got = dp.read_ffmpeg("clip.mp3", rate=16000)
    assert rate == 16000 and np.array_equal(got, raw)
    args = (tmp_path / "args.txt").read_text().split()
    assert args[args.index("-i") + 1] == "clip.mp3" and args[args.index("-ar") + 1] == "16000"
    assert args[args.index("-f") + 1] == "s16le" and args[args.index("-ac") + 1] == "1" and args[-1] == "-"
    monkeypatch.setenv("PATH", str(tmp_path / "nothing-here"))
    with pytest.raises(RuntimeError, match="no ffmpeg binary"):
        dp.read_ffmpeg("clip.mp3", rate=16000)


def test_kept_stretches_rebuild_the_trimmed_clip():
    from speech_to_text_amd.audio import audio_processing as ap
    rng = np.random.default_rng(5)
    for trial in range(12):
        n = int(rng.integers(2000, 60000))
        x = (rng.standard_normal(n) * 0.3).astype(np.float32)
        for _ in range(int(rng.integers(0, 5))):
            a = int(rng.integers(0, n)); x[a: a + int(rng.integers(500, 9000))] *= 1e-4
        for mode in ("remove", "start", "end", "start_end"):
            kw = dict(rate=16000, mode=mode, threshold=-25, min_silence=0.1, replace_by=0.4)
            keep = ap.kept_stretches(ap.block_peaks(x, 160), n, block_size=160, **kw)
            assert all(0 <= a < b <= n for a, b in keep) and all(keep[i][1] < keep[i + 1][0] for i in range(len(keep) - 1))
            rebuilt = np.concatenate([x[a:b] for a, b in keep]) if keep else x[:0]
            assert np.array_equal(rebuilt, oracle.trim_silence_rms(x, **kw))


def test_reference_resample_fixture(golden_dir):
    """`tests/__reproduction/audio_resample.npy` of the reference = `load_audio(audio_test.wav,
    rate = 22050)` (tests/test_utils_audio.py:62-64): 16 kHz int16 -> scipy Fourier resampling ->
    normalize_audio.  The host path reproduces it exactly; the resampler restated on torch.fft
    (what the device pipeline runs, there through rocFFT) to float32 rounding."""
    from speech_to_text_amd.audio import audio_processing as host
    from speech_to_text_amd.audio.device_pipeline import resample_fft
    from scipy.io import wavfile
    want = np.load(os.path.join(golden_dir, "audio_resample.npy"))
    wav = os.path.join(golden_dir, "audio_test.wav")
    rate, got = host.read_audio(wav, rate=22050, normalize=True)
    assert rate == 22050 and got.dtype == np.float32 and got.shape == want.shape == (89412,)
    assert np.array_equal(got, want)
    _, pcm = wavfile.read(wav)
    mine = host.normalize_audio(resample_fft(torch.from_numpy(pcm), 89412).numpy(), max_val=1.)
    assert np.abs(mine - want).max() < 1e-6
