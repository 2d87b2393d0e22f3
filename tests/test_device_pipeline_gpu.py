"""The audio pre-steps run on the GPU (speech_to_text_amd/audio/device_pipeline.py, csrc/audio.hip)
against the host path they replace (audio_processing.read_audio, itself checked against the oracle
in tests/test_oracle_frontend.py)."""
import os

import numpy as np
import pytest
import torch
from scipy.io import wavfile

from oracle import frontend as oracle

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pipe():
    from speech_to_text_amd.audio.device_pipeline import DeviceAudioPipeline
    return DeviceAudioPipeline()


def test_golden_wav_equals_host_path_bit_for_bit(pipe, golden_dir):
    """int16 wav -> normalise -> remove silence: the device result is the host result, and both are
    the oracle's (the notebook's 396-frame clip)."""
    from speech_to_text_amd.audio.audio_processing import WHISPER_TRIM_KWARGS, read_audio
    wav = os.path.join(golden_dir, "audio_test.wav")
    kw = dict(normalize=True, trim_silence=True, **WHISPER_TRIM_KWARGS)
    rate, dev = pipe(wav, **kw)
    _, host = read_audio(wav, rate=16000, **kw)
    assert rate == 16000 and dev.is_cuda and dev.dtype == torch.float32
    assert np.array_equal(dev.cpu().numpy(), host)
    _, pcm = wavfile.read(wav)
    assert np.array_equal(host, oracle.whisper_prepare_audio(pcm))
    assert len(host) // 160 == 396


@pytest.mark.parametrize("n", [1, 255, 4096, 16000 * 61 + 7])
def test_normalize_int16_is_numpy_bit_for_bit(pipe, n):
    rng = np.random.default_rng(n)
    x = rng.integers(-20000, 32767, n).astype(np.int16)
    want = oracle.normalize_audio(x, max_val=1.0)
    got = pipe.normalize(torch.from_numpy(x).cuda()).cpu().numpy()
    assert got.dtype == np.float32 and np.array_equal(got, want)
    flat = np.full(n, 123, np.int16)                       # a flat clip only loses its mean
    assert np.array_equal(pipe.normalize(torch.from_numpy(flat).cuda()).cpu().numpy(), np.zeros(n, np.float32))


def test_normalize_float32(pipe):
    """float32 clips: numpy sums the mean pairwise in float32, which no parallel sum reproduces bit
    for bit; the results agree to a float32 rounding of the peak-normalised signal."""
    rng = np.random.default_rng(3)
    x = (rng.standard_normal(16000 * 7 + 3) * 0.2 + 0.01).astype(np.float32)
    want = oracle.normalize_audio(x, max_val=1.0)
    got = pipe.normalize(torch.from_numpy(x).cuda()).cpu().numpy()
    assert np.abs(got - want).max() < 2e-7 and abs(np.abs(got).max() - 1.0) < 1e-6


@pytest.mark.parametrize("mode", ["remove", "start", "end", "start_end"])
def test_trim_silence_equals_host(pipe, mode):
    from speech_to_text_amd.audio.audio_processing import trim_silence_rms
    rng = np.random.default_rng(11)
    for n in (16000 * 3 + 11, 16000 * 40, 700):
        x = (rng.standard_normal(n) * 0.3).astype(np.float32)
        for _ in range(4):
            a = int(rng.integers(0, n)); x[a: a + int(rng.integers(300, 12000))] *= 1e-4
        x[: 2500] = 0.0
        kw = dict(mode=mode, threshold=-25, min_silence=0.1, replace_by=0.4)
        want = trim_silence_rms(x, rate=16000, **kw)
        got = pipe.trim_silence(torch.from_numpy(x).cuda(), **kw).cpu().numpy()
        assert np.array_equal(got, want), (mode, n, len(got), len(want))


def test_reference_resample_fixture_on_the_device(golden_dir):
    """The reference's own `audio_resample.npy` (its `load_audio(audio_test.wav, rate = 22050)`,
    tests/test_utils_audio.py:62-64) through the device pipeline: decode, 16 k -> 22.05 k Fourier
    resampling (rocFFT, float64), float64 normalisation (stt_pcm_normalize, STT_PCM_F64), one cast."""
    from speech_to_text_amd.audio.device_pipeline import DeviceAudioPipeline
    want = np.load(os.path.join(golden_dir, "audio_resample.npy"))
    rate, got = DeviceAudioPipeline(rate=22050)(os.path.join(golden_dir, "audio_test.wav"), normalize=True)
    assert rate == 22050 and got.dtype == torch.float32 and got.shape[0] == want.shape[0]
    assert np.abs(got.cpu().numpy() - want).max() < 1e-6


def test_float64_normalisation_follows_numpy(pipe):
    from speech_to_text_amd.audio.audio_processing import normalize_audio
    x = np.random.default_rng(3).standard_normal(100003) * 1234.5 + 17.0
    got = pipe.normalize(torch.from_numpy(x).cuda()).cpu().numpy()
    want = normalize_audio(x, max_val=1.)
    assert got.dtype == want.dtype == np.float32
    assert np.abs(got - want).max() <= 6e-8           # one float32 ulp at 1: the mean's summation order
    flat = np.full(5000, 3.25)
    assert np.array_equal(pipe.normalize(torch.from_numpy(flat).cuda()).cpu().numpy(), np.zeros(5000, np.float32))


def test_resampled_clip_and_wrapper(pipe, tmp_path, golden_dir):
    """A 22.05 kHz file: Fourier resampling on the device (rocFFT through torch.fft, float64) equals
    scipy.signal.resample on the host; and Whisper.get_input(device_pipeline = True) feeds the same
    mel to the engine as the host pre-steps."""
    import scipy.signal
    rng = np.random.default_rng(2)
    n = 22050 * 3 + 123
    x = (rng.standard_normal(n) * 6000).astype(np.int16)
    path = str(tmp_path / "clip22k.wav")
    wavfile.write(path, 22050, x)
    rate, dev = pipe(path, normalize=True, trim_silence=False)
    want = oracle.normalize_audio(scipy.signal.resample(x, int(n / 22050 * 16000)), max_val=1.0)
    assert rate == 16000 and dev.shape[0] == len(want)
    assert np.abs(dev.cpu().numpy() - want).max() < 1e-6

    from speech_to_text_amd.stt import Whisper
    model = Whisper(pretrained="tiny", path="random:tiny:0")
    wav = os.path.join(golden_dir, "audio_test.wav")
    a = model.get_input(wav, pad_or_trim=False).cpu().numpy()
    b = model.get_input(wav, pad_or_trim=False, device_pipeline=True).cpu().numpy()
    assert a.shape == (396, 80) and np.array_equal(a, b)
