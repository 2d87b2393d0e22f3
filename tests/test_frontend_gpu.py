"""HIP log-mel frontend against the numpy oracle and the reference's golden mel."""
import os

import numpy as np
import pytest
from scipy.io import wavfile

from oracle import frontend as oracle

pytestmark = pytest.mark.gpu

TOL = 1e-4      # abs, float32 (north-star tolerance for the frontend)


@pytest.fixture(scope="module")
def stft_fns():
    import torch
    from speech_to_text_amd.audio.stft import WhisperSTFT
    assert torch.cuda.is_available()
    return {80: WhisperSTFT(n_mel_channels=80), 128: WhisperSTFT(n_mel_channels=128)}


def _gpu_mel(fn, audio):
    import torch
    out = fn(audio)
    torch.cuda.synchronize()
    return out[0].cpu().numpy()


def test_golden_fixture(stft_fns, golden_dir):
    rate, pcm = wavfile.read(os.path.join(golden_dir, "audio_test.wav"))
    audio = oracle.load_audio_like_reference(pcm)
    got = _gpu_mel(stft_fns[80], audio)
    gold = np.load(os.path.join(golden_dir, "stft-WhisperSTFT.npy"))
    assert got.shape == gold.shape == (405, 80)
    # the reference's own tolerance for this file is 2e-3 (tests/test_utils_audio.py:109-111)
    assert np.abs(got - gold).max() < 1e-4
    assert np.abs(got - oracle.whisper_log_mel(audio)).max() < TOL


@pytest.mark.parametrize("n_mels", [80, 128])
@pytest.mark.parametrize("n", [1, 399, 400, 401, 559, 560, 39519, 39520, 39521, 79040 + 77, 480000])
def test_lengths_vs_oracle(stft_fns, n_mels, n):
    rng = np.random.default_rng(n)
    audio = rng.uniform(-1, 1, n).astype(np.float32)
    got = _gpu_mel(stft_fns[n_mels], audio)
    ref = oracle.whisper_log_mel(audio, n_mels=n_mels)
    assert got.shape == ref.shape
    assert np.abs(got - ref).max() < TOL


def test_silence_and_clamp(stft_fns):
    """Digital silence hits the 1e-10 floor, so the max-8 clamp becomes active."""
    rng = np.random.default_rng(5)
    audio = np.zeros(160 * 700, dtype=np.float32)
    audio[160 * 100:160 * 300] = rng.uniform(-1, 1, 160 * 200).astype(np.float32)
    audio[160 * 500:160 * 520] = 1e-6 * rng.uniform(-1, 1, 160 * 20).astype(np.float32)
    got = _gpu_mel(stft_fns[80], audio)
    ref = oracle.whisper_log_mel(audio)
    assert np.abs(got - ref).max() < TOL
    assert (got == got.min()).sum() > 1000        # clamped plateau exists
    # all-zero clip: everything sits on the floor
    z = _gpu_mel(stft_fns[80], np.zeros(4000, dtype=np.float32))
    assert np.allclose(z, (-10.0 + 4.0) / 4.0)
