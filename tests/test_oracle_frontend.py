"""The numpy oracle against the reference's own golden mel and known answers (CPU only)."""
import os

import numpy as np
from scipy.io import wavfile

from oracle import frontend as oracle


def _wav(golden_dir):
    rate, pcm = wavfile.read(os.path.join(golden_dir, "audio_test.wav"))
    assert rate == 16000 and pcm.dtype == np.int16 and pcm.shape == (64880,)
    return pcm


def test_oracle_reproduces_reference_golden_mel(golden_dir):
    # reference: tests/test_utils_audio.py:102-111, load_mel(audio_test.wav, WhisperSTFT()), tol 2e-3
    gold = np.load(os.path.join(golden_dir, "stft-WhisperSTFT.npy"))
    mel = oracle.whisper_log_mel(oracle.load_audio_like_reference(_wav(golden_dir)))
    assert mel.shape == gold.shape == (405, 80) and mel.dtype == np.float32
    assert np.abs(mel - gold).max() < 1e-4


def test_power_spectrum_would_not_match(golden_dir):
    # the reference uses the magnitude, not the power (stft.py:264-266, 346-353)
    gold = np.load(os.path.join(golden_dir, "stft-WhisperSTFT.npy"))
    audio = oracle.load_audio_like_reference(_wav(golden_dir))
    mag = oracle.stft_magnitude(audio)[:-1]
    mel = np.log10(np.maximum((mag ** 2) @ oracle.slaney_mel_basis(), 1e-10))
    mel = (np.maximum(mel, mel.max() - 8.0) + 4.0) / 4.0
    assert np.abs(mel - gold).max() > 0.1


def test_whisper_audio_preparation_gives_notebook_length(golden_dir):
    # notebook cell-13: audio_en.wav (same bytes as audio_test.wav) -> 396 frames, end = 3.96 s
    audio = oracle.whisper_prepare_audio(_wav(golden_dir))
    assert len(audio) == 63360
    assert oracle.whisper_log_mel(audio).shape == (396, 80)


def test_torch_formulation_equals_numpy(golden_dir):
    audio = oracle.load_audio_like_reference(_wav(golden_dir))
    a = oracle.whisper_log_mel(audio)
    b = oracle.whisper_log_mel_torch(audio)
    assert np.abs(a - b).max() < 5e-5     # two float32 summation orders, quiet bins
    c = oracle.whisper_log_mel_f64(audio)
    assert np.abs(a - c).max() < 5e-5


def test_short_and_edge_lengths():
    rng = np.random.default_rng(0)
    for n, frames in [(1, 2), (399, 2), (400, 2), (479, 2), (480, 3), (16000, 100)]:
        mel = oracle.whisper_log_mel(rng.uniform(-1, 1, n).astype(np.float32))
        assert mel.shape == (frames, 80)
    z = oracle.whisper_log_mel(np.zeros(1000, np.float32))
    assert np.allclose(z, -1.5)


def test_product_mel_basis_matches_oracle():
    from speech_to_text_amd.audio.mel_basis import hann_window, mel_filterbank
    for n_mels in (80, 128):
        a = mel_filterbank(16000, 400, n_mels, 0.0, 8000.0)
        b = oracle.slaney_mel_basis(n_mels=n_mels)
        assert a.shape == b.shape == (201, n_mels)
        assert np.abs(a - b).max() < 1e-9
        assert ((a != 0) == (b != 0)).all()
    assert np.abs(hann_window(400) - oracle.hann_periodic(400)).max() == 0


def test_pad_or_trim():
    mel = np.ones((10, 80), np.float32)
    out = oracle.pad_or_trim(mel, 16)
    assert out.shape == (16, 80) and out[10:].sum() == 0 and out[:10].sum() == 800
    assert oracle.pad_or_trim(np.ones((20, 80), np.float32), 16).shape == (16, 80)
