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
@pytest.mark.parametrize("n", [1, 399, 400, 401, 559, 560, 39519, 39520, 39521,
                               39999, 40000, 40001, 40003, 40004, 40160,      # row-transfer threshold
                               79040 + 77, 480000])
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


def test_more_tiles_than_compute_units(stft_fns):
    """12.3 min of audio = 628 tiles of 118 frames for 512 workgroups: blocks of the persistent grid
    loop, the interior tiles take the row transfer and both end tiles the clamped one."""
    n = 247 * 160 * 300 + 1234
    rng = np.random.default_rng(7)
    audio = rng.uniform(-1, 1, n).astype(np.float32)
    audio[5_000_000:5_200_000] = 0.0          # a silent stretch, so the clamp is active
    got = _gpu_mel(stft_fns[80], audio)
    ref = oracle.whisper_log_mel_torch(audio)
    assert got.shape == ref.shape
    assert np.abs(got - ref).max() < TOL


def test_one_hour_properties(stft_fns):
    """BASELINE size (1 h, 360 000 frames).  The oracle is too slow to run whole here, so:
    (1) away from the clip ends a frame depends only on its own 400 samples, hence a slice of the
    clip, transformed alone with the whole clip's maximum, reproduces the same rows bit for bit;
    (2) the result is deterministic across calls; (3) random rows agree with the oracle."""
    import torch
    n = 57_600_000
    g = torch.Generator(device="cuda"); g.manual_seed(0)
    pcm = torch.rand(n, device="cuda", generator=g) * 2 - 1
    fn = stft_fns[80]
    a = fn(pcm)[0]
    b = fn(pcm)[0]
    torch.cuda.synchronize()
    assert a.shape == (360000, 80)
    assert torch.equal(a, b)
    # rows [f0, f0 + 3000) from a slice that starts 20 frames earlier and ends 20 frames later
    host = pcm.cpu().numpy()
    rng = np.random.default_rng(3)
    for f0 in (1000, 123_456, 356_000):
        lo, hi = (f0 - 20) * 160, (f0 + 3020) * 160
        piece = oracle.whisper_log_mel(host[lo:hi])
        # the oracle clamps with the slice's own maximum; rows above both clamps must agree
        got = a[f0:f0 + 3000].cpu().numpy()
        ref = piece[20:3020]
        floor = max(got.min(), ref.min())
        mask = (got > floor + 1e-6) & (ref > floor + 1e-6)
        assert mask.mean() > 0.99
        assert np.abs(got - ref)[mask].max() < TOL


def test_two_threads_two_streams(stft_fns):
    """Two host threads, each on its own stream, calling one WhisperSTFT at the same time: plans
    (ticket counter, per-block maxima) are per stream, so each gets the single-threaded result."""
    import threading
    import torch
    fn = stft_fns[80]
    rng = np.random.default_rng(8)
    clips = [rng.uniform(-1, 1, 16000 * 50 + 17 * i).astype(np.float32) for i in range(2)]
    for c in clips:
        c[3000:40000] = 0.0                       # the clamp matters: the clip maximum must be this clip's
    clips[1] *= 1e-3
    want = [fn(c)[0].cpu().numpy() for c in clips]
    got, errors = [None, None], []

    def work(i):
        try:
            with torch.cuda.stream(torch.cuda.Stream()):
                for _ in range(20):
                    out = fn(clips[i])[0]
                got[i] = out.cpu().numpy()
        except BaseException as e:
            errors.append(e)

    threads = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors
    for i in range(2):
        assert np.array_equal(got[i], want[i])
