"""The allocation-free log-mel entry point (stt_logmel_run_static) against stt_logmel_run, both
through the C ABI: bit-identical output on two calls in a row, the second on a smaller clip
(the static workspace keeps the first call's contents), a short clip (the zero-padded copy lives
in the workspace), and a clear error beyond the stated maximum."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _both(lib, _lib, plan, pcm, n_mels):
    n = pcm.numel()
    frames = lib.stt_logmel_num_frames(n)
    ws_bytes = lib.stt_logmel_workspace_bytes(n)
    stream = torch.cuda.current_stream().cuda_stream
    want = torch.full((frames, n_mels), float("nan"), dtype=torch.float32, device=pcm.device)
    got = torch.full_like(want, float("nan"))
    ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=pcm.device)
    _lib.check(lib.stt_logmel_run(plan, pcm.data_ptr(), n, want.data_ptr(), ws.data_ptr(), ws_bytes, stream))
    _lib.check(lib.stt_logmel_run_static(plan, pcm.data_ptr(), n, got.data_ptr(), stream))
    torch.cuda.synchronize()
    return got.cpu().numpy(), want.cpu().numpy()


@pytest.mark.parametrize("n_mels", [80, 128])
def test_static_workspace_entry_point_matches_run(n_mels):
    from speech_to_text_amd import _lib
    from speech_to_text_amd.audio.stft import WhisperSTFT
    lib = _lib.load()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    fn = WhisperSTFT(n_mel_channels=n_mels)        # owns the native plan: keep it alive
    plan = fn._plan(dev)
    rng = np.random.default_rng(5)
    big = rng.uniform(-1, 1, 16000 * 600 + 77).astype(np.float32)
    big[200000:300000] = 0.0                       # a silent stretch: the clamp pass has work to do
    small = rng.uniform(-1, 1, 16000 * 7 + 123).astype(np.float32) * 0.01
    for clip in (big, small, small[:250], big[: 16000 * 31]):
        pcm = torch.from_numpy(clip).to(dev)
        got, want = _both(lib, _lib, plan, pcm, n_mels)
        assert not np.isnan(want).any()
        assert np.array_equal(got, want)           # same kernels, same workspace layout: same bits


def test_static_workspace_rejects_long_clips():
    from speech_to_text_amd import _lib
    from speech_to_text_amd.audio.stft import WhisperSTFT
    lib = _lib.load()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    fn = WhisperSTFT(n_mel_channels=80)            # owns the native plan: keep it alive
    plan = fn._plan(dev)
    limit = lib.stt_logmel_static_max_samples()
    assert limit >= 16000 * 3600                   # an hour of audio fits
    assert lib.stt_logmel_workspace_bytes(limit) <= 131072 < lib.stt_logmel_workspace_bytes(limit + 118 * 160 * 64)
    pcm = torch.zeros(1024, dtype=torch.float32, device=dev)
    mel = torch.zeros(1024, dtype=torch.float32, device=dev)
    rc = lib.stt_logmel_run_static(plan, pcm.data_ptr(), limit + 1, mel.data_ptr(), None)
    assert rc == -4                                # STT_ERR_WORKSPACE, nothing launched
    assert b"static workspace" in lib.stt_last_error()
