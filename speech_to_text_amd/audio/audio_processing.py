"""Host-side audio pre-steps of the Whisper wrapper, as the reference applies them before the
mel (utils/audio/audio_io.py:249-264): ``normalize_audio`` then RMS silence removal.  They are
O(samples) numpy passes over the whole clip and stay on the host, before any sharding.

Follows utils/audio/audio_processing.py:48-60 (normalize_audio) and :98-198 (trim_silence_rms).
"""
import itertools

import numpy as np


def normalize_audio(audio, max_val=32767, dtype=np.int16):
    if max_val <= 1.:
        dtype = np.float32
    audio = audio - np.mean(audio)
    peak = np.max(np.abs(audio))
    if peak <= 1e-9:
        return audio.astype(dtype)
    return (audio * (max_val / peak)).astype(dtype)


def trim_silence_rms(audio, *, rate, mode='start_end', threshold=-25, min_silence=0.1,
                     block_size=0.01, replace_by=0.5, min_voice_time=0.2, **kwargs):
    if isinstance(replace_by, float): replace_by = int(replace_by * rate)
    if isinstance(block_size, float): block_size = int(block_size * rate)
    block_time = block_size / rate
    amplitude_threshold = 10 ** (threshold / 20.0)

    blocks = audio
    if not np.issubdtype(blocks.dtype, np.floating):
        blocks = blocks / np.iinfo(blocks.dtype).max
    pad = len(blocks) % block_size
    if pad != 0:
        blocks = np.pad(blocks, [(0, block_size - pad)])
    blocks = blocks.reshape(-1, block_size)
    silent = np.sqrt(np.max(blocks ** 2, axis=1)) < amplitude_threshold

    # run-length groups of silent blocks that last at least `min_silence`
    starts, ends, idx = [], [], 0
    for is_silent, group in itertools.groupby(silent):
        n = sum(1 for _ in group)
        if is_silent and n * block_time >= min_silence:
            starts.append(idx * block_time)
            ends.append(min(len(audio) / rate, (idx + n) * block_time))
        idx += n
    if min_voice_time:
        for i in reversed(range(len(starts) - 1)):
            if starts[i + 1] - ends[i] < min_voice_time:
                starts.pop(i + 1)
                ends.pop(i)

    if mode == 'remove':
        half = replace_by // 2
        drop = np.zeros((len(audio),), dtype=bool)
        for start, end in zip(starts, ends):
            start, end = int(start * rate), int(end * rate)
            if start == 0:
                drop[: max(0, end - replace_by)] = True
            elif abs(end - len(audio)) <= 1:
                drop[start + replace_by:] = True
            else:
                start, end = start + half, end - half
                if start < end:
                    drop[start: end] = True
        return audio[~drop]
    if 'end' in mode and starts and abs(ends[-1] * rate - len(audio)) <= 1:
        audio = audio[: int(starts[-1] * rate) + replace_by]
    if 'start' in mode and starts and starts[0] == 0:
        audio = audio[max(0, int(ends[0] * rate) - replace_by):]
    return audio


# what Whisper ends up with in `trim_kwargs` (models/interfaces/base_audio_model.py:20-28, 121-132)
WHISPER_TRIM_KWARGS = dict(mode='remove', threshold=-25, min_silence=0.1, replace_by=0.4)


def read_audio(data, *, rate=16000, normalize=True, trim_silence=False, **kwargs):
    """`.wav` path or raw samples -> (rate, float32 audio); ffmpeg-decoded formats are not
    available here (no ffmpeg binary), resample beforehand."""
    if isinstance(data, str):
        from scipy.io import wavfile
        file_rate, audio = wavfile.read(data)
        if audio.ndim > 1:
            audio = audio[:, 0]
        if file_rate != rate:
            import scipy.signal
            audio = scipy.signal.resample(audio, int(len(audio) / file_rate * rate))
    else:
        audio = np.asarray(data)
    if normalize is True:
        audio = normalize_audio(audio, max_val=1.)
    elif normalize and normalize > 1 and np.issubdtype(audio.dtype, np.integer):
        audio = (audio / normalize).astype(np.float32)
    if trim_silence:
        audio = trim_silence_rms(audio, rate=rate, **kwargs)
    return rate, audio
