"""Host-side audio pre-steps of the Whisper wrapper, as the reference applies them before the
mel (utils/audio/audio_io.py:249-264): ``normalize_audio`` then RMS silence removal.  They are
O(samples) numpy passes over the whole clip and stay on the host, before any sharding.

Follows utils/audio/audio_processing.py:48-60 (normalize_audio) and :98-198 (trim_silence_rms).
"""
import numpy as np


def normalize_audio(audio, max_val=32767, dtype=np.int16):
    """Zero mean, peak scaled to `max_val`; a float result when max_val <= 1 (what Whisper asks)."""
    out_type = np.float32 if max_val <= 1. else dtype
    centred = np.asarray(audio) - np.mean(audio)
    peak = float(np.abs(centred).max()) if centred.size else 0.
    gain = max_val / peak if peak > 1e-9 else 1.
    return (centred * gain).astype(out_type) if gain != 1. else centred.astype(out_type)


def block_peaks(audio, block_size):
    """Peak of x^2 per block of `block_size` samples (the last block zero-padded)."""
    x = np.asarray(audio)
    if not np.issubdtype(x.dtype, np.floating):
        x = x / np.iinfo(x.dtype).max
    n_blocks = -(-len(x) // block_size)
    padded = np.zeros(n_blocks * block_size, dtype=x.dtype)
    padded[: len(x)] = x
    return (padded.reshape(n_blocks, block_size) ** 2).max(axis=1)


def _silence_intervals(peak_sq, n, rate, threshold, min_silence, block_size, min_voice_time):
    """Silent stretches of a clip of `n` samples as two float64 arrays (start, end) in seconds,
    from its per-block peaks.

    A block of `block_size` samples is silent when its peak magnitude is below the threshold
    (dB re full scale); runs of silent blocks shorter than `min_silence` do not count, and two
    stretches separated by less than `min_voice_time` of sound are joined.  Times are kept as
    block index x block duration in float64 and only the caller converts them to samples, so
    that the truncations land where the reference's do."""
    block_time = block_size / rate
    n_blocks = len(peak_sq)
    silent = np.sqrt(peak_sq) < 10 ** (threshold / 20.0)

    # run lengths: a run starts wherever the flag differs from the block before
    first = np.flatnonzero(np.r_[True, silent[1:] != silent[:-1]])
    length = np.diff(np.r_[first, n_blocks])
    keep = silent[first] & (length * block_time >= min_silence)
    start = first[keep] * block_time
    end = np.minimum(n / rate, (first[keep] + length[keep]) * block_time)
    if min_voice_time and len(start) > 1:
        # a stretch opens a new group unless less than min_voice_time of sound precedes it
        opens = np.r_[True, (start[1:] - end[:-1]) >= min_voice_time]
        closes = np.r_[opens[1:], True]
        start, end = start[opens], end[closes]
    return start, end


def kept_stretches(peak_sq, n, *, rate, mode='start_end', threshold=-25, min_silence=0.1,
                   block_size=0.01, replace_by=0.5, min_voice_time=0.2):
    """What the silence removal keeps of a clip of `n` samples, as sorted disjoint sample ranges
    [(begin, end)], decided from the per-block peaks alone (so the clip itself may live on the GPU).
    `mode='remove'`: every silent stretch is cut down to `replace_by` samples (all of it at the head
    and tail of the clip, half on either side inside); 'start' / 'end' / 'start_end' only trim the
    ends.  Same decisions as utils/audio/audio_processing.py:98-198 of the reference."""
    if isinstance(replace_by, float):
        replace_by = int(replace_by * rate)
    if isinstance(block_size, float):
        block_size = int(block_size * rate)
    start_t, end_t = _silence_intervals(peak_sq, n, rate, threshold, min_silence, block_size, min_voice_time)
    lo = (start_t * rate).astype(np.int64)          # truncation, like int()
    hi = (end_t * rate).astype(np.int64)

    if mode != 'remove':
        begin, stop = 0, n
        if 'end' in mode and len(lo) and abs(end_t[-1] * rate - n) <= 1:
            stop = min(n, int(lo[-1]) + replace_by)
        if 'start' in mode and len(lo) and start_t[0] == 0:
            begin = max(0, int(hi[0]) - replace_by)
        return [(begin, stop)] if begin < stop else []

    at_head = lo == 0
    at_tail = (np.abs(hi - n) <= 1) & ~at_head
    half = replace_by // 2
    cut_lo = np.where(at_head, 0, np.where(at_tail, lo + replace_by, lo + half))
    cut_hi = np.where(at_head, hi - replace_by, np.where(at_tail, n, hi - half))
    cut_lo, cut_hi = np.clip(cut_lo, 0, n), np.clip(cut_hi, 0, n)
    ok = cut_lo < cut_hi
    # +1 / -1 at the borders of every cut, summed up: > 0 inside any cut; the kept stretches are
    # the runs of zero
    delta = np.zeros(n + 1, dtype=np.int32)
    np.add.at(delta, cut_lo[ok], 1)
    np.add.at(delta, cut_hi[ok], -1)
    kept, pos, depth = [], 0, 0
    for e in np.flatnonzero(delta).tolist():
        if depth == 0 and e > pos:
            kept.append((pos, e))
        depth += int(delta[e])
        if depth == 0:
            pos = e
    if depth == 0 and pos < n:
        kept.append((pos, n))
    return kept


def trim_silence_rms(audio, *, rate, mode='start_end', threshold=-25, min_silence=0.1,
                     block_size=0.01, replace_by=0.5, min_voice_time=0.2, **kwargs):
    """RMS silence removal of utils/audio/audio_processing.py:98-198 (see kept_stretches)."""
    if isinstance(block_size, float):
        block_size = int(block_size * rate)
    keep = kept_stretches(block_peaks(audio, block_size), len(audio), rate=rate, mode=mode,
                          threshold=threshold, min_silence=min_silence, block_size=block_size,
                          replace_by=replace_by, min_voice_time=min_voice_time)
    if len(keep) == 1:
        return audio[keep[0][0]: keep[0][1]]
    if not keep:
        return audio[:0]
    return np.concatenate([audio[a:b] for a, b in keep])


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
