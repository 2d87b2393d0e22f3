"""One process per GPU over torch.distributed (backend "nccl" = RCCL on ROCm, "gloo" on CPU).

Long-form audio is cut into 30 s windows; independent windows are dealt round-robin to the
ranks, every rank decodes its own with replicated weights, and the per-window token lists (a few
KB) are gathered on every rank.  No collective touches the encoder or decoder.  The only
reduction in the whole path is one float: the clip-wide maximum of the log-mel when a single
clip's frontend is itself split across ranks.

The reference has no multi-GPU path (everything is batch 1, sequential: models/stt/whisper.py:
228-249); this is the MI355X-side partitioning of its per-window loop.  Its default sequential
mode does not shard inside a file: there whole files are dealt to the ranks (`transcribe_files`).
"""
import numpy as np
import torch
import torch.distributed as dist


def world():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def window_shard(n_windows, rank, world_size):
    """Indices of the windows rank `rank` decodes (round robin keeps the ranks within one window
    of each other whatever the clip length)."""
    return [w for w in range(n_windows) if w % world_size == rank]


def frame_shard(n_frames, rank, world_size, multiple=118):
    """Contiguous [begin, end) frame range of rank `rank`, cut at multiples of `multiple`
    frames (the frontend kernel's tile)."""
    per = -(-n_frames // world_size)
    per = -(-per // multiple) * multiple
    begin = min(n_frames, rank * per)
    return begin, min(n_frames, begin + per)


def gather_window_results(local, group=None):
    """local: {window index: result}.  Returns the union over ranks, on every rank."""
    rank, ws = world()
    if ws == 1:
        return dict(local)
    parts = [None] * ws
    dist.all_gather_object(parts, local, group=group)
    merged = {}
    for p in parts:
        merged.update(p)
    return merged


def transcribe_sharded(model, mel, *, batch_size=32, group=None, **kwargs):
    """Every rank calls this with the same full-clip mel; returns the ordered segments of the
    whole clip on every rank."""
    rank, ws = world()
    local = model.infer_windows(mel, batch_size=batch_size, shard=(rank, ws), **kwargs)
    merged = gather_window_results(local, group=group)
    segments = [merged[w] for w in sorted(merged)]
    return {'text': ' '.join(s['text'] for s in segments), 'segments': segments}


def file_shard(durations, rank, world_size):
    """Indices of the files rank `rank` transcribes.  The sequential mode of the reference
    (previous-text prompt and timestamp seek, models/stt/whisper.py:238-243, 296-300) makes every
    window of a file depend on the one before, so a file is the unit that can move to another
    GPU.  Files are dealt longest first, each to the rank with the least audio so far (ties to
    the lowest rank): the same deal on every rank, no communication."""
    load = [0.0] * world_size
    mine = []
    order = sorted(range(len(durations)), key=lambda i: (-float(durations[i]), i))
    for i in order:
        r = min(range(world_size), key=lambda q: (load[q], q))
        load[r] += float(durations[i])
        if r == rank:
            mine.append(i)
    return sorted(mine)


def transcribe_files(model, files, *, durations=None, group=None, **kwargs):
    """Every rank calls this with the same list of files (paths, arrays or dicts `model.infer`
    accepts); each file is transcribed in the reference's sequential mode by ONE rank
    (`file_shard`), and the per-file results are gathered on every rank, in the order of `files`.
    `durations` (seconds, or any proxy such as file sizes) balances the deal; without it files are
    weighed by their sample count when they are arrays and equally otherwise."""
    rank, ws = world()
    if durations is None:
        durations = [len(f) if hasattr(f, '__len__') and not isinstance(f, (str, dict)) else 1.0 for f in files]
    if len(durations) != len(files):
        raise ValueError('durations: %d values for %d files' % (len(durations), len(files)))
    local = {i: model.infer(files[i], **kwargs) for i in file_shard(durations, rank, ws)}
    merged = gather_window_results(local, group=group)
    return [merged[i] for i in range(len(files))]


def sharded_log_mel(stft, pcm, *, group=None, gather=True):
    """Log-mel of ONE clip computed by all ranks: rank r takes a contiguous run of frames (with
    the 200-sample halo it needs), the ranks exchange one float (all-reduce max of the piece
    maxima, the `max - 8` clamp of utils/audio/stft.py:355-357 being clip-wide), and each clamps
    its own piece.  pcm: the whole clip on the host (numpy), identical on every rank.
    Returns the local piece [frames_local, n_mels], or with gather=True the full mel."""
    rank, ws = world()
    pcm = np.ascontiguousarray(np.asarray(pcm, dtype=np.float32).reshape(-1))
    if len(pcm) < 400:
        pcm = np.pad(pcm, (0, 400 - len(pcm)))
    n_total = len(pcm)
    n_frames = n_total // 160
    begin, end = frame_shard(n_frames, rank, ws)
    count = end - begin
    lo = max(0, begin * 160 - 200)
    lo -= lo % 4
    hi = min(n_total, (end - 1) * 160 + 200 + 4) if count > 0 else min(n_total, lo + 8)
    if begin == 0:
        hi = max(hi, min(n_total, 208))
    device = torch.device('cuda', torch.cuda.current_device())
    piece = torch.from_numpy(pcm[lo:hi]).to(device)
    mel, piece_max, state = stft.piece_spectrogram(piece, lo, n_total, begin, count)
    clip_max = piece_max.clone()
    if ws > 1:
        if dist.get_backend(group) == 'gloo':
            host = clip_max.cpu()
            dist.all_reduce(host, op=dist.ReduceOp.MAX, group=group)
            clip_max = host.to(device)
        else:
            dist.all_reduce(clip_max, op=dist.ReduceOp.MAX, group=group)
    mel = stft.finish_piece(mel, clip_max, state)
    if not gather or ws == 1:
        return mel
    parts = [None] * ws
    dist.all_gather_object(parts, (begin, mel.cpu().numpy()), group=group)
    parts.sort(key=lambda p: p[0])
    return torch.from_numpy(np.concatenate([p[1] for p in parts], axis=0)).to(device)
