"""world_size-2 gloo runs of the multi-GPU orchestration (window sharding + result gather, and
the frame sharding arithmetic of the split frontend).  The numeric work is replaced by a stub
here: these tests cover who-gets-what and the exchange, not the kernels."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class _StubModel(object):
    """Looks like Whisper.infer_windows: tokens of window w are [w, w + 1]."""
    max_input_length = 3000

    def infer_windows(self, mel, *, batch_size=32, shard=None, **kw):
        n = -(-len(mel) // 3000)
        out = {}
        for w in range(n):
            if shard is None or w % shard[1] == shard[0]:
                out[w] = {'start': 30.0 * w, 'end': 30.0 * (w + 1), 'tokens': np.array([w, w + 1]),
                          'text': 'w%d' % w, 'lang': 'en'}
        return out

    def infer(self, audio, **kw):
        """Looks like Whisper.infer in sequential mode: one result per file, naming who ran it."""
        rank = dist.get_rank() if dist.is_initialized() else 0
        return {'text': 'file of %d samples' % len(audio), 'rank': rank,
                'segments': [{'start': 0.0, 'end': len(audio) / 16000.0}]}


def _worker(rank, world_size, port, tmp):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world_size)
    from speech_to_text_amd import dist as sd
    assert sd.world() == (rank, world_size)
    mel = np.zeros((3000 * 7 + 100, 80), np.float32)        # 8 windows
    res = sd.transcribe_sharded(_StubModel(), mel)
    assert [s['text'] for s in res['segments']] == ['w%d' % w for w in range(8)]
    assert res['text'] == ' '.join('w%d' % w for w in range(8))
    mine = sd.window_shard(8, rank, world_size)
    assert mine == list(range(rank, 8, world_size))
    # sequential mode: whole files are dealt to the ranks, results come back in file order
    files = [np.zeros(16000 * s, np.float32) for s in (5, 40, 7, 33, 1)]
    out = sd.transcribe_files(_StubModel(), files)
    assert [o['text'] for o in out] == ['file of %d samples' % len(f) for f in files]
    by_rank = [sum(len(f) for f, o in zip(files, out) if o['rank'] == r) for r in range(world_size)]
    assert sorted(by_rank) == sorted([16000 * 45, 16000 * 41])          # 40 + 5 | 33 + 7 + 1
    # scalar exchange used by the split frontend
    t = torch.tensor([float(rank) - 3.0])
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    assert float(t) == world_size - 1 - 3.0
    with open(os.path.join(tmp, "ok%d" % rank), "w") as f:
        f.write("ok")
    dist.destroy_process_group()


def test_two_rank_window_sharding(tmp_path):
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert os.path.exists(tmp_path / "ok0") and os.path.exists(tmp_path / "ok1")


def test_frame_shard_covers_every_frame_once():
    from speech_to_text_amd.dist import frame_shard, window_shard
    for n in (1, 239, 240, 241, 3000, 360000, 360001):
        for ws in (1, 2, 3, 8):
            seen = []
            for r in range(ws):
                b, e = frame_shard(n, r, ws)
                assert 0 <= b <= e <= n
                assert b % 118 == 0 or b == n
                seen += list(range(b, e)) if n < 5000 else [(b, e)]
            if n < 5000:
                assert seen == list(range(n))
            else:
                assert seen[0][0] == 0 and seen[-1][1] == n
                assert all(seen[i][1] == seen[i + 1][0] for i in range(ws - 1))
    assert sum(len(window_shard(1200, r, 8)) for r in range(8)) == 1200


def test_file_shard_is_a_balanced_partition():
    from speech_to_text_amd.dist import file_shard
    rng = np.random.default_rng(0)
    for n, ws in ((0, 2), (1, 4), (7, 3), (40, 8)):
        dur = rng.uniform(1, 600, n)
        parts = [file_shard(dur, r, ws) for r in range(ws)]
        assert sorted(i for p in parts for i in p) == list(range(n))
        if n >= ws:
            loads = [sum(dur[i] for i in p) for p in parts]
            assert max(loads) - min(loads) <= dur.max()            # longest-first greedy bound
    assert file_shard([3, 3, 3], 0, 2) == [0, 2] and file_shard([3, 3, 3], 1, 2) == [1]
