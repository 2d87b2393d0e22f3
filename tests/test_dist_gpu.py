"""One clip's log-mel computed by several ranks (gloo ranks sharing the box's GPU), including
ranks that own no frame at all: equal to the single-process result bit for bit."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r"""
import os, sys
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, os.environ["STT_ROOT"])
from speech_to_text_amd.audio.stft import WhisperSTFT
from speech_to_text_amd.dist import sharded_log_mel, frame_shard
torch.cuda.set_device(0)
dist.init_process_group("gloo")
rank, ws = dist.get_rank(), dist.get_world_size()
rng = np.random.default_rng(17)
pcm = rng.uniform(-1, 1, int(os.environ["STT_SAMPLES"])).astype(np.float32)
pcm[1000:9000] = 0.0
fn = WhisperSTFT()
b, e = frame_shard(len(pcm) // 160, rank, ws)
mel = sharded_log_mel(fn, pcm, gather=True).cpu().numpy()
if rank == 0:
    whole = fn(pcm)[0].cpu().numpy()
    assert mel.shape == whole.shape, (mel.shape, whole.shape)
    assert np.array_equal(mel, whole)
    print("OK frames", mel.shape[0])
owned = torch.tensor([e - b])
gathered = [torch.zeros(1, dtype=torch.long) for _ in range(ws)]
dist.all_gather(gathered, owned)
if rank == 0:
    print("OWNED", [int(g) for g in gathered])
dist.destroy_process_group()
"""


@pytest.mark.parametrize("ranks,samples", [(3, 16000 + 4011), (4, 16000 * 2), (2, 16000 * 30)])
def test_sharded_log_mel_with_idle_ranks(tmp_path, ranks, samples):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, STT_ROOT=ROOT, STT_SAMPLES=str(samples), MASTER_ADDR="127.0.0.1",
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
                          "--nproc-per-node", str(ranks), "--master-addr", "127.0.0.1",
                          "--master-port", str(29600 + ranks), str(script)],
                         env=env, capture_output=True, text=True, timeout=240)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    assert "OK frames %d" % (samples // 160) in out.stdout
    owned = eval(out.stdout.split("OWNED", 1)[1].splitlines()[0])
    assert sum(owned) == samples // 160
    if ranks > 2:
        assert 0 in owned            # at least one rank had nothing to compute and did not hang the others


FILES_WORKER = r"""
import os, sys, warnings
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, os.environ["STT_ROOT"])
from speech_to_text_amd.dist import transcribe_files, transcribe_sharded, file_shard
from speech_to_text_amd.stt import Whisper
torch.cuda.set_device(0)
dist.init_process_group("gloo")
rank, ws = dist.get_rank(), dist.get_world_size()
model = Whisper(pretrained="tiny", path="random:tiny:0", max_batch=2)
rng = np.random.default_rng(23)
files = [rng.uniform(-1, 1, 16000 * s).astype(np.float32) for s in (41, 9, 33)]
kw = dict(lang="en", max_length=24, normalize=False, trim_silence=False, logits_filter=list(range(50364, 51865)))
# sequential mode, one file per rank
got = transcribe_files(model, files, **kw)
# independent windows of one clip dealt to the ranks (real engines on every rank)
mel = model.get_input(files[0], pad_or_trim=False, normalize=False, trim_silence=False)
sharded = transcribe_sharded(model, mel, batch_size=1, lang="en", max_length=24, early_stopping=False,
                             logits_filter=kw["logits_filter"])
if rank == 0:
    assert file_shard([len(f) for f in files], 0, ws) != [0, 1, 2]
    for f, g in zip(files, got):
        want = model.infer(f, **kw)
        assert [s["start"] for s in want["segments"]] == [s["start"] for s in g["segments"]]
        for a, b in zip(want["segments"], g["segments"]):
            assert np.array_equal(a["tokens"], b["tokens"]) and a["end"] == b["end"]
    one = model.infer_windows(mel, batch_size=1, lang="en", max_length=24, early_stopping=False,
                              logits_filter=kw["logits_filter"])
    assert len(sharded["segments"]) == len(one) == 2
    for w, seg in enumerate(sharded["segments"]):
        assert np.array_equal(seg["tokens"], one[w]["tokens"])
    print("FILES OK", [len(g["segments"]) for g in got])
dist.destroy_process_group()
"""


def test_files_and_windows_dealt_to_ranks_with_real_engines(tmp_path):
    """Two gloo ranks sharing GPU 0, a real engine on each: the sequential mode with one file per
    rank (`transcribe_files`) and the independent windows of one clip dealt round-robin
    (`transcribe_sharded`) return what a single process returns."""
    script = tmp_path / "files_worker.py"
    script.write_text(FILES_WORKER)
    env = dict(os.environ, STT_ROOT=ROOT, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", "29655", str(script)],
                         env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    assert "FILES OK [2, 1, 2]" in out.stdout
