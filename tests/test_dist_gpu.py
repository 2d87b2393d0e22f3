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
