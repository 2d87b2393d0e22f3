"""Log-mel time per call as a function of how long the GPU has been kept busy with it (clock ramp)."""
import numpy as np, torch, sys, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from speech_to_text_amd.audio.stft import WhisperSTFT
hours = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
n = int(57_600_000 * hours)
fn = WhisperSTFT()
g = torch.Generator(device='cuda'); g.manual_seed(0)
pcm = (torch.rand(n, device='cuda', generator=g) * 2 - 1).float()
for _ in range(3): out = fn(pcm)
torch.cuda.synchronize()
for K in (20, 100, 400, 1600, 400, 20):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(K): out = fn(pcm)
    ev[1].record(); torch.cuda.synchronize()
    print("K=%d: %.1f us/call" % (K, ev[0].elapsed_time(ev[1]) / K * 1e3), flush=True)
