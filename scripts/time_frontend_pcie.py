"""Log-mel of a 1 h clip when the caller holds host buffers: pinned PCM -> device, kernel, mel -> pinned
host, all on one stream.  (The C ABI takes device pointers; this is what a host-side caller pays on top.)"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from speech_to_text_amd.audio.stft import WhisperSTFT
n = 57_600_000
fn = WhisperSTFT()
host = torch.from_numpy(np.random.default_rng(0).uniform(-1, 1, n).astype(np.float32)).pin_memory()
out_host = torch.empty((1, n // 160, 80), dtype=torch.float32).pin_memory()
dev = torch.empty(n, dtype=torch.float32, device="cuda")
for _ in range(2):
    dev.copy_(host, non_blocking=True); mel = fn(dev); out_host.copy_(mel, non_blocking=True)
torch.cuda.synchronize()
K = 5
for what in ("h2d + kernel + d2h", "h2d + kernel", "kernel"):
    t0 = time.perf_counter()
    for _ in range(K):
        if what != "kernel":
            dev.copy_(host, non_blocking=True)
        mel = fn(dev)
        if what == "h2d + kernel + d2h":
            out_host.copy_(mel, non_blocking=True)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / K
    print(f"{what:20s} {dt*1e3:8.3f} ms per hour of audio  {345.6e6/dt/1e9:8.1f} GB/s algorithmic", flush=True)
