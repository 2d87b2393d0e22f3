"""Encoder-only workload for rocprofv3 --kernel-trace: large-v3, B windows, a few passes."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from speech_to_text_amd.engine import WhisperEngine, whisper_dims
from speech_to_text_amd.weights import random_whisper_weights
B = int(sys.argv[1]); name = sys.argv[2] if len(sys.argv) > 2 else "large-v3"
dims = whisper_dims(name)
eng = WhisperEngine(dims, random_whisper_weights(dims, 0), max_batch=B, max_beams=1, max_text_len=8, max_prompt_len=4)
mel = torch.rand(B, 3000, dims["n_mels"], device="cuda") * 2 - 1
for it in range(4):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    eng.encode(mel); torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("%s B=%d: encoder pass %.2f ms = %.3f ms per window" % (name, B, dt * 1e3, dt * 1e3 / B), flush=True)
