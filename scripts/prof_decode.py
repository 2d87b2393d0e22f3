"""Decode-only workload for rocprofv3 --kernel-trace: large-v3, B windows, T greedy tokens
(one encode, then generation).  python scripts/prof_decode.py B T [model] [layers]
`layers` cuts the model to that many encoder and decoder layers: the per-layer kernels keep their
shapes and the engine's buffers stay small enough for the profiler at 512 windows."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from speech_to_text_amd.engine import WhisperEngine, whisper_dims
from speech_to_text_amd.weights import random_whisper_weights
B, T = int(sys.argv[1]), int(sys.argv[2])
name = sys.argv[3] if len(sys.argv) > 3 else "large-v3"
dims = whisper_dims(name)
if len(sys.argv) > 4:
    dims = dict(dims, n_audio_layer=int(sys.argv[4]), n_text_layer=int(sys.argv[4]))
eng = WhisperEngine(dims, random_whisper_weights(dims, 0), max_batch=B, max_beams=1, max_text_len=T + 8, max_prompt_len=4)
mel = torch.rand(B, 3000, dims["n_mels"], device="cuda") * 2 - 1
prompt = np.array([[50258, 50259, 50359]] * B, dtype=np.int32)
eng.encode(mel)
for it in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    eng.generate(prompt, max_new_tokens=T, eos_token=50257, pad_token=50257, early_stopping=False)
    dt = time.perf_counter() - t0
    print("%s B=%d: %d tokens %.2f ms = %.3f ms per token" % (name, B, T, dt * 1e3, dt * 1e3 / T), flush=True)
