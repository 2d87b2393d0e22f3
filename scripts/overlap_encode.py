"""Do two encoder passes overlap?  Two engines of B windows each (large-v3 encoder, `layers` layers), their passes
issued (a) back to back on one stream, (b) on two streams at once.  If the memory-bound kernels of one pass
(LayerNorm, residual epilogues) hid under the other's matrix-core kernels, (b) would be shorter than (a).
python scripts/overlap_encode.py B [layers]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from speech_to_text_amd.engine import WhisperEngine, whisper_dims
from speech_to_text_amd.weights import random_whisper_weights
B = int(sys.argv[1])
layers = int(sys.argv[2]) if len(sys.argv) > 2 else 8
dims = dict(whisper_dims("large-v3"), n_audio_layer=layers, n_text_layer=2)
w = random_whisper_weights(dims, 0)
engs = [WhisperEngine(dims, w, max_batch=B, max_beams=1, max_text_len=16, max_prompt_len=4) for _ in range(2)]
one = WhisperEngine(dims, w, max_batch=2 * B, max_beams=1, max_text_len=16, max_prompt_len=4)
mel = torch.rand(2 * B, 3000, dims["n_mels"], device="cuda") * 2 - 1
streams = [torch.cuda.Stream(), torch.cuda.Stream()]


def timed(fn, n=3):
    best = 1e9
    for _ in range(n):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    return best * 1e3


def serial():
    for i, e in enumerate(engs):
        e.encode(mel[i * B:(i + 1) * B])


def concurrent():
    for i, e in enumerate(engs):
        with torch.cuda.stream(streams[i]):
            e.encode(mel[i * B:(i + 1) * B])


for f in (serial, concurrent):
    f()
one.encode(mel)
print("2 x %d windows, %d layers: one pass of %d %.1f ms, two passes back to back %.1f ms, on two streams %.1f ms"
      % (B, layers, 2 * B, timed(lambda: one.encode(mel)), timed(serial), timed(concurrent)), flush=True)
