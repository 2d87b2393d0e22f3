"""Two engine replicas, each decoding half of the windows on its own stream, driven by two host threads."""
import sys, time, threading
import numpy as np, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from speech_to_text_amd.engine import WhisperEngine, whisper_dims
from speech_to_text_amd.weights import random_whisper_weights
name = sys.argv[1] if len(sys.argv) > 1 else 'large-v3'
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 128
R = int(sys.argv[4]) if len(sys.argv) > 4 else 2
dims = whisper_dims(name)
w = random_whisper_weights(dims, 0)
engs = [WhisperEngine(dims, w, max_batch=B // R, max_beams=1, max_text_len=steps + 8, max_prompt_len=4) for _ in range(R)]
del w
mel = torch.rand(B, 3000, dims['n_mels'], device='cuda') * 2 - 1
prompt = np.array([[50258, 50259, 50359]] * (B // R), dtype=np.int32)
streams = [torch.cuda.Stream() for _ in range(R)]
def work(r, out):
    with torch.cuda.stream(streams[r]):
        engs[r].encode(mel[r * (B // R):(r + 1) * (B // R)])
        out[r] = engs[r].generate(prompt, max_new_tokens=steps, eos_token=50257, pad_token=50257, early_stopping=False)
for it in range(3):
    torch.cuda.synchronize(); t0 = time.time()
    out = [None] * R
    th = [threading.Thread(target=work, args=(r, out)) for r in range(R)]
    [t.start() for t in th]; [t.join() for t in th]
    torch.cuda.synchronize(); t1 = time.time()
    print(f'{name} B={B} as {R} x {B // R}: {1e3 * (t1 - t0):.1f} ms per step, audio-s/s {B * 30 / (t1 - t0):.0f}', flush=True)
