"""Two engine replicas on two host threads and streams, phase-shifted: one encodes a batch while the
other decodes the previous one (encoder = matrix-core bound, decode = HBM bound)."""
import sys, time, threading
import numpy as np, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from speech_to_text_amd.engine import WhisperEngine, whisper_dims
from speech_to_text_amd.weights import random_whisper_weights
name = sys.argv[1] if len(sys.argv) > 1 else 'large-v3'
B = int(sys.argv[2]) if len(sys.argv) > 2 else 128
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 128
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 4
dims = whisper_dims(name)
w = random_whisper_weights(dims, 0)
engs = [WhisperEngine(dims, w, max_batch=B, max_beams=1, max_text_len=steps + 8, max_prompt_len=4) for _ in range(2)]
del w
mel = torch.rand(B, 3000, dims['n_mels'], device='cuda') * 2 - 1
prompt = np.array([[50258, 50259, 50359]] * B, dtype=np.int32)
streams = [torch.cuda.Stream() for _ in range(2)]
encoded = [threading.Event() for _ in range(2)]
def work(r, n):
    with torch.cuda.stream(streams[r]):
        if r == 1:
            encoded[0].wait()          # start one phase late
        for i in range(n):
            engs[r].encode(mel)
            streams[r].synchronize()
            encoded[r].set()
            engs[r].generate(prompt, max_new_tokens=steps, eos_token=50257, pad_token=50257, early_stopping=False)
for r in range(2):                      # warm-up, sequential
    work_r = threading.Thread(target=lambda r=r: (engs[r].encode(mel), engs[r].generate(prompt, max_new_tokens=8, eos_token=50257, pad_token=50257, early_stopping=False)))
    work_r.start(); work_r.join()
torch.cuda.synchronize(); t0 = time.time()
th = [threading.Thread(target=work, args=(r, iters)) for r in range(2)]
[t.start() for t in th]; [t.join() for t in th]
torch.cuda.synchronize(); t1 = time.time()
print(f'{name} 2 x B={B}, {iters} steps each, phase-shifted: {1e3 * (t1 - t0):.0f} ms, audio-s/s {2 * iters * B * 30 / (t1 - t0):.0f}', flush=True)
