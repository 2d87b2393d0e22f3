"""BASELINE config 5 shape: whisper-large-v3, beam search (5 beams), timestamp filter on."""
import sys, time
import numpy as np, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from speech_to_text_amd.engine import WhisperEngine, whisper_dims, TimestampFilter
from speech_to_text_amd.weights import random_whisper_weights
name = sys.argv[1] if len(sys.argv) > 1 else 'large-v3'
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 128
dims = whisper_dims(name)
w = random_whisper_weights(dims, 0)
eng = WhisperEngine(dims, w, max_batch=B, max_beams=5, max_text_len=steps + 8, max_prompt_len=4)
del w
mel = torch.rand(B, 3000, dims['n_mels'], device='cuda') * 2 - 1
prompt = np.array([[50258, 50259, 50359]] * B, dtype=np.int32)
filt = TimestampFilter(np.arange(1, 90, 7), np.array([220, 50257]), 50364, 50257)
for it in range(2):
    torch.cuda.synchronize(); t0 = time.time()
    eng.encode(mel); torch.cuda.synchronize(); t1 = time.time()
    toks, lens, sc = eng.generate(prompt, method='beam', num_beams=5, num_sentences=1, max_new_tokens=steps,
                                  eos_token=50257, pad_token=50257, early_stopping=False, logits_filter=filt)
    t2 = time.time()
    print(f'{name} B={B} beam 5 + timestamp filter: encode {1e3*(t1-t0):.1f} ms, {steps} tokens {1e3*(t2-t1):.1f} ms '
          f'({1e3*(t2-t1)/steps:.2f} ms/token), audio-s/s {B*30/(t2-t0):.0f}', flush=True)
print(toks.shape, lens[:4].ravel())
# the cross-attention launch of a beam token step on its own (HIP events around single launches)
L = dims['n_text_layer']
evs = []
for it in range(3 * L):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); eng.probe_kernel(0x100 | 5, it % L); b.record()
    evs.append((a, b))
torch.cuda.synchronize()
us = 1e3 * float(np.mean([a.elapsed_time(b) for a, b in evs[L:]]))
gb = B * 2 * 1500 * dims['n_text_state'] * 2 / 1e9
print(f'cross-attention, 5 queries per window, {B} windows: {us:.1f} us per launch for {gb:.3f} GB = {gb / us * 1e3:.2f} TB/s '
      f'({"VALU kernel" if os.environ.get("STT_MI355X_CROSS_VALU") else "matrix-core kernel"})', flush=True)
