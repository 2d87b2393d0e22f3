"""Single-clip latency at the shapes of the reference's notebook timings (speech_to_text.ipynb
cell-13: whisper-base, one 3.96 s clip, mel of 63 360 samples, encoder + 13 greedy tokens;
cell-15: large-v3, 14 tokens)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from speech_to_text_amd.audio.stft import WhisperSTFT
from speech_to_text_amd.engine import WhisperEngine, whisper_dims
from speech_to_text_amd.weights import random_whisper_weights

for name, n_tok in (("base", 13), ("large-v3", 14)):
    dims = whisper_dims(name)
    eng = WhisperEngine(dims, random_whisper_weights(dims, 0), max_batch=1, max_beams=1,
                        max_text_len=n_tok + 8, max_prompt_len=4)
    stft = WhisperSTFT(n_mel_channels=dims["n_mels"])
    pcm = torch.rand(63360, device="cuda") * 2 - 1
    prompt = np.array([[50258, 50259, 50359]], dtype=np.int32)
    best = [1e9, 1e9, 1e9]
    for it in range(6):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        mel = stft(pcm); torch.cuda.synchronize(); t1 = time.perf_counter()
        eng.encode(mel); torch.cuda.synchronize(); t2 = time.perf_counter()
        eng.generate(prompt, max_new_tokens=n_tok, eos_token=50257, pad_token=50257, early_stopping=False)
        t3 = time.perf_counter()
        if it:
            best = [min(b, v) for b, v in zip(best, (t1 - t0, t2 - t1, t3 - t2))]
    print(f"{name}: mel {1e3*best[0]:.3f} ms, encoder {1e3*best[1]:.2f} ms, {n_tok} greedy tokens "
          f"{1e3*best[2]:.2f} ms ({n_tok/best[2]:.0f} tokens/s), encoder+generation {1e3*(best[1]+best[2]):.2f} ms",
          flush=True)
    del eng
