"""Greedy ids of a fixed synthetic job, as a hash: to compare builds or environment switches that must not
change a result.  python scripts/token_hash.py B T [layers]"""
import hashlib, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from speech_to_text_amd.engine import WhisperEngine, whisper_dims
from speech_to_text_amd.weights import random_whisper_weights
B, T = int(sys.argv[1]), int(sys.argv[2])
dims = whisper_dims("large-v3")
if len(sys.argv) > 3:
    dims = dict(dims, n_audio_layer=int(sys.argv[3]), n_text_layer=int(sys.argv[3]))
eng = WhisperEngine(dims, random_whisper_weights(dims, 0, style="peaked"), max_batch=B, max_beams=1, max_text_len=T + 8,
                    max_prompt_len=4)
mel = torch.rand(B, 3000, dims["n_mels"], device="cuda", generator=torch.Generator(device="cuda").manual_seed(5)) * 2 - 1
prompt = np.array([[50258, 50259, 50359]] * B, dtype=np.int32)
eng.encode(mel)
toks, lens, scores = eng.generate(prompt, max_new_tokens=T, eos_token=50257, pad_token=50257, early_stopping=False)
torch.cuda.synchronize(); t0 = time.perf_counter()
toks, lens, scores = eng.generate(prompt, max_new_tokens=T, eos_token=50257, pad_token=50257, early_stopping=False)
dt = time.perf_counter() - t0
print("B=%d T=%d: ids %s scores %s distinct ids %d, %.3f ms per token" % (
    B, T, hashlib.sha1(toks.tobytes()).hexdigest()[:12], hashlib.sha1(scores.tobytes()).hexdigest()[:12],
    len(np.unique(toks)), dt * 1e3 / T), flush=True)
