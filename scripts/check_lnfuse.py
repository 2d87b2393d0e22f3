"""The decode step with the LayerNorms fused into the projections (rows <= 4) against the separate
kernels (STT_MI355X_NO_LNFUSE=1, read once per process): run twice, compare the printed digests."""
import hashlib, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from speech_to_text_amd.engine import WhisperEngine, whisper_dims
from speech_to_text_amd.weights import random_whisper_weights
name = sys.argv[1] if len(sys.argv) > 1 else "base"
dims = whisper_dims(name)
eng = WhisperEngine(dims, random_whisper_weights(dims, 0), max_batch=2, max_beams=2, max_text_len=40, max_prompt_len=4)
g = torch.Generator().manual_seed(1)
mel = (torch.rand(2, 3000, dims["n_mels"], generator=g) * 2 - 1).cuda()
eng.encode(mel)
prompt = np.array([[50258, 50259, 50359]] * 2, dtype=np.int32)
h = hashlib.sha256()
for b in (1, 2):
    eng.encode(mel[:b])
    t, l, s = eng.generate(prompt[:b], max_new_tokens=24, eos_token=50257, pad_token=50257, early_stopping=False)
    h.update(np.ascontiguousarray(t).tobytes()); h.update(np.ascontiguousarray(s).tobytes())
    t, l, s = eng.generate(prompt[:b], method="beam", num_beams=2, num_sentences=1, max_new_tokens=12, eos_token=50257,
                           pad_token=50257, early_stopping=False)
    h.update(np.ascontiguousarray(t).tobytes()); h.update(np.ascontiguousarray(s).tobytes())
lg = eng.decoder_logits(np.array([[50258]], dtype=np.int32)) if False else None
print(name, "digest", h.hexdigest()[:16], "last scores", s.ravel()[:2])
