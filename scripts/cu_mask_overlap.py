"""Can an encoder pass (matrix-core bound) and a generation (HBM bound) share the chip if each gets its own
compute units?  Two engines of B windows (large-v3); one encodes on a stream whose CU mask holds `enc_share`
tenths of the CUs while the other generates on a stream with the rest.  Prints each alone on the whole chip,
each alone on its share, and both at once.
python scripts/cu_mask_overlap.py B tokens enc_share_tenths [layers]"""
import ctypes, os, sys, threading, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from speech_to_text_amd.engine import WhisperEngine, whisper_dims
from speech_to_text_amd.weights import random_whisper_weights

B, T, share = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
dims = whisper_dims("large-v3")
if len(sys.argv) > 4:
    dims = dict(dims, n_audio_layer=int(sys.argv[4]), n_text_layer=int(sys.argv[4]))
torch.cuda.init()
hip = ctypes.CDLL("libamdhip64.so")
n_cu = torch.cuda.get_device_properties(0).multi_processor_count


def masked_stream(keep):
    """A stream restricted to the CUs i with keep(i); bits spread evenly whatever the bit -> (XCD, CU) order."""
    words = [0] * ((n_cu + 31) // 32)
    for i in range(n_cu):
        if keep(i):
            words[i // 32] |= 1 << (i % 32)
    arr = (ctypes.c_uint32 * len(words))(*words)
    s = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(s), len(words), arr)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(s.value), sum(bin(w).count("1") for w in words)


enc_stream, n_enc = masked_stream(lambda i: i % 10 < share)
gen_stream, n_gen = masked_stream(lambda i: i % 10 >= share)
w = random_whisper_weights(dims, 0, style="peaked")
engs = [WhisperEngine(dims, w, max_batch=B, max_beams=1, max_text_len=T + 8, max_prompt_len=4) for _ in range(2)]
mel = torch.rand(B, 3000, dims["n_mels"], device="cuda") * 2 - 1
prompt = np.array([[50258, 50259, 50359]] * B, dtype=np.int32)
kw = dict(max_new_tokens=T, eos_token=50257, pad_token=50257, early_stopping=False)
for e in engs:
    e.encode(mel)
    e.generate(prompt, **kw)
torch.cuda.synchronize()


def timed(fn):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3


def enc_on(stream, e):
    with torch.cuda.stream(stream):
        e.encode(mel)
        stream.synchronize()


def gen_on(stream, e):
    with torch.cuda.stream(stream):
        e.generate(prompt, **kw)


full = torch.cuda.current_stream()
t_enc = timed(lambda: enc_on(full, engs[1]))
t_gen = timed(lambda: gen_on(full, engs[0]))
gen_on(gen_stream, engs[0])            # records the token-step graph for this stream's first use
t_enc_m = timed(lambda: enc_on(enc_stream, engs[1]))
t_gen_m = timed(lambda: gen_on(gen_stream, engs[0]))


def both():
    th = threading.Thread(target=enc_on, args=(enc_stream, engs[1]))
    th.start()
    gen_on(gen_stream, engs[0])
    th.join()


t_both = min(timed(both), timed(both))
print("B=%d T=%d: whole chip encode %.0f ms, generate %.0f ms (sum %.0f); encode on %d CUs %.0f ms, generate on %d CUs "
      "%.0f ms; both at once %.0f ms" % (B, T, t_enc, t_gen, t_enc + t_gen, n_enc, t_enc_m, n_gen, t_gen_m, t_both), flush=True)
