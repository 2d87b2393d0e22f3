"""Per-lane dump of the two-lanes-per-frame log-mel kernel (first tile) against the numpy
dataflow of csrc/gen_logmel2.py: sub-DFT outputs, magnitudes, projection sums."""
import ctypes as C, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "speech_to_text_amd", "csrc"))
import gen_logmel2 as G
from gen_logmel import live_nodes, evaluate
from speech_to_text_amd import _lib
from speech_to_text_amd.audio.stft import WhisperSTFT
from speech_to_text_amd.audio.mel_basis import mel_filterbank

fn = WhisperSTFT()
rng = np.random.default_rng(0)
n = 160 * 400
x = rng.standard_normal(n).astype(np.float32) * 0.1
dbg = torch.zeros(256 * 512, device="cuda")
lib = _lib.load()
lib.stt_logmel_set_debug.argtypes = [C.c_void_p]
lib.stt_logmel_set_debug(C.c_void_p(dbg.data_ptr()))
got = fn(x)[0].float().cpu().numpy()
torch.cuda.synchronize()
lib.stt_logmel_set_debug(C.c_void_p(0))
d = dbg.cpu().numpy().reshape(256, 512)

hann = (0.5 - 0.5 * np.cos(2.0 * np.pi * np.arange(400) / 400)).astype(np.float32).astype(np.float64)
pad = np.pad(x.astype(np.float64), 200, mode="reflect")
g, bins, subs = G.build_fft()
roots = [r for b in bins for r in (b.re.n, b.im.n) if r is not None]
live = live_nodes(g, roots)
# boundary order as in the generator
boundary, n_sa = {}, 0
for j in range(4):
    rts = [r for b in subs[j] for r in (b.re.n, b.im.n) if r is not None]
    for nn in sorted(set(rts) & live):
        if nn not in boundary:
            boundary[nn] = n_sa; n_sa += 1
for frame in (3, 40, 100):
    wave, l = frame // 32, frame % 32
    fr = pad[frame * 160: frame * 160 + 400] * hann
    e, o = fr[0::2], fr[1::2]
    for half in (0, 1):
        tid = wave * 64 + half * 32 + l
        ve, vo = evaluate(g, live, e[half::2]), evaluate(g, live, o[half::2])
        want = np.zeros(200)
        for nn, sl in boundary.items():
            want[2 * sl], want[2 * sl + 1] = ve[nn], vo[nn]
        err = np.abs(d[tid, :200] - want)
        print("frame %d half %d: sub-DFT outputs max err %.3g (scale %.3g) worst slot %d" % (frame, half, err.max(), np.abs(want).max(), err.argmax() // 2))
    spec = np.abs(np.fft.rfft(fr))
    for half in (0, 1):
        tid = wave * 64 + half * 32 + l
        bad = []
        for i, (_, kl, ku) in enumerate(G.kappa_slots()):
            kap = kl if half == 0 else ku
            if kap is None: continue
            for c, b in enumerate(G.bins_of(kap)):
                if abs(d[tid, 200 + 4 * i + c] - spec[b]) > 1e-3 * spec.max():
                    bad.append((i, c, b, float(d[tid, 200 + 4 * i + c]), float(spec[b])))
        print("frame %d half %d: magnitudes wrong %d of 104" % (frame, half, len(bad)), bad[:6])
fb = mel_filterbank(16000, 400, 80, 0.0, 8000.0)
for frame in (3, 40):
    wave, l = frame // 32, frame % 32
    fr = pad[frame * 160: frame * 160 + 400] * hann
    mel = np.abs(np.fft.rfft(fr)) @ fb
    lo, up = d[wave * 64 + l, 304:304 + 48], d[wave * 64 + 32 + l, 304:304 + 32]
    print("frame %d: projection err lower %.3g upper %.3g (scale %.3g)" % (frame, np.abs(lo - mel[:48]).max(), np.abs(up - mel[48:]).max(), mel.max()))
