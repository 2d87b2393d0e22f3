"""Which DFT bins does the log-mel kernel get wrong?  One clip per bin b: a tone at b * 40 Hz
(exactly bin b of the 400-point frame), compared with the numpy oracle frame by frame."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import frontend as oracle
from speech_to_text_amd.audio.stft import WhisperSTFT
n_mels = int(sys.argv[1]) if len(sys.argv) > 1 else 80
fn = WhisperSTFT(n_mel_channels=n_mels)
n = 160 * 300
t = np.arange(n)
bad = []
for b in range(0, 201):
    x = (0.5 * np.cos(2 * np.pi * b * t / 400.0 + 0.3) + 0.01 * np.cos(2 * np.pi * 3.3 * t / 400.0)).astype(np.float32)
    got = fn(x)[0].float().cpu().numpy()
    ref = oracle.whisper_log_mel(x, n_mels=n_mels)
    err = np.abs(got - ref)
    # skip the frames that touch the reflected ends
    e = err[5:-5]
    if e.max() > 1e-4:
        f = np.unravel_index(e.argmax(), e.shape)
        bad.append((b, float(e.max()), int(f[0]) + 5, int(f[1]), float((e.max(axis=1) > 1e-4).mean())))
print("bins wrong:", len(bad))
for r in bad[:60]:
    print("bin %3d  max err %.4f at frame %d filter %d  frames wrong %.2f" % r)
