import numpy as np, torch, sys
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import frontend as oracle
from speech_to_text_amd.audio.stft import WhisperSTFT
fn = WhisperSTFT()
for n in [64880, 160*247*2+50]:
    rng = np.random.default_rng(0)
    a = rng.uniform(-1, 1, n).astype(np.float32)
    got = fn(a)[0].cpu().numpy()
    ref = oracle.whisper_log_mel(a)
    d = np.abs(got - ref)
    bad = np.argwhere(d > 1e-4)
    print(n, got.shape, 'max', d.max(), 'nbad', len(bad))
    rows = np.unique(bad[:, 0]); cols = np.unique(bad[:, 1])
    print('bad rows', rows[:50], len(rows)); print('bad cols', cols[:80], len(cols))
    if len(bad):
        r = rows[0]
        print('row', r, 'got', got[r, :8], 'ref', ref[r, :8])
    print('nan', np.isnan(got).sum())
