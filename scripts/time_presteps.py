"""File -> mel with the pre-steps on the host (reference order: read, normalise, trim, then upload)
against the device pipeline (upload int16 once, everything else on the GPU)."""
import os, sys, time
import numpy as np, torch
from scipy.io import wavfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from speech_to_text_amd.audio.audio_processing import WHISPER_TRIM_KWARGS, read_audio
from speech_to_text_amd.audio.device_pipeline import DeviceAudioPipeline
from speech_to_text_amd.audio.stft import WhisperSTFT

fn, pipe = WhisperSTFT(), DeviceAudioPipeline()
kw = dict(normalize=True, trim_silence=True, **WHISPER_TRIM_KWARGS)
rng = np.random.default_rng(0)
long_path = "/tmp/clip10min.wav"
x = (rng.standard_normal(16000 * 600) * 5000).astype(np.int16)
for a in range(0, len(x), 16000 * 20):
    x[a: a + 16000 * 2] //= 3000
wavfile.write(long_path, 16000, x)
for name, path in (("golden 4 s clip", os.path.join(ROOT, "tests", "golden", "audio_test.wav")), ("10 min clip", long_path)):
    for label, run in (("host pre-steps", lambda: fn(read_audio(path, rate=16000, **kw)[1])),
                       ("device pipeline", lambda: fn(pipe(path, **kw)[1]))):
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            out = run()
        torch.cuda.synchronize()
        print("%-16s %-16s %8.2f ms per file -> mel %s" % (name, label, (time.perf_counter() - t0) * 100, tuple(out.shape)))
