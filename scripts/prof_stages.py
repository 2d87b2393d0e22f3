"""Where does `rocprofv3 --kernel-trace` die on the 512-window large-v3 engine?  Runs the engine's life
stage by stage and prints a time-stamped line to stderr before and after each, so the last line
printed names the stage:   python scripts/prof_stages.py [windows=512] [layers=32] [tokens=4]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
T0 = time.time()
def say(msg):
    sys.stderr.write("[%7.2f s] %s\n" % (time.time() - T0, msg)); sys.stderr.flush()
from speech_to_text_amd.engine import WhisperEngine, whisper_dims
from speech_to_text_amd.weights import random_whisper_weights
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
L = int(sys.argv[2]) if len(sys.argv) > 2 else 32
T = int(sys.argv[3]) if len(sys.argv) > 3 else 4
dims = dict(whisper_dims("large-v3"), n_audio_layer=L, n_text_layer=L)
say("weights ...")
w = random_whisper_weights(dims, 0)
say("engine create + finalize (upload, allocate) ...")
eng = WhisperEngine(dims, w, max_batch=B, max_beams=1, max_text_len=T + 8, max_prompt_len=4)
torch.cuda.synchronize(); say("engine built")
from speech_to_text_amd.audio.stft import WhisperSTFT
fn = WhisperSTFT(n_mel_channels=dims["n_mels"])
pcm = torch.rand(B * 480000, device="cuda") * 2 - 1
torch.cuda.synchronize(); say("pcm on the device (%d samples); log-mel ..." % pcm.numel())
mel = fn(pcm)[0].view(B, 3000, dims["n_mels"])
torch.cuda.synchronize(); say("log-mel done; encode ...")
eng.encode(mel)
torch.cuda.synchronize(); say("encoded; generate ...")
prompt = np.array([[50258, 50259, 50359]] * B, dtype=np.int32)
eng.generate(prompt, max_new_tokens=T, eos_token=50257, pad_token=50257, early_stopping=False)
say("generated (token-step graph recorded and replayed %d times); generate again (cached graph) ..." % (T - 2))
if os.environ.get("STAGES_ENCODE_BETWEEN"):
    eng.encode(mel)
    say("encoded again (enqueued); generate ...")
eng.generate(prompt, max_new_tokens=T, eos_token=50257, pad_token=50257, early_stopping=False)
say("generated again; destroying the engine ...")
del eng
torch.cuda.synchronize(); say("done")
