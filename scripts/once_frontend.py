import torch, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from speech_to_text_amd.audio.stft import WhisperSTFT
n = 57_600_000
fn = WhisperSTFT()
pcm = (torch.rand(n, device='cuda') * 2 - 1).float()
out = fn(pcm); torch.cuda.synchronize()
print("second call"); out = fn(pcm); torch.cuda.synchronize()
