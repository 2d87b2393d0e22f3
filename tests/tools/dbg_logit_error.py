"""Engine vs oracle teacher-forced logits on peaked weights: how large is the bf16 error in nats?"""
import dataclasses, sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import model as om
from tests import _parity as P
from speech_to_text_amd.engine import WhisperEngine

def run(name, dims, B, seed, style):
    w = P.bf16_round(om.random_weights(dims, seed=seed, style=style))
    eng = WhisperEngine(P.dims_dict(dims), w, max_batch=B, max_beams=1, max_text_len=40, max_prompt_len=24)
    o = om.Whisper(dims, w)
    ob = om.Whisper(dims, w, round_acts=True)
    g = torch.Generator().manual_seed(1)
    mel = torch.rand(B, 3000, dims.n_mels, generator=g) * 2 - 1
    with torch.no_grad():
        enc = o.encoder(mel)
    got_enc = eng.encode(mel, return_output=True).cpu()
    toks = np.random.default_rng(3).integers(0, dims.n_vocab, size=(B, 16)).astype(np.int32)
    got = eng.decoder_logits(toks).cpu()
    with torch.no_grad():
        ref, _ = o.decoder(torch.from_numpy(toks).long(), o.cross_kv(enc))
        ref2, _ = o.decoder(torch.from_numpy(toks).long(), o.cross_kv(got_enc))
        encb = ob.encoder(mel)
        ref3, _ = ob.decoder(torch.from_numpy(toks).long(), ob.cross_kv(encb))
    print("   bf16-emulating oracle: enc relL2 %.2e (fp32 oracle vs it %.2e)" % (P.rel_l2(got_enc.numpy(), encb.numpy()), P.rel_l2(encb.numpy(), enc.numpy())))
    for tag, r in (("oracle-enc", ref), ("bf16-emul", ref3)):
        lg, lr = torch.log_softmax(got, -1), torch.log_softmax(r, -1)
        top = lr.topk(5, -1).indices
        d = (lg.gather(-1, top) - lr.gather(-1, top))
        # error of pairwise differences among the top-5
        dd = (d[..., :1] - d[..., 1:]).abs()
        print("%s %s %s: enc relL2 %.2e  logits std %.2f relL2 %.2e maxabs %.3f | top5 logprob err rms %.3f max %.3f | pairwise diff err rms %.3f max %.3f" % (
            name, style, tag, P.rel_l2(got_enc.numpy(), enc.numpy()), float(r.std()), P.rel_l2(got.numpy(), r.numpy()),
            float((got - r).abs().max()), float(d.pow(2).mean().sqrt()), float(d.abs().max()),
            float(dd.pow(2).mean().sqrt()), float(dd.max())), flush=True)

small = om.Dims(n_mels=80, n_audio_ctx=1500, n_audio_state=128, n_audio_head=2, n_audio_layer=2, n_vocab=1000, n_text_ctx=64, n_text_state=128, n_text_head=2, n_text_layer=2)
for style in ("plain", "peaked"):
    run("small", small, 3, 0, style)
    run("base", om.Dims.named("base"), 2, 11, style)
    run("wide2", dataclasses.replace(om.Dims.named("large-v3"), n_audio_layer=2, n_text_layer=2), 2, 7, style)
