"""Second half of __graft_entry__.smoke(): a tiny Whisper (seeded synthetic weights) through the
engine, checked against the torch oracle (teacher-forced logits and greedy tokens)."""
import numpy as np
import torch


def run():
    from oracle import model as om
    from speech_to_text_amd.engine import WhisperEngine
    dims = om.Dims(n_mels=80, n_audio_ctx=1500, n_audio_state=128, n_audio_head=2, n_audio_layer=2,
                   n_vocab=1000, n_text_ctx=64, n_text_state=128, n_text_head=2, n_text_layer=2)
    w = {k: v.to(torch.bfloat16).float() for k, v in om.random_weights(dims, seed=0, style="peaked").items()}
    keys = ("n_mels", "n_audio_ctx", "n_audio_state", "n_audio_head", "n_audio_layer", "n_vocab",
            "n_text_ctx", "n_text_state", "n_text_head", "n_text_layer")
    eng = WhisperEngine({k: getattr(dims, k) for k in keys}, w, max_batch=2, max_beams=1,
                        max_text_len=24, max_prompt_len=4)
    oracle = om.Whisper(dims, w)
    mel = torch.rand(2, 3000, 80, generator=torch.Generator().manual_seed(1)) * 2 - 1
    enc = eng.encode(mel, return_output=True).cpu()
    prompt = np.array([[1, 2, 3]] * 2, dtype=np.int32)
    toks, lens, _ = eng.generate(prompt, max_new_tokens=8, eos_token=999, pad_token=999, early_stopping=False)
    with torch.no_grad():
        ref_enc = oracle.encoder(mel)
        rel = float((enc - ref_enc).norm() / ref_enc.norm())
        assert rel < 1e-2, "encoder mismatch vs oracle: rel L2 %g" % rel
        full = torch.cat([torch.from_numpy(prompt).long(), torch.from_numpy(toks).long()], dim=1)
        logits, _ = oracle.decoder(full[:, :-1], oracle.cross_kv(ref_enc))
        lp = torch.log_softmax(logits[:, 2:], dim=-1)
        best = lp.max(dim=-1).values
        mine = lp.gather(2, torch.from_numpy(toks).long()[:, :, None])[:, :, 0]
        # near-tie allowance: 0.04 x the standard deviation of the logits row
        assert bool(((best - mine) <= 0.04 * logits[:, 2:].std(dim=-1)).all()), "greedy tokens are not the oracle's argmax"
    print("smoke ok: encoder rel L2 %.2e, %d greedy tokens agree with the oracle" % (rel, toks.size))
