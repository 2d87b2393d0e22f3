"""Generation with the token selection in Python (torch on the GPU), one `stt_decode_step` per token.

The device loop of `stt_generate` covers greedy / beam search with the filters it knows (index
lists, the Whisper timestamp rules).  What it cannot run goes through here instead, one cached
decoder step at a time, like the reference's own eager mode (its Whisper filter branches in Python
and is eager-only there too, models/stt/whisper.py:380-410):

* an arbitrary callable `logits_filter(scores, tokens = ..., state = ..., **kwargs)`
  (architectures/generation_utils.py:518-520);
* `return_logits` under beam search (the per-step log-probabilities gathered by beam index,
  generation_utils.py:422-428, 551-561).

The decoder pass is the engine's (same kernels, same K/V cache, re-linked instead of gathered for
beam search); only scaling, filtering, log-softmax, selection and bookkeeping run as torch ops on
the logits tensor the step returns.  Semantics follow generation_utils.py:195-278 (greedy body),
:333-438 (beam body), :484-548 (process_logits, select_next_token); the beam index uses
`b * num_beams`, which is the reference's at batch 1.
"""
import collections

import numpy as np
import torch

# what the reference hands a filter as `state` (generation_utils.py:578-586): the step counter, the
# finished flags and the decoder state, which is None before the first step (whisper.py:355, 415)
LoopState = collections.namedtuple("LoopState", ["t", "finished", "state", "padding_mask"])


def _process(logits, lengths, generated, state, temperature, length_temperature, logits_filter, extra):
    """generation_utils.py:484-524 on a [n, V] float32 CUDA tensor."""
    z = logits
    if temperature:
        z = z / float(temperature)
    if length_temperature is not None:
        z = z * ((lengths + 1).to(z.dtype) ** float(length_temperature))[:, None]
    if logits_filter is not None:
        if callable(logits_filter):
            z = logits_filter(z, tokens=generated, state=state, **extra)
            if not isinstance(z, torch.Tensor):
                z = torch.as_tensor(np.asarray(z), device=logits.device)
            z = z.to(logits.dtype)
        else:
            idx = torch.as_tensor(np.asarray(logits_filter, dtype=np.int64).reshape(-1), device=z.device)
            z = z.index_fill(1, idx, float("-inf"))
    return torch.log_softmax(z, dim=-1)


def generate(engine, prompt, *, method="greedy", num_beams=1, num_sentences=1, max_new_tokens,
             eos_token, pad_token, early_stopping=True, temperature=None, length_temperature=None,
             length_power=None, logits_filter=None, return_logits=False, seed=None, check_every=8,
             filter_kwargs=None):
    """Same results layout as `WhisperEngine.generate`: numpy tokens / lengths / scores ([B, steps],
    [B], [B] for greedy; [B, num_sentences, ...] for beam search), plus the per-step log-probabilities
    as a CUDA tensor ([B, steps, V] or [B, num_sentences, steps, V]) when `return_logits`.
    The engine must hold the encoded batch (`engine.encode`)."""
    prompt = np.ascontiguousarray(np.asarray(prompt, dtype=np.int32))
    if prompt.ndim == 1:
        prompt = prompt[None]
    b, s = prompt.shape
    beam = method in ("beam", "beam_search")
    if not beam and method not in ("greedy", "sample"):
        raise ValueError("unsupported inference method: {}".format(method))
    nb = int(num_beams) if beam else 1
    nsent = int(num_sentences) if beam else 1
    steps, n = int(max_new_tokens), b * nb
    dev = engine.device
    extra = dict(filter_kwargs or {})
    v = engine.dims["n_vocab"]
    gen = None
    if temperature:
        gen = torch.Generator(device=dev)
        gen.manual_seed(int(seed) & (2 ** 63 - 1)) if seed is not None else gen.seed()

    inputs = torch.from_numpy(prompt).to(dev).repeat_interleave(nb, dim=0).contiguous()
    generated = torch.full((n, steps), int(pad_token), dtype=torch.long, device=dev)
    lengths = torch.full((n,), s, dtype=torch.long, device=dev)
    scores = torch.zeros(n, dtype=torch.float32, device=dev)
    finished = torch.zeros(n, dtype=torch.bool, device=dev)
    kept = torch.full((n, steps, v), float("-inf"), dtype=torch.float32, device=dev) if return_logits else None
    pad_row = torch.full((1, v), float("-inf"), dtype=torch.float32, device=dev)
    pad_row[0, int(pad_token)] = 0.0
    window_base = torch.arange(b, device=dev).repeat_interleave(nb) * nb
    parents = None
    for t in range(steps):
        logits = engine.decode_step(inputs, seq_per_window=nb, reset=(t == 0), parents=parents)
        state = LoopState(t=t, finished=finished, state=None if t == 0 else engine, padding_mask=None)
        lp = _process(logits, lengths, generated, state, temperature, length_temperature, logits_filter, extra)
        if not beam:
            if temperature:
                tok = torch.multinomial(torch.softmax(lp, dim=-1), 1, generator=gen)[:, 0]
            else:
                tok = lp.argmax(dim=-1)
            gain = lp.gather(1, tok[:, None])[:, 0]
            tok = torch.where(finished, torch.full_like(tok, int(pad_token)), tok)
            scores = scores + torch.where(finished, torch.zeros_like(gain), gain)
            if kept is not None:
                kept[:, t] = lp
        else:
            with_scores = scores[:, None] + torch.where(finished[:, None], pad_row, lp)
            ranked = with_scores
            if length_power is not None:
                ranked = ranked / ((lengths + 1).to(ranked.dtype)[:, None] ** float(length_power))
            ranked = ranked.reshape(b, nb * v)
            if t == 0:
                ranked = ranked[:, :v]              # every beam holds the same prompt: one row competes
            if temperature:
                flat = torch.multinomial(torch.softmax(ranked, dim=-1), nb, replacement=True, generator=gen)
            else:
                flat = torch.topk(ranked, nb, dim=1).indices
            flat = flat.reshape(n)
            parents = (flat // v + window_base)
            tok = flat % v
            lengths, finished = lengths[parents], finished[parents]
            scores = with_scores[parents, tok]
            tok = torch.where(finished, torch.full_like(tok, int(pad_token)), tok)
            generated = generated[parents]
            if kept is not None:
                kept = kept[parents]
                kept[:, t] = lp[parents]
            parents = parents.to(torch.int32).contiguous()
        finished = finished | (tok == int(eos_token))
        lengths = lengths + (~finished).long()
        generated[:, t] = tok
        inputs = tok[:, None].to(torch.int32).contiguous()
        # beam search stops exactly where the reference does (unfinished lower beams must not
        # overtake afterwards) and so does a call that keeps the per-step scores; plain greedy
        # results cannot change after the end, so it asks the device only every few steps
        if early_stopping and (beam or return_logits or (t + 1) % max(1, int(check_every)) == 0):
            if bool(finished.view(b, nb)[:, :nsent].all()):
                break
    tokens = generated.view(b, nb, steps)[:, :nsent].to(torch.int32).cpu().numpy()
    lens = (lengths - s).view(b, nb)[:, :nsent].to(torch.int32).cpu().numpy()
    scs = scores.view(b, nb)[:, :nsent].cpu().numpy()
    if kept is not None:
        kept = kept.view(b, nb, steps, v)[:, :nsent]
    if not beam:
        out = (tokens[:, 0], lens[:, 0], scs[:, 0])
        return out + (kept[:, 0],) if return_logits else out
    return (tokens, lens, scs, kept) if return_logits else (tokens, lens, scs)
