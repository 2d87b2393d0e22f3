"""Generation with the token selection in Python (torch on the GPU), one `stt_decode_step` per token.

The device loop of `stt_generate` covers greedy / beam search with the filters it knows (index
lists, the Whisper timestamp rules).  What it cannot run goes through here instead, one cached
decoder step at a time, like the reference's own eager mode (its Whisper filter branches in Python
and is eager-only there too, models/stt/whisper.py:380-410):

* an arbitrary callable `logits_filter(scores, tokens = ..., state = ..., **kwargs)`
  (architectures/generation_utils.py:518-520);
* `return_logits` under beam search (the per-step log-probabilities gathered by beam index,
  generation_utils.py:422-428, 551-561);
* `return_attention` / `return_last_attention` (the attention weights of every / the last decoder
  layer, accumulated over the steps: generation_utils.py:576-610);
* `return_state` and `initial_state` (the decoder state after the loop, and a loop that starts
  from it: generation_utils.py:298, 478).

The decoder pass is the engine's (same kernels, same K/V cache, re-linked instead of gathered for
beam search); only scaling, filtering, log-softmax, selection and bookkeeping run as torch ops on
the logits tensor the step returns.  Semantics follow generation_utils.py:195-278 (greedy body),
:333-438 (beam body), :484-548 (process_logits, select_next_token); the beam index uses
`b * num_beams`, which is the reference's at batch 1.
"""
import collections
import collections.abc

import numpy as np
import torch

# what the reference hands a filter as `state` (generation_utils.py:578-586): the step counter, the
# finished flags and the decoder state, which is None before the first step (whisper.py:355, 415)
LoopState = collections.namedtuple("LoopState", ["t", "finished", "state", "padding_mask"])


class EngineState(collections.abc.Mapping):
    """What `InferenceOutput.state` holds: the decoder state after a loop, per layer name
    ('layer_<i>', architectures/transformers/transformer_arch.py:560, 693-694) the pair ((k, v) of
    the self-attention, (k, v) of the encoder attention) as float32 CUDA tensors [n, heads, length,
    64].  The numbers stay in the engine's own cache (bf16, beam histories linked rather than
    gathered) and are copied out when a layer is read; the same object passed back as
    `initial_state` continues the loop from that cache without running anything again.  Both work
    only until the engine is used for something else (`valid()`)."""

    def __init__(self, engine, beam_index=None):
        self.engine, self.epoch = engine, engine._epoch
        self.n_seq, self.seq_per_window, self.length = engine.step_state()
        # the last selection of a beam search has not reached the cache (it is re-linked at the next
        # step): applied to the copies instead (generation_utils.py:433-436)
        self.beam_index = beam_index

    def valid(self):
        return self.engine._epoch == self.epoch

    def _layers(self):
        return ["layer_%d" % i for i in range(self.engine.dims["n_text_layer"])]

    def __iter__(self):
        return iter(self._layers())

    def __len__(self):
        return self.engine.dims["n_text_layer"]

    def __getitem__(self, name):
        if name not in self._layers():
            raise KeyError(name)
        if not self.valid():
            raise RuntimeError("the engine has been used since this state was returned: its cache now "
                               "holds something else")
        (k, v), enc = self.engine.export_state(int(name.split("_")[1]))
        if self.beam_index is not None:
            k, v = k[self.beam_index], v[self.beam_index]
        return (k, v), enc


def _gather_concat(prev, new, parents):
    """generation_utils.py:576-610 for one step: both sides follow the beam index, self-attention
    rows grow by the new position, rows of the new tokens are appended."""
    out = {}
    for key, cur in new.items():
        if parents is not None:
            cur = cur[parents]
        if prev is None:
            out[key] = cur
            continue
        old = prev[key][parents] if parents is not None else prev[key]
        if "enc" not in key:
            old = torch.nn.functional.pad(old, (0, cur.shape[-1] - old.shape[-1]))
        out[key] = torch.cat([old, cur], dim=-2)
    return out


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
             filter_kwargs=None, return_attention=False, return_last_attention=False, return_state=False,
             initial_state=None, extras=None):
    """Same results layout as `WhisperEngine.generate`: numpy tokens / lengths / scores ([B, steps],
    [B], [B] for greedy; [B, num_sentences, ...] for beam search), plus the per-step log-probabilities
    as a CUDA tensor ([B, steps, V] or [B, num_sentences, steps, V]) when `return_logits`.
    The engine must hold the encoded batch (`engine.encode`).
    `return_attention` / `return_last_attention`: `extras['attention_weights']` becomes the dict
    {'attn_layer_<i>': [n, heads, tokens fed, positions], 'enc_attn_layer_<i>': [n, heads, tokens fed,
    1500]} of every / the last decoder layer (n = B x beams, rows following the beam index).
    `return_state`: `extras['state']` becomes an EngineState.  `initial_state`: an EngineState this
    engine returned and has not been used since; `prompt` then holds only the tokens that follow its
    history (greedy / sampling; the beams of a beam search would also need their scores)."""
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

    history = 0
    if initial_state is not None:
        if not isinstance(initial_state, EngineState) or initial_state.engine is not engine or not initial_state.valid():
            raise ValueError("initial_state must be the state this engine returned last, with nothing run on the "
                             "engine in between")
        if beam or initial_state.beam_index is not None or initial_state.n_seq != n:
            raise ValueError("initial_state continues a greedy / sampling loop over the same %d sequences"
                             % initial_state.n_seq)
        history = initial_state.length
    which = "all" if return_attention else "last" if return_last_attention else None
    first_layer = 0 if return_attention else engine.dims["n_text_layer"] - 1
    weights = None

    inputs = torch.from_numpy(prompt).to(dev).repeat_interleave(nb, dim=0).contiguous()
    generated = torch.full((n, steps), int(pad_token), dtype=torch.long, device=dev)
    lengths = torch.full((n,), history + s, dtype=torch.long, device=dev)
    scores = torch.zeros(n, dtype=torch.float32, device=dev)
    finished = torch.zeros(n, dtype=torch.bool, device=dev)
    kept = torch.full((n, steps, v), float("-inf"), dtype=torch.float32, device=dev) if return_logits else None
    pad_row = torch.full((1, v), float("-inf"), dtype=torch.float32, device=dev)
    pad_row[0, int(pad_token)] = 0.0
    window_base = torch.arange(b, device=dev).repeat_interleave(nb) * nb
    parents = None
    for t in range(steps):
        fresh = t == 0 and initial_state is None
        logits = engine.decode_step(inputs, seq_per_window=nb, reset=fresh, parents=parents, attention=which)
        step_weights = None
        if which is not None:
            logits, own, enc = logits
            step_weights = {}
            for j in range(own.shape[0]):
                step_weights["attn_layer_%d" % (first_layer + j)] = own[j]
                step_weights["enc_attn_layer_%d" % (first_layer + j)] = enc[j]
        state = LoopState(t=t, finished=finished, state=None if fresh else engine, padding_mask=None)
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
            if step_weights is not None:
                weights = _gather_concat(weights, step_weights, parents)
            parents = parents.to(torch.int32).contiguous()
        if step_weights is not None and not beam:
            weights = _gather_concat(weights, step_weights, None)
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
    lens = (lengths - s - history).view(b, nb)[:, :nsent].to(torch.int32).cpu().numpy()
    if extras is not None:
        extras["attention_weights"] = weights
        extras["state"] = EngineState(engine, parents.long() if beam and parents is not None else None) \
            if return_state else None
    scs = scores.view(b, nb)[:, :nsent].cpu().numpy()
    if kept is not None:
        kept = kept.view(b, nb, steps, v)[:, :nsent]
    if not beam:
        out = (tokens[:, 0], lens[:, 0], scs[:, 0])
        return out + (kept[:, 0],) if return_logits else out
    return (tokens, lens, scs, kept) if return_logits else (tokens, lens, scs)
