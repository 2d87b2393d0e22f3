"""CPU oracle (torch float32) for the Whisper encoder / decoder / generation path.

TEST INFRASTRUCTURE ONLY: imported by tests, the smoke check and the benchmark's
cpu_baseline leg, never by the product package.

PARITY UNPINNED for everything in this file: the reference holds no test, golden
vector or fixture for the encoder, decoder, KV cache, greedy or beam search
(`Grep infer|Whisper|beam|greedy tests/` in the reference finds nothing), its
arithmetic lives in Keras 3 (not installed here, version unpinned) and no weights
ship with it.  What follows restates, from the reference's source text, the
computation of
  architectures/transformers/whisper_arch.py:39-104, 121-155   (encoder stem, sinusoids)
  architectures/current_blocks.py:297-313, 338-339             (zero-pad + valid conv + act)
  architectures/layers/residual_multi_head_attention.py:167-315 (pre-norm MHA, KV state)
  architectures/transformers/transformer_arch.py:385-503, 593-597 (layer, final norm)
  architectures/transformers/text_transformer_arch.py:154-193, 399-443 (token + position embedding)
  architectures/transformers/gpt2_arch.py:98-101               (tied-embedding logits)
  architectures/generation_utils.py:57-548, 652-823            (greedy, beam, logits processing)
  models/stt/whisper.py:354-416                                (timestamp / logits filter)
and is cross-checked structurally against `transformers.WhisperModel` with the same
random weights in tests/test_oracle_model.py (that model is not the reference either).

Weights use OpenAI Whisper names (what `models/weights_converter.py` starts from):
  encoder.conv{1,2}.{weight,bias}                 weight [out, in, 3]
  encoder.blocks.N.attn.{query,key,value,out}.{weight,bias}   weight [out, in], key has no bias
  encoder.blocks.N.{attn_ln,mlp_ln}.{weight,bias}
  encoder.blocks.N.mlp.{0,2}.{weight,bias}
  encoder.ln_post.{weight,bias}
  decoder.token_embedding.weight [V, d]; decoder.positional_embedding [n_ctx, d]
  decoder.blocks.N.{attn,cross_attn}.*, {attn_ln,cross_attn_ln,mlp_ln}.*, mlp.{0,2}.*
  decoder.ln.{weight,bias}
"""
import math
from dataclasses import dataclass

import numpy as np
import torch
import torch.nn.functional as F


@dataclass
class Dims:
    n_mels: int = 80
    n_audio_ctx: int = 1500
    n_audio_state: int = 384
    n_audio_head: int = 6
    n_audio_layer: int = 4
    n_vocab: int = 51865
    n_text_ctx: int = 448
    n_text_state: int = 384
    n_text_head: int = 6
    n_text_layer: int = 4

    @staticmethod
    def named(name):
        table = {
            "tiny": (80, 384, 6, 4, 51865), "base": (80, 512, 8, 6, 51865),
            "small": (80, 768, 12, 12, 51865), "medium": (80, 1024, 16, 24, 51865),
            "large-v3": (128, 1280, 20, 32, 51866),
        }
        m, d, h, l, v = table[name]
        return Dims(m, 1500, d, h, l, v, 448, d, h, l)


def random_weights(dims, seed=0, dtype=torch.float32, style="plain"):
    """Seeded synthetic weights of the right shapes (no checkpoint ships with the
    reference).  Scaled so that activations stay O(1) through the stack.

    style="plain": with the tied embedding at std 0.3 the input token dominates its own logit,
    so a greedy decode repeats the last prompt token whatever the audio - fine for encoder
    states and teacher-forced logits, useless as a generation test.
    style="peaked": same draws, rescaled afterwards (small embedding and positions, final
    LayerNorm gain 6, every query matrix x4 for sharp attention, decoder attention outputs x2,
    rows of every second MLP matrix centred so that the positive mean of the GELU outputs does
    not add the same vector at every position of a deep stack) so that the next token depends
    on the audio (cross-attention) and on the history (self-attention) and the winner usually
    leads by about a nat."""
    g = torch.Generator().manual_seed(seed)
    w = {}

    def lin(name, out, inp, bias=True, scale=1.0):
        w[name + ".weight"] = torch.randn(out, inp, generator=g) * (scale / math.sqrt(inp))
        if bias:
            w[name + ".bias"] = torch.randn(out, generator=g) * 0.02

    def ln(name, d):
        w[name + ".weight"] = 1.0 + 0.1 * torch.randn(d, generator=g)
        w[name + ".bias"] = 0.05 * torch.randn(d, generator=g)

    d = dims.n_audio_state
    w["encoder.conv1.weight"] = torch.randn(d, dims.n_mels, 3, generator=g) / math.sqrt(3 * dims.n_mels)
    w["encoder.conv1.bias"] = torch.randn(d, generator=g) * 0.02
    w["encoder.conv2.weight"] = torch.randn(d, d, 3, generator=g) / math.sqrt(3 * d)
    w["encoder.conv2.bias"] = torch.randn(d, generator=g) * 0.02
    for i in range(dims.n_audio_layer):
        p = "encoder.blocks.%d." % i
        lin(p + "attn.query", d, d); lin(p + "attn.key", d, d, bias=False)
        lin(p + "attn.value", d, d); lin(p + "attn.out", d, d, scale=0.5)
        ln(p + "attn_ln", d); ln(p + "mlp_ln", d)
        lin(p + "mlp.0", 4 * d, d); lin(p + "mlp.2", d, 4 * d, scale=0.5)
    ln("encoder.ln_post", d)
    d = dims.n_text_state
    w["decoder.token_embedding.weight"] = torch.randn(dims.n_vocab, d, generator=g) * 0.3
    w["decoder.positional_embedding"] = torch.randn(dims.n_text_ctx, d, generator=g) * 0.1
    for i in range(dims.n_text_layer):
        p = "decoder.blocks.%d." % i
        for a in ("attn", "cross_attn"):
            lin(p + a + ".query", d, d); lin(p + a + ".key", d, d, bias=False)
            lin(p + a + ".value", d, d); lin(p + a + ".out", d, d, scale=0.5)
            ln(p + a + "_ln", d)
        ln(p + "mlp_ln", d)
        lin(p + "mlp.0", 4 * d, d); lin(p + "mlp.2", d, 4 * d, scale=0.5)
    ln("decoder.ln", d)
    if style == "peaked":
        for k in w:
            if k == "decoder.token_embedding.weight":
                w[k] = w[k] * (0.05 / 0.3)
            elif k == "decoder.positional_embedding":
                w[k] = w[k] * 0.5
            elif k == "decoder.ln.weight":
                w[k] = w[k] * 6.0
            elif k.endswith("query.weight"):
                w[k] = w[k] * 4.0
            elif k.startswith("decoder.") and k.endswith("attn.out.weight"):
                w[k] = w[k] * 2.0
            elif k.endswith("mlp.2.weight"):
                w[k] = w[k] - w[k].mean(dim=1, keepdim=True)
    elif style != "plain":
        raise ValueError("unknown style %r" % (style,))
    return {k: v.to(dtype) for k, v in w.items()}


def sinusoids(length, channels, max_timescale=10000):
    """whisper_arch.py:93-104."""
    inc = np.log(max_timescale) / (channels // 2 - 1)
    inv = torch.exp(-inc * torch.arange(channels // 2, dtype=torch.float32))
    t = torch.arange(length, dtype=torch.float32)[:, None] * inv[None, :]
    return torch.cat([torch.sin(t), torch.cos(t)], dim=1)


def gelu(x, approximate="tanh"):
    # keras.activations.gelu default approximate=True (tanh form); 'none' = exact erf
    return F.gelu(x, approximate=approximate)


def layer_norm(x, w, prefix, eps=1e-5):
    return F.layer_norm(x, (x.shape[-1],), w[prefix + ".weight"], w[prefix + ".bias"], eps)


def linear(x, w, prefix):
    return F.linear(x, w[prefix + ".weight"], w.get(prefix + ".bias"))


def _split(x, n_head):
    b, t, d = x.shape
    return x.view(b, t, n_head, d // n_head).permute(0, 2, 1, 3)


def _merge(x):
    b, h, t, dh = x.shape
    return x.permute(0, 2, 1, 3).reshape(b, t, h * dh)


def attention(q, k, v, mask=None):
    """softmax(q k^T / sqrt(d_head)) v with where(mask, logits, -inf) (rmha.py:167-208)."""
    logits = torch.einsum("bhij,bhkj->bhik", q, k) / math.sqrt(q.shape[-1])
    if mask is not None:
        logits = torch.where(mask, logits, torch.tensor(float("-inf")))
    return torch.matmul(torch.softmax(logits, dim=-1), v)


class Whisper:
    """`round_acts=False`: the reference's arithmetic, float32 throughout.

    `round_acts=True`: the same network with every tensor that the MI355X engine keeps in bf16
    rounded to bf16 at the same place (GEMM inputs, stored Q/K/V, encoder attention weights before
    the P V product, attention and first-MLP outputs, encoder output, cross K/V); sums, residual
    stream, LayerNorm, softmax and logits stay float32.  It is not the reference: it separates,
    in a test, what bf16 storage costs (engine vs this) from what an error would look like
    (engine vs the float32 oracle has to stay inside the stated bf16 tolerance either way)."""

    def __init__(self, dims, weights, gelu_approximate="tanh", round_acts=False):
        self.dims = dims
        self.w = {k: v.float() for k, v in weights.items()}
        self.act = gelu_approximate
        self.pos = sinusoids(dims.n_audio_ctx, dims.n_audio_state)
        self.round_acts = round_acts

    def r(self, x):
        return x.to(torch.bfloat16).float() if self.round_acts else x

    def _attention(self, q, k, v, mask=None, round_p=False):
        if not (self.round_acts and round_p):
            return attention(q, k, v, mask)
        # flash-style: weights relative to the row maximum, rounded for the P V product, the
        # normaliser summed from the unrounded ones
        logits = torch.einsum("bhij,bhkj->bhik", q, k) / math.sqrt(q.shape[-1])
        p = torch.exp(logits - logits.max(dim=-1, keepdim=True).values)
        return torch.matmul(self.r(p), v) / p.sum(dim=-1, keepdim=True)

    # ------------------------------------------------------------- encoder
    def stem(self, mel):
        """mel [B, 3000, n_mels] -> [B, 1500, d] (whisper_arch.py:140-155)."""
        w = self.w
        x = self.r(mel.float()).permute(0, 2, 1)
        x = self.r(gelu(F.conv1d(x, w["encoder.conv1.weight"], w["encoder.conv1.bias"], padding=1), self.act))
        x = gelu(F.conv1d(x, w["encoder.conv2.weight"], w["encoder.conv2.bias"], stride=2, padding=1), self.act)
        x = x.permute(0, 2, 1)
        return x + self.pos[None, : x.shape[1]]

    def _self_block(self, x, p, n_head, mask=None, cache=None, round_p=False):
        """x + out(attn(LN(x))) (rmha.py:210-315); cache = (k, v) of earlier positions."""
        w, r = self.w, self.r
        h = r(layer_norm(x, w, p + "attn_ln"))
        q = _split(r(linear(h, w, p + "attn.query")), n_head)
        k = _split(r(linear(h, w, p + "attn.key")), n_head)
        v = _split(r(linear(h, w, p + "attn.value")), n_head)
        if cache is not None:
            k = torch.cat([cache[0], k], dim=2)
            v = torch.cat([cache[1], v], dim=2)
        o = linear(r(_merge(self._attention(q, k, v, mask, round_p))), w, p + "attn.out")
        return x + o, (k, v)

    def _mlp_block(self, x, p):
        """ffn(LN(x)) + x with normalize='middle' (transformer_arch.py:478-492)."""
        w, r = self.w, self.r
        h = r(layer_norm(x, w, p + "mlp_ln"))
        return x + linear(r(gelu(linear(h, w, p + "mlp.0"), self.act)), w, p + "mlp.2")

    def encoder(self, mel, return_layers=False):
        x = self.stem(mel)
        layers = [x]
        for i in range(self.dims.n_audio_layer):
            p = "encoder.blocks.%d." % i
            x, _ = self._self_block(x, p, self.dims.n_audio_head, round_p=True)
            x = self._mlp_block(x, p)
            layers.append(x)
        out = layer_norm(x, self.w, "encoder.ln_post")
        return (out, layers) if return_layers else out

    # ------------------------------------------------------------- decoder
    def cross_kv(self, enc):
        """Per layer (k, v) of the encoder output, computed once per window
        (rmha.py:241-245: no LayerNorm on the encoder output, key has no bias)."""
        kv, r = [], self.r
        enc = r(enc)
        for i in range(self.dims.n_text_layer):
            p = "decoder.blocks.%d.cross_attn." % i
            kv.append((_split(r(linear(enc, self.w, p + "key")), self.dims.n_text_head),
                       _split(r(linear(enc, self.w, p + "value")), self.dims.n_text_head)))
        return kv

    def decoder(self, tokens, cross, cache=None, offset=0, return_attention=False):
        """tokens [B, S] at positions offset.. -> logits [B, S, V], new self cache (and, with
        `return_attention`, the dict {'attn_layer_<i>': [B, H, S, offset + S], 'enc_attn_layer_<i>':
        [B, H, S, 1500]} of transformer_arch.py:697-703).
        token + learned position embedding, no scaling (text_transformer_arch.py:173-193,
        399-443); per layer self-attn -> cross-attn -> LN -> FFN; final LN; logits with the
        tied embedding (gpt2_arch.py:98-101)."""
        w, dims = self.w, self.dims
        b, s = tokens.shape
        x = w["decoder.token_embedding.weight"][tokens] + w["decoder.positional_embedding"][offset: offset + s][None]
        mask = None
        if s > 1:
            total = offset + s
            mask = (torch.arange(total)[None, :] <= (offset + torch.arange(s))[:, None])[None, None]
        new_cache, weights = [], {}

        def probabilities(q, k, m=None):
            logits = torch.einsum("bhij,bhkj->bhik", q, k) / math.sqrt(q.shape[-1])
            if m is not None:
                logits = torch.where(m, logits, torch.tensor(float("-inf")))
            return torch.softmax(logits, dim=-1)

        for i in range(dims.n_text_layer):
            p = "decoder.blocks.%d." % i
            hq = self.r(layer_norm(x, w, p + "attn_ln")) if return_attention else None
            x, kv = self._self_block(x, p, dims.n_text_head, mask, None if cache is None else cache[i])
            new_cache.append(kv)
            if return_attention:
                qs = _split(self.r(linear(hq, w, p + "attn.query")), dims.n_text_head)
                weights["attn_layer_%d" % i] = probabilities(qs, kv[0], mask)
            h = self.r(layer_norm(x, w, p + "cross_attn_ln"))
            q = linear(h, w, p + "cross_attn.query")
            if b * s > 256:                 # the engine's large-row path stores the query in bf16
                q = self.r(q)
            q = _split(q, dims.n_text_head)
            if return_attention:
                weights["enc_attn_layer_%d" % i] = probabilities(q, cross[i][0])
            o = linear(self.r(_merge(attention(q, cross[i][0], cross[i][1]))), w, p + "cross_attn.out")
            x = x + o
            x = self._mlp_block(x, p)
        x = self.r(layer_norm(x, w, "decoder.ln"))
        logits = x @ w["decoder.token_embedding.weight"].T
        return (logits, new_cache, weights) if return_attention else (logits, new_cache)


# ---------------------------------------------------------------------------
# generation (generation_utils.py)
# ---------------------------------------------------------------------------
@dataclass
class FilterConfig:
    """The opt-in Whisper logits filter (whisper.py:354-416)."""
    remove_tokens: tuple = ()               # suppressed at every step after the first
    remove_tokens_first: tuple = ()         # suppressed at the first step (adds ' ' and EOS)
    timestamp_begin: int = 50364
    eos: int = 50257
    max_initial_timestamp: int = 1


def timestamp_filter(scores, generated, t, cfg):
    """scores [B, V] raw logits, generated [B, >=t] tokens produced so far."""
    scores = scores.clone()
    v = scores.shape[-1]
    ninf = float("-inf")
    if t == 0:
        scores[:, list(cfg.remove_tokens_first)] = ninf
        scores[:, : cfg.timestamp_begin] = ninf
        if cfg.max_initial_timestamp > 0:
            scores[:, cfg.timestamp_begin + cfg.max_initial_timestamp:] = ninf
        return scores
    scores[:, list(cfg.remove_tokens)] = ninf
    last = generated[:, t - 1] >= cfg.timestamp_begin
    penult = torch.ones_like(last) if t < 2 else generated[:, t - 2] >= cfg.timestamp_begin
    for b in range(scores.shape[0]):
        if last[b] and not penult[b]:
            scores[b, : cfg.eos] = ninf
        if last[b] and penult[b]:
            scores[b, cfg.timestamp_begin:] = ninf
    logp = torch.log_softmax(scores, dim=-1)
    ts = torch.logsumexp(logp[:, cfg.timestamp_begin:], dim=-1)
    tx = logp[:, : cfg.timestamp_begin].max(dim=-1).values
    for b in range(scores.shape[0]):
        if ts[b] > tx[b]:
            scores[b, : cfg.timestamp_begin] = ninf
    return scores


def process_logits(scores, lengths, generated, t, temperature=None, length_temperature=None,
                   logits_filter=None):
    """generation_utils.py:484-524."""
    if temperature is not None:
        scores = scores / temperature
    if length_temperature is not None:
        scores = scores * ((lengths + 1).to(scores.dtype) ** length_temperature)[:, None]
    if logits_filter is not None:
        if isinstance(logits_filter, FilterConfig):
            scores = timestamp_filter(scores, generated, t, logits_filter)
        else:
            scores = scores.clone()
            scores[:, list(logits_filter)] = float("-inf")
    return torch.log_softmax(scores, dim=-1)


def gumbel_noise(seed, stream, n):
    """The random numbers of the categorical draw of generation_utils.py:546, as the HIP path
    generates them (the reference's `keras.random.categorical` is unseeded and backend specific,
    so only the distribution can be shared with it): element (stream << 32 | j), j < n, of the
    SplitMix64 sequence started at `seed`, top 23 bits -> u in (0, 1), g = -log(-log u) in
    float32.  argmax(x + g) is a draw from softmax(x)."""
    m = (1 << 64) - 1
    with np.errstate(over="ignore"):
        ctr = (np.uint64(stream) << np.uint64(32)) | np.arange(n, dtype=np.uint64)
        z = (np.uint64(seed & m) + ctr * np.uint64(0x9E3779B97F4A7C15))
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    u = ((z >> np.uint64(41)).astype(np.float32) + np.float32(0.5)) * np.float32(1.0 / 8388608.0)
    return torch.from_numpy(-np.log(-np.log(u, dtype=np.float32), dtype=np.float32))


def greedy(model, mel=None, prompt=None, *, enc=None, max_new_tokens=None, max_length=None,
           eos=50257, pad=50257, early_stopping=True, logits_filter=None, temperature=None,
           length_temperature=None, return_logits=False, sample_seed=None):
    """generation_utils.py:171-300.  prompt [B, S] int64.  Returns dict(tokens [B, max_steps],
    lengths [B] (EOS not counted), scores [B]).  `sample_seed` given: the token is drawn from
    softmax(lp) (what select_next_token does when a temperature is set, :546-548) with the
    noise of `gumbel_noise`, stream t * B + b."""
    if enc is None:
        enc = model.encoder(mel)
    cross = model.cross_kv(enc)
    b, s = prompt.shape
    max_steps = (max_length - s) if max_length is not None else max_new_tokens
    generated = torch.full((b, max_steps), pad, dtype=torch.long)
    lengths = torch.full((b,), s, dtype=torch.long)
    scores = torch.zeros(b)
    finished = torch.zeros(b, dtype=torch.bool)
    all_logits = []
    inputs, cache, offset = prompt, None, 0
    for t in range(max_steps):
        logits, cache = model.decoder(inputs, cross, cache, offset)
        offset += inputs.shape[1]
        z = logits[:, -1, :]
        if return_logits:
            all_logits.append(z)
        lp = process_logits(z, lengths, generated, t, temperature, length_temperature, logits_filter)
        if sample_seed is None:
            tok = lp.argmax(dim=-1)
        else:
            noise = torch.stack([gumbel_noise(sample_seed, t * b + i, lp.shape[1]) for i in range(b)])
            tok = (lp + noise).argmax(dim=-1)
        sc = lp.gather(1, tok[:, None])[:, 0]
        tok = torch.where(finished, torch.tensor(pad), tok)
        scores = scores + torch.where(finished, torch.tensor(0.0), sc)
        finished = finished | (tok == eos)
        lengths = lengths + (~finished).long()
        generated[:, t] = tok
        inputs = tok[:, None]
        if early_stopping and bool(finished.all()):
            break
    out = dict(tokens=generated, lengths=lengths - s, scores=scores)
    if return_logits:
        out["logits"] = torch.stack(all_logits, dim=1)
    return out


def beam_search(model, mel=None, prompt=None, *, enc=None, num_beams=5, num_sentences=1,
                max_new_tokens=None, max_length=None, eos=50257, pad=50257, early_stopping=True,
                logits_filter=None, length_power=None, temperature=None, length_temperature=None,
                sample_seed=None):
    """generation_utils.py:302-481, with the beam index b * num_beams + idx // V (the
    reference omits `* num_beams`, which only matters for batch > 1).  `sample_seed` given: the
    num_beams candidates are categorical draws with replacement over the flattened scores
    (:382-384, 546), draw d of batch element i with the noise stream t * B * beams + i * beams + d."""
    if enc is None:
        enc = model.encoder(mel)
    b, s = prompt.shape
    nb = num_beams
    v = model.dims.n_vocab
    cross = [(k.repeat_interleave(nb, 0), vv.repeat_interleave(nb, 0)) for k, vv in model.cross_kv(enc)]
    max_steps = (max_length - s) if max_length is not None else max_new_tokens
    eb = b * nb
    generated = torch.full((eb, max_steps), pad, dtype=torch.long)
    lengths = torch.full((eb,), s, dtype=torch.long)
    scores = torch.zeros(eb)
    finished = torch.zeros(eb, dtype=torch.bool)
    eos_mask = torch.full((1, v), float("-inf"))
    eos_mask[0, pad] = 0.0
    inputs, cache, offset = prompt.repeat_interleave(nb, 0), None, 0
    batch_add = torch.arange(b).repeat_interleave(nb) * nb
    for t in range(max_steps):
        logits, cache = model.decoder(inputs, cross, cache, offset)
        offset += inputs.shape[1]
        lp = process_logits(logits[:, -1, :], lengths, generated, t, temperature, length_temperature,
                            logits_filter)
        lws = scores[:, None] + torch.where(finished[:, None], eos_mask, lp)
        r = lws
        if length_power is not None:
            r = r / ((lengths + 1).to(r.dtype)[:, None] ** length_power)
        r = r.reshape(b, nb * v)
        if t == 0:
            r = r[:, :v]
        if sample_seed is None:
            idx = torch.topk(r, nb, dim=1).indices.reshape(eb)
        else:
            idx = torch.stack([(r[i // nb] + gumbel_noise(sample_seed, t * eb + i, r.shape[1])).argmax()
                               for i in range(eb)])
        beam = idx // v + batch_add
        tok = idx % v
        lengths, finished = lengths[beam], finished[beam]
        scores = lws[beam].gather(1, tok[:, None])[:, 0]
        tok = torch.where(finished, torch.tensor(pad), tok)
        finished = finished | (tok == eos)
        lengths = lengths + (~finished).long()
        generated = generated[beam]
        generated[:, t] = tok
        cache = [(k[beam], vv[beam]) for k, vv in cache]
        inputs = tok[:, None]
        if early_stopping and bool(finished.view(b, nb)[:, :num_sentences].all()):
            break
    return dict(tokens=generated.view(b, nb, -1)[:, :num_sentences],
                lengths=(lengths - s).view(b, nb)[:, :num_sentences],
                scores=scores.view(b, nb)[:, :num_sentences])


def process_model_output(tokens, lengths):
    """utils/text/tokens_processing.py:20-39: tokens[b][(beam)][:length] as python lists."""
    tokens, lengths = np.asarray(tokens), np.asarray(lengths)
    if lengths.ndim == 0:
        return tokens[: int(lengths)].tolist()
    return [process_model_output(t, l) for t, l in zip(tokens, lengths)]
