"""Weight sources for the engine.

* ``random_whisper_weights``: seeded synthetic weights with the OpenAI Whisper state-dict names
  and shapes (no checkpoint ships with the reference and none can be downloaded offline).
* ``from_openai_state_dict`` / ``from_hf_state_dict``: name normalisation for real checkpoints,
  the same two source formats the reference's converter accepts
  (models/weights_converter.py:49-73, architectures/transformers/whisper_arch.py:157-169, 201-214).
* ``from_keras_layout``: the layout the reference's converter produces (Dense kernels [in, out],
  Conv1D kernels [k, in, out]) back to the engine's [out, in] / [out, in, k].
"""
import math
import re

import numpy as np
import torch


def random_whisper_weights(dims, seed=0, style="plain"):
    """style "plain": unit-scale draws; a greedy decode over them repeats the last prompt token
    whatever the audio (the tied embedding dominates its own logit).  style "peaked": the same
    draws rescaled (small embedding and positions, final LayerNorm gain 6, query matrices x 4,
    decoder attention outputs x 2, rows of the second MLP matrices centred) so that the next token
    depends on the audio and on the history - what a benchmark's self-check or a generation test
    needs."""
    if style not in ("plain", "peaked"):
        raise ValueError("unknown style %r" % (style,))
    g = torch.Generator().manual_seed(seed)
    w = {}

    def lin(name, out, inp, bias=True, scale=1.0):
        w[name + ".weight"] = torch.randn(out, inp, generator=g) * (scale / math.sqrt(inp))
        if bias:
            w[name + ".bias"] = torch.randn(out, generator=g) * 0.02

    def ln(name, d):
        w[name + ".weight"] = 1.0 + 0.1 * torch.randn(d, generator=g)
        w[name + ".bias"] = 0.05 * torch.randn(d, generator=g)

    d, m = dims["n_audio_state"], dims["n_mels"]
    w["encoder.conv1.weight"] = torch.randn(d, m, 3, generator=g) / math.sqrt(3 * m)
    w["encoder.conv1.bias"] = torch.randn(d, generator=g) * 0.02
    w["encoder.conv2.weight"] = torch.randn(d, d, 3, generator=g) / math.sqrt(3 * d)
    w["encoder.conv2.bias"] = torch.randn(d, generator=g) * 0.02
    for i in range(dims["n_audio_layer"]):
        p = "encoder.blocks.%d." % i
        lin(p + "attn.query", d, d); lin(p + "attn.key", d, d, bias=False)
        lin(p + "attn.value", d, d); lin(p + "attn.out", d, d, scale=0.5)
        ln(p + "attn_ln", d); ln(p + "mlp_ln", d)
        lin(p + "mlp.0", 4 * d, d); lin(p + "mlp.2", d, 4 * d, scale=0.5)
    ln("encoder.ln_post", d)
    d = dims["n_text_state"]
    w["decoder.token_embedding.weight"] = torch.randn(dims["n_vocab"], d, generator=g) * 0.3
    w["decoder.positional_embedding"] = torch.randn(dims["n_text_ctx"], d, generator=g) * 0.1
    for i in range(dims["n_text_layer"]):
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
    return w


def dims_from_openai_state_dict(sd):
    d = sd["encoder.conv1.weight"].shape[0]
    n_enc = 1 + max(int(k.split(".")[2]) for k in sd if k.startswith("encoder.blocks."))
    n_dec = 1 + max(int(k.split(".")[2]) for k in sd if k.startswith("decoder.blocks."))
    return dict(n_mels=sd["encoder.conv1.weight"].shape[1], n_audio_ctx=1500, n_audio_state=d,
                n_audio_head=d // 64, n_audio_layer=n_enc,
                n_vocab=sd["decoder.token_embedding.weight"].shape[0],
                n_text_ctx=sd["decoder.positional_embedding"].shape[0],
                n_text_state=sd["decoder.token_embedding.weight"].shape[1],
                n_text_head=sd["decoder.token_embedding.weight"].shape[1] // 64, n_text_layer=n_dec)


def from_openai_state_dict(sd):
    """OpenAI `model_state_dict` (what `load_whisper` returns, whisper_arch.py:292-316) as is."""
    out = {}
    for k, v in sd.items():
        if k == "encoder.positional_embedding" or k.endswith("attn.key.bias"):
            continue
        out[k] = v
    return out


_HF_RULES = [
    (r"^model\.", ""),
    (r"\.layers\.(\d+)\.", r".blocks.\1."),
    (r"\.self_attn_layer_norm\.", ".attn_ln."), (r"\.encoder_attn_layer_norm\.", ".cross_attn_ln."),
    (r"\.final_layer_norm\.", ".mlp_ln."),
    (r"\.self_attn\.", ".attn."), (r"\.encoder_attn\.", ".cross_attn."),
    (r"\.q_proj\.", ".query."), (r"\.k_proj\.", ".key."), (r"\.v_proj\.", ".value."),
    (r"\.out_proj\.", ".out."), (r"\.fc1\.", ".mlp.0."), (r"\.fc2\.", ".mlp.2."),
    (r"^encoder\.layer_norm\.", "encoder.ln_post."), (r"^decoder\.layer_norm\.", "decoder.ln."),
    (r"^decoder\.embed_tokens\.weight$", "decoder.token_embedding.weight"),
    (r"^decoder\.embed_positions\.weight$", "decoder.positional_embedding"),
]


def from_hf_state_dict(sd):
    """transformers `WhisperForConditionalGeneration.state_dict()` names -> OpenAI names."""
    out = {}
    for k, v in sd.items():
        if k.startswith("proj_out") or "encoder.embed_positions" in k or k.endswith("k_proj.bias"):
            continue
        name = k
        for pat, rep in _HF_RULES:
            name = re.sub(pat, rep, name)
        out[name] = v
    return out


def from_keras_layout(weights):
    """Undo `transpose_weights` of the reference (models/weights_converter.py:252-271): Dense
    kernels arrive as [in, out], Conv1D kernels as [k, in, out]; embeddings and vectors as is."""
    out = {}
    for k, v in weights.items():
        a = np.asarray(v)
        if k.endswith(".weight") and a.ndim == 2 and "embedding" not in k:
            a = a.T
        elif a.ndim == 3:
            a = np.transpose(a, (2, 1, 0))
        out[k] = np.ascontiguousarray(a)
    return out
