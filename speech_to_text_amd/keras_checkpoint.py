"""A model directory written by the reference (pretrained_models/<name>/saving/...), read in-process.

`keras.Model.save_weights` puts every layer's variables under `<layer path>/vars/<i>` of a
`.weights.h5` file (custom_train_objects/checkpoint_manager.py:148-167) and
`BaseModel.save_models_config` writes the serialised model next to it as `config_models.json`
(models/interfaces/base_model.py:742-752).  This module reads both without keras or h5py (the
HDF5 subset is hdf5_lite.py) and returns OpenAI-named tensors in OpenAI layouts, which is what the
engine takes.

Group names depend on the Keras version (attribute names, container names `dense`, `dense_1`, or
the layers' own names `query_layer`, `layer_3`), so the mapping recognises every spelling of a
role and settles what names cannot tell by shape:
  kernel [d, 4d] / [4d, d] -> mlp.0 / mlp.2,  [3, n_mels, d] / [3, d, d] -> conv1 / conv2,
  [V, d] / [n_ctx, d] 2-D tables of the decoder -> token / positional embedding.
Layouts are turned back into the OpenAI ones, undoing models/weights_converter.py:252-271
(Dense kernel [in, out] -> [out, in], Conv1D kernel [k, in, out] -> [out, in, k]).

PARITY UNPINNED: the reference ships no checkpoint and keras is not installed here, so the layout
was taken from the reference's layer code (architectures/transformers/transformer_arch.py:346-376,
559-565, architectures/layers/residual_multi_head_attention.py:108-133) and from the Keras-3
saving format, and is exercised on synthetic files written with h5py.
"""
import glob
import json
import os
import re

import numpy as np

from .hdf5_lite import Hdf5File


def latest_checkpoint(directory):
    """The `*.weights.h5` a model directory (or its `saving/` folder) is restored from, or None.
    With a `checkpoint.json` it is the one the reference's CheckpointManager would load
    (custom_train_objects/checkpoint_manager.py:88-98, 169-190): `loaded` indexes the list
    `checkpoints`, whose entry's `counter` names the file `ckpt-{counter:04d}.weights.h5`; `loaded =
    'best'` means `best.weights.h5` when it exists, else the newest entry.  Without that file (or when
    it names a file that is gone) the highest counter wins."""
    for folder in (os.path.join(directory, "saving"), directory):
        found = sorted(glob.glob(os.path.join(folder, "*.weights.h5")))
        if not found:
            continue
        state = os.path.join(folder, "checkpoint.json")
        if os.path.exists(state):
            try:
                with open(state) as f:
                    info = json.load(f)
                entries, loaded = info.get("checkpoints") or [], info.get("loaded", -1)
                best = os.path.join(folder, "best.weights.h5")
                if loaded == "best" and os.path.exists(best):
                    return best
                if entries:
                    entry = entries[loaded] if isinstance(loaded, int) and -len(entries) <= loaded < len(entries) else entries[-1]
                    cand = os.path.join(folder, "ckpt-%04d.weights.h5" % int(entry["counter"]))
                    if os.path.exists(cand):
                        return cand
            except (OSError, ValueError, KeyError, TypeError):
                pass
        numbered = [p for p in found if re.search(r"ckpt-\d+\.weights\.h5$", p)]
        return (numbered or found)[-1]
    return None


def read_vars(path):
    """{group path of a layer: [its variables in order]} of a Keras-3 .weights.h5 file."""
    found = {}
    with Hdf5File(path) as f:
        for name in f.names():
            parts = name.split("/")
            if len(parts) >= 2 and parts[-2] == "vars":
                found.setdefault("/".join(parts[:-2]), []).append((parts[-1], name))
        out = {}
        for layer, entries in found.items():
            entries.sort(key=lambda e: int(e[0]) if e[0].isdigit() else 1 << 30)
            out[layer] = [f.read(name) for _, name in entries]
    return out


_INDEX = re.compile(r"(?:transformer_layer|layer)(?:_(\d+))?$")


def _role(tokens):
    """(side, layer index or None, module, part) from the path components of a layer."""
    side = "decoder" if any("decoder" in t for t in tokens) else "encoder" if any("encoder" in t for t in tokens) else None
    index = None
    for t in tokens:
        m = _INDEX.match(t)
        if m:
            index = int(m.group(1) or 0)
    has = lambda *names: any(t in names for t in tokens)
    if has("enc_attention", "enc_mha"):
        module = "cross_attn"
    elif has("attention", "mha"):
        module = "attn"
    elif has("ffn"):
        module = "mlp"
    else:
        module = None
    part = None
    for names, p in ((("wq", "query_layer"), "query"), (("wk", "key_layer"), "key"), (("wv", "value_layer"), "value"),
                     (("output_layer",), "out"), (("inp_norm_layer", "norm_input"), "ln")):
        if tokens[-1] in names:
            part = p
    return side, index, module, part


def to_openai(layers):
    """Layer variables keyed by group path -> {OpenAI name: array in OpenAI layout}."""
    out, conv, tables = {}, [], []
    for path, arrays in sorted(layers.items()):
        tokens = path.lower().split("/")
        if tokens[0] == "optimizer":
            continue
        side, index, module, part = _role(tokens)
        first = arrays[0]
        if first.ndim == 3:                                    # Conv1D kernel [k, in, out] (+ bias)
            conv.append((path, arrays))
            continue
        if side is None:
            raise ValueError("cannot tell encoder from decoder for %r" % path)
        if index is None:
            if first.ndim == 2 and len(arrays) == 1:           # embedding tables
                tables.append(first)
            elif first.ndim == 1 and len(arrays) == 2:         # the block's final LayerNorm
                name = "encoder.ln_post" if side == "encoder" else "decoder.ln"
                out[name + ".weight"], out[name + ".bias"] = arrays
            else:
                raise ValueError("unexpected variables at %r: %s" % (path, [a.shape for a in arrays]))
            continue
        prefix = "%s.blocks.%d." % (side, index)
        if module in ("attn", "cross_attn"):
            if part == "ln":
                out[prefix + module + "_ln.weight"], out[prefix + module + "_ln.bias"] = arrays
            elif part in ("query", "key", "value", "out"):
                out[prefix + "%s.%s.weight" % (module, part)] = np.ascontiguousarray(first.T)
                if len(arrays) > 1:
                    out[prefix + "%s.%s.bias" % (module, part)] = arrays[1]
            else:
                raise ValueError("unknown attention part at %r" % path)
        elif module == "mlp":
            which = "0" if first.shape[1] > first.shape[0] else "2"          # [d, 4d] up, [4d, d] down
            out[prefix + "mlp.%s.weight" % which] = np.ascontiguousarray(first.T)
            if len(arrays) > 1:
                out[prefix + "mlp.%s.bias" % which] = arrays[1]
        elif first.ndim == 1 and len(arrays) == 2:             # the layer's own `norm` (before the FFN)
            out[prefix + "mlp_ln.weight"], out[prefix + "mlp_ln.bias"] = arrays
        else:
            raise ValueError("unplaced variables at %r: %s" % (path, [a.shape for a in arrays]))
    if len(conv) != 2:
        raise ValueError("expected the two Conv1D layers of the stem, found %d" % len(conv))
    conv.sort(key=lambda pa: (pa[1][0].shape[1] == pa[1][0].shape[2], pa[0]))        # [3, n_mels, d] first
    for i, (_, arrays) in enumerate(conv, start=1):
        out["encoder.conv%d.weight" % i] = np.ascontiguousarray(np.transpose(arrays[0], (2, 1, 0)))
        out["encoder.conv%d.bias" % i] = arrays[1]
    if len(tables) != 2:
        raise ValueError("expected token and positional embedding, found %d tables" % len(tables))
    tables.sort(key=lambda a: -a.shape[0])
    out["decoder.token_embedding.weight"], out["decoder.positional_embedding"] = tables
    return {k: np.asarray(v, dtype=np.float32) for k, v in out.items()}



def _find(config, *names):
    """First value stored under one of `names` anywhere in a (nested) serialised Keras config."""
    if isinstance(config, dict):
        for n in names:
            if n in config and not isinstance(config[n], (dict, list)):
                return config[n]
        for v in config.values():
            got = _find(v, *names)
            if got is not None:
                return got
    elif isinstance(config, list):
        for v in config:
            got = _find(v, *names)
            if got is not None:
                return got
    return None


def _gelu_form(value):
    """Which GELU a serialised activation means: True = tanh approximation (what the name 'gelu'
    resolves to in Keras 3, architectures/layers/custom_activations.py:67-75), False = exact erf."""
    if value is None:
        return None
    if isinstance(value, dict):
        name = str(value.get("class_name") or value.get("name") or _find(value, "name") or "").lower()
        approximate = _find(value, "approximate")
        if "gelu" in name:
            return True if approximate is None else bool(approximate)
        value = name
    name = str(value).lower()
    if name in ("gelu", "gelu_tanh", "gelu_new", "gelu_pytorch_tanh", "quick_gelu_tanh"):
        return True
    if name in ("gelu_exact", "gelu_erf", "gelu_python"):
        return False
    raise ValueError("activation %r: the engine implements GELU (tanh or erf form) only" % (value,))


def model_config(directory):
    """What `saving/config_models.json` says about the model, as far as the engine needs it:
    {'dims': {...the sizes it states...}, 'gelu_tanh': bool or None}.  Keys are looked up by the
    names the reference's hyper-parameter sets use (whisper_arch.py:39-91), with or without their
    `encoder_` / `decoder_` prefixes; absent values are simply not reported (the weights settle them)."""
    path = None
    for folder in (os.path.join(directory, "saving"), directory):
        cand = os.path.join(folder, "config_models.json")
        if os.path.exists(cand):
            path = cand
            break
    if path is None:
        return {"dims": {}, "gelu_tanh": None}
    with open(path) as f:
        config = json.load(f)
    model = config.get("model", config)
    dims = {}
    for key, names in (("n_mels", ("encoder_n_mel_channels", "n_mel_channels")),
                       ("n_audio_state", ("encoder_embedding_dim",)), ("n_audio_layer", ("encoder_num_layers",)),
                       ("n_audio_head", ("encoder_mha_num_heads",)), ("n_audio_ctx", ("encoder_max_input_length",)),
                       ("n_vocab", ("decoder_vocab_size", "vocab_size")), ("n_text_state", ("decoder_embedding_dim",)),
                       ("n_text_layer", ("decoder_num_layers",)), ("n_text_head", ("decoder_mha_num_heads",)),
                       ("n_text_ctx", ("decoder_max_input_length",))):
        v = _find(model, *names)
        if isinstance(v, (int, float)) and not isinstance(v, bool):
            dims[key] = int(v)
    forms = {_gelu_form(_find(model, n)) for n in ("encoder_ffn_activation", "decoder_ffn_activation", "ffn_activation",
                                                    "encoder_activation")} - {None}
    if len(forms) > 1:
        raise ValueError("%s mixes the tanh and the erf form of GELU: the engine runs one form everywhere" % path)
    return {"dims": dims, "gelu_tanh": forms.pop() if forms else None}


def load_model_dir(directory):
    """(OpenAI-named weights, config) of a directory the reference wrote, or None when it holds no
    Keras checkpoint.  Raises when `config_models.json` states a size the weights contradict."""
    ckpt = latest_checkpoint(directory)
    if ckpt is None:
        return None
    weights = to_openai(read_vars(ckpt))
    config = model_config(directory)
    d = weights["encoder.conv1.weight"].shape[0]
    have = {"n_mels": weights["encoder.conv1.weight"].shape[1], "n_audio_state": d, "n_text_state": weights["decoder.token_embedding.weight"].shape[1],
            "n_vocab": weights["decoder.token_embedding.weight"].shape[0], "n_text_ctx": weights["decoder.positional_embedding"].shape[0],
            "n_audio_layer": 1 + max(int(k.split(".")[2]) for k in weights if k.startswith("encoder.blocks.")),
            "n_text_layer": 1 + max(int(k.split(".")[2]) for k in weights if k.startswith("decoder.blocks."))}
    for key, value in config["dims"].items():
        if key in have and have[key] != value:
            raise ValueError("%s: config_models.json says %s = %d, the checkpoint %s holds %d"
                             % (directory, key, value, os.path.basename(ckpt), have[key]))
    for key in ("n_audio_head", "n_text_head"):
        if key in config["dims"] and config["dims"][key] * 64 != d:
            raise ValueError("%s: %s = %d with width %d: the engine needs a head size of 64"
                             % (directory, key, config["dims"][key], d))
    return weights, config
