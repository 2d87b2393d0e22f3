#!/usr/bin/env python3
"""Keras-3 `.weights.h5` checkpoint of the reference's Whisper -> `weights.npz` with OpenAI names.

The reference saves its models with `keras.Model.save_weights` into
`pretrained_models/<name>/saving/ckpt-NNNN.weights.h5` next to `config_models.json`
(custom_train_objects/checkpoint_manager.py:137-167).  `MI355XRuntime.load_engine(<model dir>)`
reads such a directory directly (speech_to_text_amd/keras_checkpoint.py, own HDF5 reader); this
tool is the same conversion as a file-to-file step through h5py (run it where h5py exists, e.g.
`/opt/conda/bin/python3.9 tools/convert_keras_weights.py <model dir>`): it writes
`saving/weights.npz`, and its `--make-synthetic` writes the test checkpoints.  Nothing here needs
keras.

A Keras-3 weights file is a tree of groups that follows the attribute names of the model
(`encoder/transformer_layers/transformer_layer_3/attention/wq/vars/0` ...), every layer's variables
under `vars/0, 1, ...` in creation order (kernel, bias / gamma, beta / embeddings).  Group names
depend on the Keras version (attribute names, container names `dense`, `dense_1`, or the
layers' own names `query_layer`, `dense_1`, `layer_3`), so the mapping below recognises every
spelling of a role and settles what names cannot tell by shape:
  kernel [d, 4d] / [4d, d] -> mlp.0 / mlp.2,  [3, n_mels, d] / [3, d, d] -> conv1 / conv2,
  [V, d] / [n_ctx, d] 2-D tables of the decoder -> token / positional embedding.
Layouts are turned back into the OpenAI ones, undoing models/weights_converter.py:252-271
(Dense kernel [in, out] -> [out, in], Conv1D kernel [k, in, out] -> [out, in, k]).

PARITY UNPINNED: the reference ships no checkpoint and keras is not installed here, so the
layout was taken from the reference's layer code (architectures/transformers/transformer_arch.py:
346-376, 559-565, architectures/layers/residual_multi_head_attention.py:108-133) and from the
Keras-3 saving format, and is exercised on a synthetic file written by `--make-synthetic`.
"""
import argparse
import glob
import json
import os
import re
import sys

import numpy as np


def read_vars(path):
    """{group path of a layer: [its variables in order]} of a Keras-3 .weights.h5 file."""
    import h5py
    found = {}

    def visit(name, obj):
        if isinstance(obj, h5py.Group) and name.split("/")[-1] == "vars" and len(obj.keys()):
            keys = sorted(obj.keys(), key=lambda k: int(k) if k.isdigit() else 1 << 30)
            found[name[: -len("/vars")]] = [np.asarray(obj[k]) for k in keys]

    with h5py.File(path, "r") as f:
        f.visititems(visit)
    return found


sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from speech_to_text_amd.keras_checkpoint import to_openai  # noqa: E402  (numpy only: the mapping the runtime uses)


def write_synthetic(path, n_mels=80, d=128, layers=1, vocab=512, ctx=32, seed=0):
    """A file in the attribute-name layout with random values; returns the OpenAI-named truth."""
    import h5py
    rng = np.random.default_rng(seed)
    truth = {}

    def put(f, group, arrays):
        g = f.require_group(group + "/vars")
        for i, a in enumerate(arrays):
            g.create_dataset(str(i), data=a.astype(np.float32))

    def dense(f, group, name, n_out, n_in, bias=True):
        w = (rng.standard_normal((n_out, n_in)) / np.sqrt(n_in)).astype(np.float32)
        truth[name + ".weight"] = w
        arrays = [w.T]
        if bias:
            truth[name + ".bias"] = (0.1 * rng.standard_normal(n_out)).astype(np.float32)
            arrays.append(truth[name + ".bias"])
        put(f, group, arrays)

    def norm(f, group, name):
        truth[name + ".weight"] = (1 + 0.1 * rng.standard_normal(d)).astype(np.float32)
        truth[name + ".bias"] = (0.1 * rng.standard_normal(d)).astype(np.float32)
        put(f, group, [truth[name + ".weight"], truth[name + ".bias"]])

    with h5py.File(path, "w") as f:
        for i, (n_in, group) in enumerate(((n_mels, "conv1d"), (d, "conv1d_1")), start=1):
            w = (rng.standard_normal((d, n_in, 3)) / np.sqrt(3 * n_in)).astype(np.float32)
            truth["encoder.conv%d.weight" % i] = w
            truth["encoder.conv%d.bias" % i] = (0.1 * rng.standard_normal(d)).astype(np.float32)
            put(f, "encoder/feature_extractor/layers/" + group, [np.transpose(w, (2, 1, 0)), truth["encoder.conv%d.bias" % i]])
        for side in ("encoder", "decoder"):
            for l in range(layers):
                g = "%s/transformer_layers/transformer_layer%s" % (side, "" if l == 0 else "_%d" % l)
                p = "%s.blocks.%d." % (side, l)
                for attr, module in (("attention", "attn"),) + ((("enc_attention", "cross_attn"),) if side == "decoder" else ()):
                    dense(f, g + "/%s/wq" % attr, p + module + ".query", d, d)
                    dense(f, g + "/%s/wk" % attr, p + module + ".key", d, d, bias=False)
                    dense(f, g + "/%s/wv" % attr, p + module + ".value", d, d)
                    dense(f, g + "/%s/output_layer" % attr, p + module + ".out", d, d)
                    norm(f, g + "/%s/inp_norm_layer" % attr, p + module + "_ln")
                norm(f, g + "/norm", p + "mlp_ln")
                dense(f, g + "/ffn/layers/dense", p + "mlp.0", 4 * d, d)
                dense(f, g + "/ffn/layers/dense_1", p + "mlp.2", d, 4 * d)
            norm(f, side + "/norm", "encoder.ln_post" if side == "encoder" else "decoder.ln")
        truth["decoder.token_embedding.weight"] = (0.3 * rng.standard_normal((vocab, d))).astype(np.float32)
        truth["decoder.positional_embedding"] = (0.3 * rng.standard_normal((ctx, d))).astype(np.float32)
        put(f, "decoder/embeddings/token_embedding_layer", [truth["decoder.token_embedding.weight"]])
        put(f, "decoder/embeddings/pos_embedding_layer", [truth["decoder.positional_embedding"]])
    return truth


def convert(source, out=None):
    if os.path.isdir(source):
        found = sorted(glob.glob(os.path.join(source, "saving", "*.weights.h5")) + glob.glob(os.path.join(source, "*.weights.h5")))
        if not found:
            raise FileNotFoundError("no *.weights.h5 under %s" % source)
        source = found[-1]                                     # the latest checkpoint
    weights = to_openai(read_vars(source))
    out = out or os.path.join(os.path.dirname(source), "weights.npz")
    np.savez(out, **weights)
    return out, weights


def main():
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("source", help="model directory or .weights.h5 file")
    ap.add_argument("--out", default=None)
    ap.add_argument("--make-synthetic", action="store_true", help="write a small random checkpoint to `source` (a file) first")
    a = ap.parse_args()
    if a.make_synthetic:
        truth = write_synthetic(a.source)
        np.savez(a.source.replace(".weights.h5", ".truth.npz"), **truth)
    out, weights = convert(a.source, a.out)
    print("wrote %s: %d tensors, %.1f M parameters" % (out, len(weights), sum(v.size for v in weights.values()) / 1e6))


if __name__ == "__main__":
    main()
