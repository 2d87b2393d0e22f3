#!/usr/bin/env python3
"""Writes the oracle-side fixtures of the model-size parity tests (tests/test_model_sizes_gpu.py).

Run in the build container (CPU only, a few minutes, ~20 GB of host memory for large-v3):

    python tests/golden/make_model_fixtures.py [base_b32] [large_v3_full]

Everything comes from oracle/model.py (the torch float32 restatement of the reference's
encoder / decoder / greedy loop) run on seeded synthetic weights rounded to bf16 - the weights
the HIP engine holds.  Seeds are stored, so the GPU test regenerates the same weights and mel
windows with the same torch CPU generator and only the expensive oracle outputs travel:

    enc_rows, enc_samples   encoder output rows [windows, len(enc_rows), d] of the oracle
    enc_mean, enc_std       per-window statistics of the full oracle encoder output
    tokens, scores          the oracle's greedy ids [windows, steps] and summed log-probabilities
    margins                 log-prob gap between the oracle's best and second-best token per step
                            (a gap below the test's near-tie allowance marks a step where a bf16
                            engine may legitimately pick the runner-up)
    logit_std               standard deviation of the oracle's logits row per step (the near-tie
                            allowance is stated relative to it, tests/_parity.py)

The reference holds no fixture for this part of the path (parity unpinned, see the oracle's
header); these files pin the HIP path to the oracle, not the oracle to the reference.
"""
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
from oracle import model as om  # noqa: E402

CASES = {
    # BASELINE configs[2]: whisper-base, 32 windows, greedy
    "base_b32": dict(model="base", windows=32, steps=32, weight_seed=11, mel_seed=12,
                     prompt=[50258, 50259, 50359], enc_rows=[0, 1, 499, 750, 1000, 1499],
                     sample_windows=32),
    # BASELINE configs[3] model: all 32 layers of large-v3, two windows
    "large_v3_full": dict(model="large-v3", windows=2, steps=12, weight_seed=0, mel_seed=21,
                          prompt=[50258, 50259, 50359], enc_rows=list(range(0, 1500, 100)) + [1499],
                          sample_windows=2),
}


def case_inputs(case):
    """Weights (bf16-rounded float32) and mel windows of a case; shared with the GPU test."""
    dims = om.Dims.named(case["model"])
    w = om.random_weights(dims, seed=case["weight_seed"], style="peaked")
    for k in list(w):
        w[k] = w[k].to(torch.bfloat16).float()
    g = torch.Generator().manual_seed(case["mel_seed"])
    mel = torch.rand(case["windows"], 3000, dims.n_mels, generator=g) * 2 - 1
    return dims, w, mel


def make(name):
    case = CASES[name]
    t0 = time.time()
    dims, w, mel = case_inputs(case)
    oracle = om.Whisper(dims, w)
    prompt = torch.tensor([case["prompt"]] * case["windows"], dtype=torch.long)
    with torch.no_grad():
        enc = torch.cat([oracle.encoder(mel[i: i + 4]) for i in range(0, case["windows"], 4)])
        print("%s: encoder %.1f s" % (name, time.time() - t0), flush=True)
        out = om.greedy(oracle, enc=enc, prompt=prompt, max_new_tokens=case["steps"], eos=50257,
                        pad=50257, early_stopping=False, return_logits=True)
    lp = torch.log_softmax(out["logits"], dim=-1)
    top2 = lp.topk(2, dim=-1).values
    margins = (top2[..., 0] - top2[..., 1]).numpy().astype(np.float32)
    rows = np.array(case["enc_rows"], dtype=np.int32)
    sw = case["sample_windows"]
    np.savez_compressed(
        os.path.join(HERE, "whisper_%s.npz" % name),
        model=np.array(case["model"]), weight_seed=np.int64(case["weight_seed"]),
        mel_seed=np.int64(case["mel_seed"]), prompt=np.array(case["prompt"], dtype=np.int32),
        enc_rows=rows, enc_samples=enc[:sw, rows.tolist()].numpy().astype(np.float32),
        enc_mean=enc.mean(dim=(1, 2)).numpy().astype(np.float32),
        enc_std=enc.std(dim=(1, 2)).numpy().astype(np.float32),
        tokens=out["tokens"].numpy().astype(np.int32), scores=out["scores"].numpy().astype(np.float32),
        margins=margins, logit_std=out["logits"].std(dim=-1).numpy().astype(np.float32))
    print("%s: done in %.1f s, %d steps with a margin < 0.05" % (name, time.time() - t0,
                                                                 int((margins < 0.05).sum())), flush=True)


if __name__ == "__main__":
    torch.set_num_threads(os.cpu_count() or 8)
    for n in (sys.argv[1:] or list(CASES)):
        make(n)
