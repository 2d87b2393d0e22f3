"""Shared helpers of the GPU parity tests: the HIP engine against oracle/model.py.

Stated tolerance of the bf16 engine against the float32 oracle: teacher-forced logits within a
relative L2 of 2e-2.  Measured on the weights these tests use it is 3e-3 .. 1e-2, and the oracle
with its activations rounded to bf16 at the engine's storage points (`round_acts=True`) sits just
as far from the float32 oracle as the engine does: that distance is what bf16 storage costs, so
it cannot be tested away.  A free-running comparison of two decodes therefore stops meaning
anything after the first near-tie.  Both checkers below follow the ENGINE's history and ask the
oracle, step by step, whether each choice the engine made is the oracle's choice or loses to it
by less than the near-tie allowance: every step of every sequence is checked, and a wrong token
that is not a near-tie fails.

Near-tie allowance: MARGIN_REL x the standard deviation of the oracle's logits row.  A logit
error of 0.02 std (the stated tolerance) on each of two candidates moves their gap by up to
0.04 std; a lead smaller than that cannot be called either way.  On these weights that is
0.14 (V = 1000 test model) to 0.43 nats (large-v3 width) against a median lead of 1.4 - 2 nats.
"""
import json
import os

import numpy as np
import torch

from oracle import model as om

MARGIN_REL = 0.04


# Measured "engine token == oracle argmax" counts per test, committed next to the fixtures.  A GPU
# run checks each count against the committed one (one step of slack: a change of rounding order
# may move a single near-tie) and writes what it measured to gpurun_out/token_parity_measured.json,
# which is copied over the committed file when a kernel change moves a count on purpose.
_RECORD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "token_parity.json")
_MEASURED = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "gpurun_out",
                         "token_parity_measured.json")


def record_exact(name, exact, total):
    """Asserts that `exact` of `total` steps is no worse than the committed measurement minus one
    (or, for a test without a committed entry yet, at least 90 % of the steps) and records it."""
    known = {}
    if os.path.exists(_RECORD):
        with open(_RECORD) as f:
            known = json.load(f)
    if name in known and known[name]["total"] == total:
        assert exact >= known[name]["exact"] - 1, "%s: %d of %d steps equal the oracle's argmax, %d were recorded" % (
            name, exact, total, known[name]["exact"])
    else:
        assert exact >= 0.9 * total, (name, exact, total)
    measured = {}
    try:
        if os.path.exists(_MEASURED):
            with open(_MEASURED) as f:
                measured = json.load(f)
        measured[name] = {"exact": int(exact), "total": int(total)}
        os.makedirs(os.path.dirname(_MEASURED), exist_ok=True)
        with open(_MEASURED, "w") as f:
            json.dump(measured, f, indent=1, sort_keys=True)
    except OSError:
        pass                                     # read-only tree: the assertion above is what counts


def row_margin(logits_row):
    """Near-tie allowance in nats for one row of oracle logits."""
    return MARGIN_REL * float(logits_row.std())

DIM_KEYS = ("n_mels", "n_audio_ctx", "n_audio_state", "n_audio_head", "n_audio_layer", "n_vocab",
            "n_text_ctx", "n_text_state", "n_text_head", "n_text_layer")


def dims_dict(d):
    return {k: getattr(d, k) for k in DIM_KEYS}


def bf16_round(w):
    return {k: v.to(torch.bfloat16).float() for k, v in w.items()}


def rel_l2(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / np.linalg.norm(b))


def check_greedy_steps(oracle, enc, prompt, tokens, eos, pad, logits_filter=None,
                       temperature=None, length_temperature=None, scores=None, engine_logits=None):
    """Teacher-forces the oracle on the engine's greedy tokens [B, steps].  At every step of
    every unfinished sequence the engine's token must be the oracle's argmax or lose to it by
    less than the near-tie allowance; finished sequences must continue with `pad`.  `scores` [B] (the engine's
    summed log-probabilities) are compared with the oracle's sum over the engine's tokens;
    `engine_logits` [B, steps, V] (the engine's processed scores, return_logits) with the oracle's
    log-probabilities of the same step: the same entries at -inf, the others within the near-tie
    allowance as an rms over the row and within three times that each.
    Returns (steps where the token is exactly the oracle's argmax, steps checked)."""
    tokens = np.asarray(tokens)
    b, steps = tokens.shape
    cross = oracle.cross_kv(enc)
    cache, offset, inputs = None, 0, torch.from_numpy(np.asarray(prompt)).long()
    lengths = torch.full((b,), inputs.shape[1])
    finished = torch.zeros(b, dtype=torch.bool)
    gen = torch.from_numpy(tokens).long()
    want, slack = np.zeros(b), np.zeros(b)
    exact = total = 0
    for t in range(steps):
        logits, cache = oracle.decoder(inputs, cross, cache, offset)
        offset += inputs.shape[1]
        lp = om.process_logits(logits[:, -1], lengths, gen, t, temperature=temperature,
                               length_temperature=length_temperature, logits_filter=logits_filter)
        for i in range(b):
            if finished[i]:
                assert tokens[i, t] == pad, (i, t, tokens[i, t])
                continue
            total += 1
            z = logits[i, -1] / (temperature or 1.0)
            if length_temperature is not None:
                z = z * float(lengths[i] + 1) ** length_temperature
            margin = row_margin(z)
            if engine_logits is not None:
                got, ref = np.asarray(engine_logits[i, t]), lp[i].numpy()
                assert np.array_equal(np.isneginf(got), np.isneginf(ref)), (i, t)
                live = ~np.isneginf(ref)
                # a log-probability is a logit minus the row's normaliser, each good to the stated
                # 0.02 std: 0.04 std as an rms over the row (a shift of the normaliser moves every
                # entry alike); the worst of ~1000 entries sits near 0.08 std on either side of any
                # change of rounding order, so the bound on it is 0.12 std
                err = np.abs(got[live] - ref[live])
                assert np.sqrt((err ** 2).mean()) < margin, (i, t, np.sqrt((err ** 2).mean()), margin)
                assert err.max() < 3 * margin, (i, t, err.max(), margin)
            best = int(lp[i].argmax())
            if best == tokens[i, t]:
                exact += 1
            else:
                lost = float(lp[i, best] - lp[i, tokens[i, t]])
                assert lost < margin, "sequence %d step %d: engine %d, oracle %d leads by %.3f nats (allowance %.3f)" % (
                    i, t, tokens[i, t], best, lost, margin)
            want[i] += float(lp[i, tokens[i, t]])
            slack[i] += margin / 2
        tok = gen[:, t]
        finished = finished | (tok == eos)
        lengths = lengths + (~finished).long()
        inputs = tok[:, None]
    if scores is not None:
        assert (np.abs(want - np.asarray(scores)) < slack + 0.02).all(), (want, scores, slack)
    return exact, total


def check_beam_steps(engine, oracle, enc, prompt, num_beams, steps, eos, pad,
                     engine_filter=None, oracle_filter=None, length_power=None,
                     length_temperature=None):
    """Beam search, one transition at a time (generation_utils.py:333-438).  The engine is run
    with max_new_tokens = 1 .. steps and every beam returned; for each step the oracle scores
    all num_beams * V continuations of the engine's beams of the step before and the engine's
    new beams must be the oracle's best num_beams continuations, in the oracle's order, where
    candidates closer than the near-tie allowance count as tied.  Checks every window; returns the number of
    transitions checked."""
    prompt = np.asarray(prompt, dtype=np.int32)
    b, s = prompt.shape
    nb, v = num_beams, oracle.dims.n_vocab
    kw = dict(method="beam", num_beams=nb, num_sentences=nb, eos_token=eos, pad_token=pad,
              early_stopping=False, logits_filter=engine_filter, length_power=length_power,
              length_temperature=length_temperature)
    prev_tok = np.zeros((b, nb, 0), dtype=np.int32)
    prev_sc = np.zeros((b, nb), dtype=np.float32)
    checked = 0
    all_cross = oracle.cross_kv(enc)
    for t in range(steps):
        tok, lens, sc = engine.generate(prompt, max_new_tokens=t + 1, **kw)
        assert tok.shape == (b, nb, t + 1)
        for w in range(b):
            cross = [(k[w: w + 1].expand(nb, -1, -1, -1), vv[w: w + 1].expand(nb, -1, -1, -1))
                     for k, vv in all_cross]
            hist = torch.from_numpy(prev_tok[w]).long()                       # [nb, t]
            full = torch.cat([torch.from_numpy(prompt[w: w + 1]).long().expand(nb, -1), hist], dim=1)
            logits, _ = oracle.decoder(full, cross)
            if t == 0:
                finished, n_gen = torch.zeros(nb, dtype=torch.bool), torch.zeros(nb, dtype=torch.long)
            else:
                is_eos = hist == eos
                finished = is_eos.any(dim=1)
                n_gen = torch.where(finished, is_eos.float().argmax(dim=1), torch.tensor(t))
            lengths = s + n_gen
            lp = om.process_logits(logits[:, -1], lengths, hist, t, length_temperature=length_temperature,
                                   logits_filter=oracle_filter)
            eos_row = torch.full((v,), float("-inf"))
            eos_row[pad] = 0.0
            lws = torch.from_numpy(prev_sc[w])[:, None] + torch.where(finished[:, None], eos_row[None], lp)
            r = lws if length_power is None else lws / ((lengths + 1).float()[:, None] ** length_power)
            if t == 0:
                r, lws = r[:1], lws[:1]
            flat = r.reshape(-1)
            z = logits[:, -1]
            if length_temperature is not None:
                z = z * ((lengths + 1).float() ** length_temperature)[:, None]
            margin = row_margin(z)
            kth = float(torch.topk(flat, nb).values[-1])      # -inf while fewer than nb candidates are finite
            chosen = []
            for j in range(nb):
                # a new beam extends the engine's old beam with the same history (the best such
                # one when two old beams coincide)
                parents = [p for p in range(r.shape[0]) if np.array_equal(prev_tok[w, p], tok[w, j, :t])]
                assert parents, "window %d step %d beam %d does not extend a beam of the step before" % (w, t, j)
                new = int(tok[w, j, t])
                # a slot no finite candidate was left for: any token, score -inf
                dead = all(float(r[q, new]) == float("-inf") for q in parents)
                free = [q for q in parents if dead or (q, new) not in [c[0] for c in chosen]]
                assert free, "window %d step %d: candidate kept twice" % (w, t)
                p = max(free, key=lambda q: float(r[q, new]))
                val = float(r[p, new])
                assert (dead and kth == val) or val > kth - margin, "window %d step %d beam %d: candidate %.4f below the cut %.4f" % (
                    w, t, j, val, kth)
                if j and not dead:
                    assert val < chosen[-1][1] + margin, "window %d step %d: beams out of order" % (w, t)
                chosen.append(((p, new), val))
                if dead:
                    assert sc[w, j] == float("-inf") or sc[w, j] < -1e30, (w, t, j, sc[w, j])
                    continue
                assert abs(float(lws[p, new]) - float(sc[w, j])) < 2 * margin, (w, t, j, float(lws[p, new]), sc[w, j])
                # bookkeeping of the step (generation_utils.py:413-431)
                was_finished = bool(finished[p])
                if was_finished:
                    assert new == pad
                want_len = int(n_gen[p]) + (0 if (was_finished or new == eos) else 1)
                assert int(lens[w, j]) == want_len, (w, t, j, int(lens[w, j]), want_len)
            clear = (flat > kth + margin).nonzero()[:, 0].tolist()
            kept = {p * v + new for (p, new), _ in chosen}
            assert set(clear) <= kept, "window %d step %d: a clear winner was dropped" % (w, t)
            checked += 1
        prev_tok, prev_sc = tok, sc
    return checked
