"""stt_decode_step (one cached decoder step, selection in Python) and what is built on it: arbitrary
callable logits filters and per-step scores under beam search, against the device loop of
stt_generate (same kernels, so greedy ids must be identical) and against oracle/model.py.  Also the
engine's behaviour when a re-allocation fails and when an earlier call left non-finite data behind."""
import numpy as np
import pytest
import torch

from oracle import model as om
from tests import _parity as P

pytestmark = pytest.mark.gpu

EOS = PAD = 999


@pytest.fixture(scope="module")
def small():
    from speech_to_text_amd.engine import WhisperEngine
    dims = om.Dims(n_mels=80, n_audio_ctx=1500, n_audio_state=128, n_audio_head=2, n_audio_layer=2,
                   n_vocab=1000, n_text_ctx=64, n_text_state=128, n_text_head=2, n_text_layer=2)
    w = P.bf16_round(om.random_weights(dims, seed=0, style="peaked"))
    engine = WhisperEngine(P.dims_dict(dims), w, max_batch=3, max_beams=5, max_text_len=48, max_prompt_len=8)
    oracle = om.Whisper(dims, w)
    g = torch.Generator().manual_seed(1)
    mel = (torch.rand(3, 3000, 80, generator=g) * 2 - 1)
    with torch.no_grad():
        enc = oracle.encoder(mel)
    return dims, engine, oracle, mel, enc


PROMPT = np.array([[1, 2, 3]] * 3, dtype=np.int32)


def test_step_logits_equal_teacher_forced_pass(small):
    """prompt in one step, then three single tokens: the logits of every step are those of the
    teacher-forced pass over the same tokens (oracle tolerance), and a second history started with
    reset gives the same bits as the first."""
    dims, engine, oracle, mel, enc = small
    engine.encode(mel)
    toks = np.random.default_rng(3).integers(0, 998, size=(3, 6)).astype(np.int32)
    with torch.no_grad():
        ref, _ = oracle.decoder(torch.from_numpy(toks).long(), oracle.cross_kv(enc))
    runs = []
    for _ in range(2):
        got = [engine.decode_step(toks[:, :3], reset=True)]
        for j in range(3, 6):
            got.append(engine.decode_step(toks[:, j: j + 1]))
        runs.append(torch.stack(got, dim=1).cpu().numpy())
    assert np.array_equal(runs[0], runs[1])
    assert P.rel_l2(runs[0], ref[:, 2:].numpy()) < 2e-2


def test_step_argument_checks(small):
    from speech_to_text_amd._lib import SttError
    dims, engine, oracle, mel, enc = small
    engine.encode(mel)
    with pytest.raises(SttError):
        engine.decode_step(PROMPT[:2], reset=True)              # 2 sequences for 3 encoded windows
    engine.decode_step(PROMPT, reset=True)
    with pytest.raises(SttError):
        engine.decode_step(np.zeros((6, 1), np.int32), seq_per_window=2)     # another history size
    with pytest.raises(SttError):
        engine.decode_step(np.zeros((3, 46), np.int32))         # 3 + 46 > 48 positions
    engine.encode(mel)                                          # an encode ends the history
    with pytest.raises(SttError):
        engine.decode_step(np.zeros((3, 1), np.int32))


def test_eager_greedy_equals_device_loop(small):
    """No filter: the Python loop over decode_step and the device loop run the same kernels on the
    same inputs, so the ids are identical and the scores differ by log-softmax rounding only."""
    from speech_to_text_amd import eager
    dims, engine, oracle, mel, enc = small
    engine.encode(mel)
    want = engine.generate(PROMPT, max_new_tokens=20, eos_token=EOS, pad_token=PAD, early_stopping=False)
    got = eager.generate(engine, PROMPT, max_new_tokens=20, eos_token=EOS, pad_token=PAD, early_stopping=False)
    assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])
    assert np.abs(got[2] - want[2]).max() < 1e-3


def _parity_filter(scores, tokens, state, **_):
    """Something no index list or timestamp rule expresses: at step t only ids with the parity of
    t + (last token) survive; row 1 additionally may not repeat any token it already produced."""
    t = int(state.t)
    ids = torch.arange(scores.shape[-1], device=scores.device)[None]
    last = tokens[:, t - 1] if t > 0 else torch.zeros_like(tokens[:, 0])
    keep = (ids % 2) == ((t + last) % 2)[:, None]
    scores = scores.masked_fill(~keep, float("-inf"))
    if t > 0:
        seen = torch.zeros_like(scores, dtype=torch.bool)
        seen[1].index_fill_(0, tokens[1, :t], True)
        scores = scores.masked_fill(seen, float("-inf"))
    return scores


def test_callable_filter_greedy_against_oracle(small):
    """An arbitrary Python callable as `logits_filter` (generation_utils.py:518-520) through
    WhisperEngine.generate: every step's choice is the oracle's under the same filter."""
    dims, engine, oracle, mel, enc = small
    engine.encode(mel)
    seen_kwargs = {}

    def flt(scores, tokens, state, **kw):
        seen_kwargs.update(kw)
        assert scores.is_cuda and tokens.is_cuda and tokens.shape == (3, 16)
        assert (state.state is None) == (int(state.t) == 0)
        return _parity_filter(scores, tokens, state)

    tokens, lengths, scores = engine.generate(PROMPT, max_new_tokens=16, eos_token=EOS, pad_token=PAD,
                                              early_stopping=False, logits_filter=flt,
                                              filter_kwargs={"lang": "en"})
    assert seen_kwargs == {"lang": "en"}
    for t in range(16):
        par = (t + (tokens[:, t - 1] if t else 0)) % 2
        assert ((tokens[:, t] % 2) == par).all()
    assert len(set(tokens[1].tolist())) == 16
    # the oracle under the same rule, teacher-forced on the engine's ids
    cross = oracle.cross_kv(enc)
    gen = torch.from_numpy(tokens).long()
    inputs, cache, offset = torch.from_numpy(PROMPT).long(), None, 0
    exact = 0
    with torch.no_grad():
        for t in range(16):
            logits, cache = oracle.decoder(inputs, cross, cache, offset)
            offset += inputs.shape[1]
            z = logits[:, -1]
            st = type("S", (), {"t": t, "state": None if t == 0 else 1})()
            lp = torch.log_softmax(_parity_filter(z.clone(), gen, st), dim=-1)
            for i in range(3):
                best = int(lp[i].argmax())
                if best == tokens[i, t]:
                    exact += 1
                else:
                    assert float(lp[i, best] - lp[i, tokens[i, t]]) < P.row_margin(z[i]), (i, t)
            inputs = gen[:, t: t + 1]
    assert exact >= 44, exact


def test_beam_return_logits_and_timestamp_rules(small):
    """`return_logits` under beam search (generation_utils.py:422-428) runs the Python loop; with the
    Whisper timestamp rules as a torch function it must pick the beams the device loop picks, and
    the kept rows are the log-probabilities the chosen tokens were scored with."""
    from speech_to_text_amd.engine import TimestampFilter
    dims, engine, oracle, mel, enc = small
    engine.encode(mel)
    filt = TimestampFilter(remove_tokens=[10, 11, 12], remove_tokens_with_space=[10, 11, 12, 13, 900],
                           timestamp_begin_idx=920, eos_token_idx=900)
    kw = dict(method="beam", num_beams=4, num_sentences=2, max_new_tokens=10, eos_token=900, pad_token=900,
              early_stopping=False, logits_filter=filt)
    want = engine.generate(PROMPT, **kw)
    tokens, lengths, scores, logits = engine.generate(PROMPT, return_logits=True, **kw)
    assert logits.shape == (3, 2, 10, 1000) and logits.is_cuda
    if np.array_equal(tokens, want[0]):
        assert np.array_equal(lengths, want[1]) and np.abs(scores - want[2]).max() < 1e-3
    else:               # two candidates closer than float32 log-softmax rounding may swap: the scores still agree
        assert np.abs(np.sort(scores, axis=1) - np.sort(want[2], axis=1)).max() < 1e-3
    lg = logits.cpu().numpy()
    picked = np.take_along_axis(lg, tokens[..., None].astype(np.int64), axis=-1)[..., 0]
    # a sequence's score is the sum of its steps up to and including its EOS (pads add nothing)
    ended = np.cumsum(tokens == 900, axis=-1) - (tokens == 900) > 0
    assert np.isfinite(picked[~ended]).all()
    assert np.abs(np.where(ended, 0.0, picked).sum(axis=-1) - scores).max() < 1e-3
    assert (tokens[:, :, 0] == 920).all()                            # the only id the first step allows
    assert np.isneginf(lg[:, :, 0, :920]).all() and np.isneginf(lg[:, :, 0, 921:]).all()


def test_runtime_accepts_any_callable(small):
    """Through the Runtime interface, as models/stt/whisper.py:247-249 calls it."""
    from speech_to_text_amd.runtimes import build_runtime
    rt = build_runtime("mi355x", path="random:tiny:3")
    rt.set_tokens(sos_token=50258, eos_token=50257, pad_token=50257)
    mel = torch.rand(2, 3000, 80) * 2 - 1
    calls = []

    def only_small_ids(scores, tokens, state, **kw):
        calls.append(int(state.t))
        return scores.masked_fill(torch.arange(scores.shape[-1], device=scores.device)[None] >= 100, float("-inf"))

    out = rt(mel, tokens=np.array([[50258, 50259, 50359]] * 2), max_new_tokens=6, logits_filter=only_small_ids,
             early_stopping=False, lang="en")
    assert calls == list(range(6))
    assert out.tokens.shape == (2, 6) and (out.tokens < 100).all()
    out = rt(mel, tokens=np.array([[50258, 50259, 50359]] * 2), max_new_tokens=4, method="beam", num_beams=3,
             return_logits=True, early_stopping=False)
    assert out.logits.shape == (2, 1, 4, rt.vocab_size)


def test_failed_reserve_leaves_a_working_engine():
    """A capacity that cannot fit the card: the call fails with an error, the engine goes back to
    the capacity it had and still decodes (after a new encode), bit-identically."""
    from speech_to_text_amd.engine import WhisperEngine
    from speech_to_text_amd._lib import SttError
    dims = om.Dims(n_mels=80, n_audio_ctx=1500, n_audio_state=128, n_audio_head=2, n_audio_layer=2,
                   n_vocab=1000, n_text_ctx=64, n_text_state=128, n_text_head=2, n_text_layer=2)
    w = P.bf16_round(om.random_weights(dims, seed=0, style="peaked"))
    engine = WhisperEngine(P.dims_dict(dims), w, max_batch=2, max_beams=1, max_text_len=32, max_prompt_len=8)
    mel = torch.rand(2, 3000, 80, generator=torch.Generator().manual_seed(4)) * 2 - 1
    engine.encode(mel)
    before = engine.generate(PROMPT[:2], max_new_tokens=8, eos_token=EOS, pad_token=PAD, early_stopping=False)
    with pytest.raises(SttError) as err:
        engine.reserve(max_batch=400000)             # ~1.4 TB for the encoder's activations alone
    assert "stt_engine_reserve" in str(err.value)
    assert engine.usable() and engine.max_batch == 2
    with pytest.raises(SttError):                    # the encoded batch is gone, and the call says so
        engine.generate(PROMPT[:2], max_new_tokens=8, eos_token=EOS, pad_token=PAD, early_stopping=False)
    engine.encode(mel)
    after = engine.generate(PROMPT[:2], max_new_tokens=8, eos_token=EOS, pad_token=PAD, early_stopping=False)
    assert np.array_equal(before[0], after[0]) and np.array_equal(before[2], after[2])


def test_smaller_batch_after_non_finite_window(small):
    """Window 2 of a three-window call is NaN; the next call encodes two windows.  The attention
    kernel's last key tile reads a few positions past the batch in the V^T image (under zero
    weights): what the NaN window left there must not reach the new call."""
    dims, engine, oracle, mel, enc = small
    bad = mel.clone()
    bad[2] = float("nan")
    engine.encode(bad)
    got = engine.encode(mel[:2], return_output=True).cpu().numpy()
    assert np.isfinite(got).all()
    clean = engine.encode(mel[:2], return_output=True).cpu().numpy()
    assert np.array_equal(got, clean)


def test_per_window_token_limits_equal_separate_calls(small):
    """`max_new_tokens_per_window`: a window ends after its own number of new tokens exactly as a call
    of its own with that max_new_tokens does (same ids, lengths and scores; pad afterwards), for
    greedy and for beam search; and with early stopping on, where ended windows are skipped by the
    attention kernels, the windows still running are not disturbed."""
    dims, engine, oracle, mel, enc = small
    engine.encode(mel)
    caps = [5, 12, 9]
    kw = dict(eos_token=EOS, pad_token=PAD)
    free = engine.generate(PROMPT, max_new_tokens=12, early_stopping=False, **kw)
    for early in (False, True):
        tok, lens, sc = engine.generate(PROMPT, max_new_tokens=12, early_stopping=early,
                                        max_new_tokens_per_window=caps, **kw)
        for i, c in enumerate(caps):
            assert np.array_equal(tok[i, :c], free[0][i, :c]) and (tok[i, c:] == PAD).all(), (early, i)
            assert lens[i] == c
            alone = engine.generate(PROMPT, max_new_tokens=c, early_stopping=False, **kw)
            assert np.array_equal(alone[0][i], tok[i, :c]) and alone[2][i] == sc[i], (early, i)
    bkw = dict(method="beam", num_beams=3, num_sentences=3, **kw)
    tok, lens, sc = engine.generate(PROMPT, max_new_tokens=12, early_stopping=True,
                                    max_new_tokens_per_window=caps, **bkw)
    for i, c in enumerate(caps):
        alone = engine.generate(PROMPT, max_new_tokens=c, early_stopping=False, **bkw)
        assert np.array_equal(alone[0][i], tok[i, :, :c]) and (tok[i, :, c:] == PAD).all(), i
        assert np.array_equal(alone[1][i], lens[i]) and np.array_equal(alone[2][i], sc[i]), i


def test_ended_windows_are_skipped_without_changing_results(small):
    """Mixed lengths through a real EOS: window 0 ends at its 4th token, the others run on.  With early
    stopping on the attention kernels skip window 0 from then on; ids, lengths and scores of every
    window equal those of the run that computes everything (early stopping off) up to the step
    at which the last window ends."""
    dims, engine, oracle, mel, enc = small
    engine.encode(mel)
    free, _, _ = engine.generate(PROMPT, max_new_tokens=16, eos_token=EOS, pad_token=PAD, early_stopping=False)
    others = set(free[1].tolist()) | set(free[2].tolist())
    at = next(j for j in range(2, 12) if int(free[0, j]) not in others and int(free[0, j]) not in free[0, :j].tolist())
    eos = int(free[0, at])
    full = engine.generate(PROMPT, max_new_tokens=16, eos_token=eos, pad_token=PAD, early_stopping=False)
    skip = engine.generate(PROMPT, max_new_tokens=16, eos_token=eos, pad_token=PAD, early_stopping=True,
                           check_every=5)
    assert full[1].tolist() == [at, 16, 16]
    for a, b in zip(full, skip):
        assert np.array_equal(a, b)


def test_attention_weights_against_oracle(small):
    """stt_decode_step_attention: the weights of every decoder layer for the prompt pass and for
    single-token steps are the oracle's softmax(q k^T / 8): within 5e-3 + 5 % of the oracle that
    rounds to bf16 where the engine does, and within 0.1 absolute of the float32 oracle (the peaked
    test weights give scores of tens of units, on which the 0.4 % of bf16 q and k move a weight of
    0.5 by several hundredths from the second layer on: storage noise, as the rounding oracle shows); rows sum to one, entries past a token's own
    position are zero, and asking for the weights changes no logit bit of a single-token step."""
    dims, engine, oracle, mel, enc = small
    engine.encode(mel)
    toks = np.random.default_rng(5).integers(0, 998, size=(3, 6)).astype(np.int32)
    rounded = om.Whisper(dims, oracle.w, round_acts=True)
    with torch.no_grad():
        _, _, ref = oracle.decoder(torch.from_numpy(toks).long(), oracle.cross_kv(enc), return_attention=True)
        _, _, ref_r = rounded.decoder(torch.from_numpy(toks).long(), rounded.cross_kv(enc), return_attention=True)

    def close(got, key, index):
        got = got.cpu().numpy()
        assert np.abs(got - ref[key][index].numpy()).max() < 0.1
        want = ref_r[key][index].numpy()
        assert (np.abs(got - want) - 0.05 * want).max() < 5e-3

    plain = [engine.decode_step(toks[:, :3], reset=True)] + [engine.decode_step(toks[:, j: j + 1]) for j in (3, 4, 5)]
    lg, own, cross = engine.decode_step(toks[:, :3], reset=True, attention="all")
    assert own.shape == (2, 3, 2, 3, 3) and cross.shape == (2, 3, 2, 3, 1500)
    # 3 queries per window: the plain pass runs the matrix-core cross-attention kernel, the pass that
    # returns weights the vector-unit one (same sums in another order)
    assert P.rel_l2(lg.cpu().numpy(), plain[0].cpu().numpy()) < 1e-2
    for layer in range(2):
        close(own[layer], "attn_layer_%d" % layer, (slice(None), slice(None), slice(0, 3), slice(0, 3)))
        close(cross[layer], "enc_attn_layer_%d" % layer, (slice(None), slice(None), slice(0, 3)))
    assert float(own[:, :, :, 0, 1:].abs().max()) == 0.0 and float(own[:, :, :, 1, 2].abs().max()) == 0.0
    engine.decode_step(toks[:, :3], reset=True)      # the history of `plain` again (its prompt pass included)
    for j in (3, 4, 5):
        lg, own, cross = engine.decode_step(toks[:, j: j + 1], attention="all")
        assert torch.equal(lg, plain[j - 2])
        assert own.shape == (2, 3, 2, 1, j + 1)
        for layer in range(2):
            close(own[layer, :, :, 0], "attn_layer_%d" % layer, (slice(None), slice(None), j, slice(0, j + 1)))
            close(cross[layer, :, :, 0], "enc_attn_layer_%d" % layer, (slice(None), slice(None), j))
        assert np.abs(own.sum(-1).cpu().numpy() - 1).max() < 1e-4 and np.abs(cross.sum(-1).cpu().numpy() - 1).max() < 1e-4
    lg, own, cross = engine.decode_step(toks[:, :1], attention="last")       # 'last' = the last layer only
    assert own.shape[0] == 1 and cross.shape[0] == 1


def test_exported_state_against_oracle_cache(small):
    """stt_engine_export_state: the self-attention cache after a few steps and the encoder-attention
    K/V are the oracle's (bf16 storage: 1e-2 relative)."""
    dims, engine, oracle, mel, enc = small
    engine.encode(mel)
    toks = np.random.default_rng(6).integers(0, 998, size=(3, 5)).astype(np.int32)
    with torch.no_grad():
        cross = oracle.cross_kv(enc)
        _, cache = oracle.decoder(torch.from_numpy(toks).long(), cross)
    engine.decode_step(toks[:, :3], reset=True)
    engine.decode_step(toks[:, 3:4])
    engine.decode_step(toks[:, 4:5])
    assert engine.step_state() == (3, 1, 5)
    for layer in range(2):
        (k, v), (ek, ev) = engine.export_state(layer)
        assert k.shape == (3, 2, 5, 64) and ek.shape == (3, 2, 1500, 64)
        assert P.rel_l2(k.cpu().numpy(), cache[layer][0].numpy()) < 1e-2
        assert P.rel_l2(v.cpu().numpy(), cache[layer][1].numpy()) < 1e-2
        assert P.rel_l2(ek.cpu().numpy(), cross[layer][0].numpy()) < 1e-2
        assert P.rel_l2(ev.cpu().numpy(), cross[layer][1].numpy()) < 1e-2


def test_runtime_state_attention_and_continuation():
    """`return_state`, `return_last_attention`, `return_attention` and `initial_state` through the
    Runtime call (generation_utils.py:84-87, 293-300): the ids are those of the plain call; a loop
    continued from the returned state produces the ids the uninterrupted loop produces; a state that
    the engine has moved past is refused."""
    from speech_to_text_amd.runtimes import build_runtime
    rt = build_runtime("mi355x", path="random:tiny:3:peaked")
    rt.set_tokens(sos_token=50258, eos_token=50257, pad_token=50257)
    mel = torch.rand(2, 3000, 80, generator=torch.Generator().manual_seed(2)) * 2 - 1
    prompt = np.array([[50258, 50259, 50359]] * 2)
    whole = rt(mel, tokens=prompt, max_new_tokens=10, early_stopping=False)
    first = rt(mel, tokens=prompt, max_new_tokens=6, early_stopping=False, return_state=True,
               return_last_attention=True)
    assert np.array_equal(first.tokens, whole.tokens[:, :6])
    layers = rt.dims["n_text_layer"]
    att = first.attention_weights
    assert sorted(att) == ["attn_layer_%d" % (layers - 1), "enc_attn_layer_%d" % (layers - 1)]
    heads = rt.dims["n_text_head"]
    assert att["attn_layer_%d" % (layers - 1)].shape == (2, heads, 8, 8)          # 3 + 5 tokens fed
    assert att["enc_attn_layer_%d" % (layers - 1)].shape == (2, heads, 8, 1500)
    assert float((att["enc_attn_layer_%d" % (layers - 1)].sum(-1) - 1).abs().max()) < 1e-4
    st = first.state
    assert len(st) == layers and st.length == 8 and st.valid()
    (k, v), (ek, ev) = st["layer_0"]
    assert k.shape == (2, heads, 8, 64) and ek.shape == (2, heads, 1500, 64)
    rest = rt(None, tokens=first.tokens[:, -1:], initial_state=st, max_new_tokens=4, early_stopping=False)
    assert np.array_equal(rest.tokens, whole.tokens[:, 6:10])
    assert not st.valid()
    with pytest.raises(ValueError):
        rt(None, tokens=first.tokens[:, -1:], initial_state=st, max_new_tokens=2)
    with pytest.raises(RuntimeError):
        st["layer_0"]
    every = rt(mel, tokens=prompt, max_new_tokens=3, early_stopping=False, return_attention=True)
    assert len(every.attention_weights) == 2 * layers
    assert np.array_equal(every.tokens, whole.tokens[:, :3])


def test_beam_attention_follows_the_beam_index(small):
    """Under beam search the accumulated weights are gathered by the beam index at every step
    (generation_utils.py:580-587): the rows returned for a final beam are those of one teacher-forced
    pass over that beam's own tokens."""
    from speech_to_text_amd import eager
    dims, engine, oracle, mel, enc = small
    engine.encode(mel)
    extras = {}
    toks, lens, scs = eager.generate(engine, PROMPT, method="beam", num_beams=3, num_sentences=3, max_new_tokens=5,
                                     eos_token=EOS, pad_token=PAD, early_stopping=False, return_last_attention=True,
                                     return_state=True, extras=extras)
    att = extras["attention_weights"]
    assert att["attn_layer_1"].shape == (9, 2, 7, 7) and att["enc_attn_layer_1"].shape == (9, 2, 7, 1500)
    (k, v), _ = extras["state"]["layer_1"]
    assert k.shape == (9, 2, 7, 64)
    fed = np.concatenate([np.repeat(PROMPT, 3, axis=0), toks.reshape(9, 5)[:, :4]], axis=1).astype(np.int32)
    engine.encode(mel)
    _, own, cross = engine.decode_step(fed, seq_per_window=3, reset=True, attention="last")
    # (one pass over 7 tokens against 5 passes: other GEMM and attention kernels fill the cache, and the
    # peaked weights amplify their rounding as in test_attention_weights_against_oracle)
    for got, want in ((att["attn_layer_1"], own[0]), (att["enc_attn_layer_1"], cross[0])):
        assert float(((got - want).abs() - 0.05 * want).max()) < 5e-3
    # ... while the beams of a window do differ by more than that, so a wrong gather would show
    assert float((att["attn_layer_1"][0] - att["attn_layer_1"][1]).abs().max()) > 2e-2
    (k2, v2), _ = engine.export_state(1)
    assert P.rel_l2(k.cpu().numpy(), k2.cpu().numpy()) < 1e-2 and P.rel_l2(v.cpu().numpy(), v2.cpu().numpy()) < 1e-2
