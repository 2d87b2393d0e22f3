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
