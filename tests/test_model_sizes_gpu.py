"""Parity at the model sizes BASELINE.json names: whisper-base with 32 windows (configs[2]),
whisper-large-v3 (configs[3]: all 32 layers; configs[4]: beam 5 + timestamp filter), against
oracle/model.py on the same bf16-rounded seeded weights.

The oracle's encoder over 32 base windows / a 32-layer large-v3 stack takes minutes of CPU, so
its outputs come from the committed fixtures tests/golden/whisper_*.npz (written by
tests/golden/make_model_fixtures.py, which also defines the seeds); the decoder side is cheap
and runs live: the oracle is teacher-forced on the engine's own tokens (tests/_parity.py).

Stated tolerances (bf16 GEMM inputs, fp32 accumulation): encoder states relative L2 <= 1e-2 and
max abs <= 0.05 * std(reference); teacher-forced logits relative L2 <= 2e-2; token ids equal
to the oracle's choice at every step except where the oracle's own lead is under the near-tie
allowance of tests/_parity.py (0.04 x the standard deviation of the logits row).
"""
import os
import sys

import numpy as np
import pytest
import torch

from oracle import model as om
from tests import _parity as P

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
sys.path.insert(0, GOLDEN)


_FIXTURES = {}


def _fixture(name):
    """Seeded weights and mel windows of a fixture case (generated once per test session: the
    large-v3 weights are 1.5 G draws) with the oracle outputs committed for them."""
    if name not in _FIXTURES:
        _FIXTURES[name] = _load_fixture(name)
    return _FIXTURES[name]


def _load_fixture(name):
    import make_model_fixtures as mk
    f = np.load(os.path.join(GOLDEN, "whisper_%s.npz" % name))
    case = mk.CASES[name]
    assert int(f["weight_seed"]) == case["weight_seed"] and int(f["mel_seed"]) == case["mel_seed"]
    dims, w, mel = mk.case_inputs(case)
    return f, case, dims, w, mel


def _check_encoder_against_fixture(got, f):
    """got: engine encoder output [B, 1500, d] (numpy).  Fixture rows: relative L2 and max abs;
    whole windows: mean and standard deviation."""
    rows = f["enc_rows"].tolist()
    ref = f["enc_samples"]
    sub = got[: ref.shape[0], rows]
    std = float(f["enc_std"].mean())
    rel = P.rel_l2(sub, ref)
    assert rel < 1e-2, rel
    assert np.abs(sub - ref).max() < 0.05 * std, (np.abs(sub - ref).max(), std)
    assert np.abs(got.mean(axis=(1, 2)) - f["enc_mean"]).max() < 2e-3
    assert np.abs(got.std(axis=(1, 2)) / f["enc_std"] - 1).max() < 2e-3
    return rel


def _near_ties(f):
    """[windows, steps] bool: steps at which the oracle's own lead is below the near-tie allowance."""
    return f["margins"] < P.MARGIN_REL * f["logit_std"]


def _check_tokens_against_fixture(tokens, f):
    """Engine ids against the oracle's free-running greedy ids: identical up to (not including)
    the first step of a sequence where the oracle's own lead is below the allowance - from there on the
    two decodes may have parted for a legitimate reason and the teacher-forced check takes over.
    Returns (sequences identical to the oracle over all steps, sequences)."""
    ref, near = f["tokens"], _near_ties(f)
    same = 0
    for i in range(ref.shape[0]):
        close = np.nonzero(near[i])[0]
        upto = int(close[0]) if len(close) else ref.shape[1]
        assert np.array_equal(tokens[i, :upto], ref[i, :upto]), (i, tokens[i, :upto], ref[i, :upto])
        same += int(np.array_equal(tokens[i], ref[i]))
    return same, ref.shape[0]


def test_whisper_base_batch_32():
    """BASELINE configs[2]: whisper-base (d 512, 8 heads, 6 layers, V 51 865), 32 windows, greedy."""
    from speech_to_text_amd.engine import WhisperEngine
    f, case, dims, w, mel = _fixture("base_b32")
    steps = case["steps"]
    prompt = np.array([case["prompt"]] * case["windows"], dtype=np.int32)
    engine = WhisperEngine(P.dims_dict(dims), w, max_batch=32, max_beams=1,
                           max_text_len=len(case["prompt"]) + steps + 1, max_prompt_len=4)
    enc_gpu = engine.encode(mel, return_output=True).cpu()
    _check_encoder_against_fixture(enc_gpu.numpy(), f)
    tokens, lengths, scores = engine.generate(prompt, max_new_tokens=steps, eos_token=50257,
                                              pad_token=50257, early_stopping=False)
    assert tokens.shape == (32, steps)
    same, n = _check_tokens_against_fixture(tokens, f)
    near_ties = int(_near_ties(f).any(axis=1).sum())
    assert same >= n - near_ties, (same, n, near_ties)
    print("whisper-base x32: %d of %d windows identical to the oracle's free-running decode, "
          "%d windows hold a near-tie" % (same, n, near_ties))
    # every step of every window: the oracle, fed the engine's encoder output (checked above)
    # and the engine's history, agrees with the engine's choice
    oracle = om.Whisper(dims, w)
    with torch.no_grad():
        exact, total = P.check_greedy_steps(oracle, enc_gpu, prompt, tokens, 50257, 50257, scores=scores)
    assert total == 32 * steps
    P.record_exact("whisper_base_batch_32", exact, total)
    assert (lengths == steps).all()


def test_large_v3_all_layers():
    """BASELINE configs[3] model: large-v3 (d 1280, 20 heads, 32 + 32 layers, 128 mel bins,
    V 51 866), two windows: encoder states after 32 layers, greedy ids, teacher-forced logits."""
    from speech_to_text_amd.engine import WhisperEngine
    f, case, dims, w, mel = _fixture("large_v3_full")
    steps = case["steps"]
    prompt = np.array([case["prompt"]] * case["windows"], dtype=np.int32)
    engine = WhisperEngine(P.dims_dict(dims), w, max_batch=2, max_beams=1, max_text_len=24, max_prompt_len=16)
    enc_gpu = engine.encode(mel, return_output=True).cpu()
    _check_encoder_against_fixture(enc_gpu.numpy(), f)
    tokens, lengths, scores = engine.generate(prompt, max_new_tokens=steps, eos_token=50257,
                                              pad_token=50257, early_stopping=False)
    same, n = _check_tokens_against_fixture(tokens, f)
    if not _near_ties(f).any():             # no near-tie in the fixture: ids identical to the oracle's
        assert same == n
        assert np.abs(scores - f["scores"]).max() < 0.02 * float(f["logit_std"].mean()) * steps
    oracle = om.Whisper(dims, w)
    with torch.no_grad():
        exact, total = P.check_greedy_steps(oracle, enc_gpu, prompt, tokens, 50257, 50257, scores=scores)
        assert exact >= total - int(_near_ties(f).sum()) - 1
        P.record_exact("large_v3_all_layers", exact, total)
        # teacher-forced logits over 13 arbitrary tokens (26 rows: the skinny decode GEMMs at
        # N = 1280 / 3840 / 5120 and the ragged V = 51 866 logits tile)
        toks = np.random.default_rng(5).integers(0, dims.n_vocab, size=(2, 13)).astype(np.int32)
        got = engine.decoder_logits(toks).cpu().numpy()
        ref, _ = oracle.decoder(torch.from_numpy(toks).long(), oracle.cross_kv(enc_gpu))
    assert P.rel_l2(got, ref.numpy()) < 2e-2


@pytest.mark.timeout(900)
def test_large_v3_600_windows_per_call():
    """The call the benchmark makes: all 32 + 32 layers of large-v3 over 600 windows at once (buffers
    above 4 GiB, 900 000-row encoder GEMMs, 600-row decode GEMMs in five row groups).  The two fixture
    windows sit at the first, middle and last positions of the batch (0 | 299 | 598 and 1 | 300 |
    599) among 594 windows of noise: the copies must come out bit-identical wherever they sit, and at
    the last positions they must meet the fixture (encoder rows, greedy ids) and the oracle's
    step-by-step choice like the two-window call does."""
    from speech_to_text_amd.engine import WhisperEngine
    f, case, dims, w, mel2 = _fixture("large_v3_full")
    steps = case["steps"]
    n = 600
    engine = WhisperEngine(P.dims_dict(dims), w, max_batch=n, max_beams=1,
                           max_text_len=len(case["prompt"]) + steps + 1, max_prompt_len=4)
    g = torch.Generator(device="cuda").manual_seed(77)
    mel = torch.rand(n, 3000, dims.n_mels, generator=g, device="cuda") * 2 - 1
    spots = {0: (0, n // 2 - 1, n - 2), 1: (1, n // 2, n - 1)}
    for src, where in spots.items():
        for i in where:
            mel[i] = mel2[src].cuda()
    enc = engine.encode(mel, return_output=True)
    prompt = np.array([case["prompt"]] * n, dtype=np.int32)
    tokens, lengths, scores = engine.generate(prompt, max_new_tokens=steps, eos_token=50257,
                                              pad_token=50257, early_stopping=False)
    assert tokens.shape == (n, steps) and (lengths == steps).all()
    for src, where in spots.items():
        for i in where[1:]:
            assert torch.equal(enc[i], enc[where[0]]), "encoder output of window %d differs from its copy at %d" % (i, where[0])
            assert np.array_equal(tokens[i], tokens[where[0]]), (i, tokens[i], tokens[where[0]])
            assert scores[i] == scores[where[0]], (i, scores[i], scores[where[0]])
    assert len({tuple(r) for r in tokens.tolist()}) > 32              # the decode depends on the window's audio
    last = [n - 2, n - 1]
    enc_last = enc[last].cpu()
    del enc, mel
    _check_encoder_against_fixture(enc_last.numpy(), f)
    same, cnt = _check_tokens_against_fixture(tokens[last], f)
    if not _near_ties(f).any():
        assert same == cnt
    oracle = om.Whisper(dims, w)
    with torch.no_grad():
        exact, total = P.check_greedy_steps(oracle, enc_last, prompt[:2], tokens[last], 50257, 50257,
                                            scores=scores[last])
    assert total == 2 * steps and exact >= total - int(_near_ties(f).sum()) - 1, (exact, total)
    P.record_exact("large_v3_600_windows", exact, total)
    del engine
    torch.cuda.empty_cache()


@pytest.fixture(scope="module")
def wide():
    """large-v3 width (d 1280, 20 heads, 128 mel bins, V 51 866) with two layers per stack, three
    windows; the oracle runs live."""
    from speech_to_text_amd.engine import WhisperEngine
    import dataclasses
    dims = dataclasses.replace(om.Dims.named("large-v3"), n_audio_layer=2, n_text_layer=2)
    w = P.bf16_round(om.random_weights(dims, seed=7, style="peaked"))
    engine = WhisperEngine(P.dims_dict(dims), w, max_batch=3, max_beams=5, max_text_len=40, max_prompt_len=16)
    oracle = om.Whisper(dims, w)
    g = torch.Generator().manual_seed(8)
    mel = torch.rand(3, 3000, 128, generator=g) * 2 - 1
    with torch.no_grad():
        enc = oracle.encoder(mel)
    return dims, engine, oracle, mel, enc


def test_large_v3_width_encoder_and_logits(wide):
    dims, engine, oracle, mel, enc = wide
    got = engine.encode(mel, return_output=True).cpu().numpy()
    ref = enc.numpy()
    assert P.rel_l2(got, ref) < 1e-2
    assert np.abs(got - ref).max() < 0.05 * ref.std()
    toks = np.random.default_rng(9).integers(0, dims.n_vocab, size=(3, 11)).astype(np.int32)
    logits = engine.decoder_logits(toks).cpu().numpy()
    with torch.no_grad():
        want, _ = oracle.decoder(torch.from_numpy(toks).long(), oracle.cross_kv(enc))
    assert P.rel_l2(logits, want.numpy()) < 2e-2


def test_large_v3_width_greedy(wide):
    dims, engine, oracle, mel, enc = wide
    engine.encode(mel)
    prompt = np.array([[50258, 50259, 50359]] * 3, dtype=np.int32)
    tokens, lengths, scores = engine.generate(prompt, max_new_tokens=24, eos_token=50257, pad_token=50257,
                                              early_stopping=False)
    with torch.no_grad():
        ref = om.greedy(oracle, enc=enc, prompt=torch.from_numpy(prompt).long(), max_new_tokens=24,
                        eos=50257, pad=50257, early_stopping=False)
        exact, total = P.check_greedy_steps(oracle, enc, prompt, tokens, 50257, 50257, scores=scores)
    assert total == 72
    P.record_exact("large_v3_width_greedy", exact, total)
    assert len(set(tokens.reshape(-1).tolist())) > 8          # not a degenerate repeat
    if exact == total:
        assert np.array_equal(tokens, ref["tokens"].numpy())


def test_large_v3_width_beam_5_with_timestamps(wide):
    """BASELINE configs[4]: beam search with 5 beams and the Whisper timestamp rules together, at
    the large-v3 width and vocabulary (timestamps from id 50 364, as the reference's tokenizer
    has them): every transition of every window against the oracle."""
    from speech_to_text_amd.engine import TimestampFilter
    dims, engine, oracle, mel, enc = wide
    engine.encode(mel)
    prompt = np.array([[50258, 50259, 50359]] * 3, dtype=np.int32)
    remove = [1, 2, 7, 8, 9, 10, 14, 25, 50258, 50360, 50361, 50362, 50363]
    first = [220, 50257] + remove
    filt = TimestampFilter(remove_tokens=remove, remove_tokens_with_space=first,
                           timestamp_begin_idx=50364, eos_token_idx=50257)
    cfg = om.FilterConfig(remove_tokens=tuple(remove), remove_tokens_first=tuple(first),
                          timestamp_begin=50364, eos=50257)
    with torch.no_grad():
        n = P.check_beam_steps(engine, oracle, enc, prompt, 5, 10, 50257, 50257, engine_filter=filt,
                               oracle_filter=cfg)
    assert n == 30
    tokens, _, _ = engine.generate(prompt, method="beam", num_beams=5, num_sentences=1, max_new_tokens=10,
                                   eos_token=50257, pad_token=50257, early_stopping=False, logits_filter=filt)
    assert (tokens[:, 0, 0] == 50364).all()                   # the only id the first step allows
    for row in tokens[:, 0]:
        ts = row >= 50364
        assert not (ts[:-2] & ts[1:-1] & ts[2:]).any()        # never three timestamps in a row
