"""The HIP encoder / decoder / generation path against the torch float32 oracle run on the
same bf16-rounded weights (seeded synthetic weights: the reference ships none)."""
import numpy as np
import pytest
import torch

from oracle import model as om
from tests import _parity as P
from tests._parity import bf16_round as _bf16_round, dims_dict as _dims_dict, rel_l2 as _rel_l2

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def small():
    from speech_to_text_amd.engine import WhisperEngine
    dims = om.Dims(n_mels=80, n_audio_ctx=1500, n_audio_state=128, n_audio_head=2, n_audio_layer=2,
                   n_vocab=1000, n_text_ctx=64, n_text_state=128, n_text_head=2, n_text_layer=2)
    # "peaked" weights: the generated ids depend on the audio and the history (with the plain
    # ones every decode repeats the last prompt token, which tests nothing)
    w = _bf16_round(om.random_weights(dims, seed=0, style="peaked"))
    engine = WhisperEngine(_dims_dict(dims), w, max_batch=3, max_beams=5, max_text_len=48, max_prompt_len=8)
    oracle = om.Whisper(dims, w)
    g = torch.Generator().manual_seed(1)
    mel = (torch.rand(3, 3000, 80, generator=g) * 2 - 1)
    return dims, engine, oracle, mel


def test_encoder_states(small):
    dims, engine, oracle, mel = small
    got = engine.encode(mel, return_output=True).cpu().numpy()
    with torch.no_grad():
        ref = oracle.encoder(mel).numpy()
    assert got.shape == ref.shape == (3, 1500, 128)
    # stated bf16 tolerance: relative L2 <= 1e-2 and max abs <= 0.05 * std(ref)
    assert _rel_l2(got, ref) < 1e-2
    assert np.abs(got - ref).max() < 0.05 * ref.std()


def test_encoder_attention_reference_moves(small, monkeypatch):
    """The encoder attention keeps a softmax reference per query that it moves only when a tile
    of keys raises the maximum far enough above it (a branch that ordinary inputs take on a few
    tiles only).  With the threshold lowered to 1/16 (STT_MI355X_ATTN_LIFT, read by every encoder
    pass) nearly every tile takes that branch, and with it raised to 100 only the first one does:
    all three settings must meet the encoder tolerance, and they differ from one another by no more
    than each differs from the oracle (a moved reference rescales the bf16-rounded weights by a
    factor that is not a power of two, so the roundings differ, nothing else)."""
    dims, engine, oracle, mel = small
    with torch.no_grad():
        ref = oracle.encoder(mel).numpy()
    outs = []
    for lift in (None, "0.0625", "100"):
        if lift is None:
            monkeypatch.delenv("STT_MI355X_ATTN_LIFT", raising=False)
        else:
            monkeypatch.setenv("STT_MI355X_ATTN_LIFT", lift)
        got = engine.encode(mel, return_output=True).cpu().numpy()
        assert _rel_l2(got, ref) < 1e-2, lift
        outs.append(got)
    assert _rel_l2(outs[1], outs[0]) < 5e-3 and _rel_l2(outs[2], outs[0]) < 5e-3
    assert not np.array_equal(outs[1], outs[0])       # the hook did change the path taken


def test_short_window_is_zero_padded(small):
    dims, engine, oracle, mel = small
    short = mel[:1, :1234]
    got = engine.encode(short, return_output=True).cpu().numpy()
    padded = torch.zeros(1, 3000, 80)
    padded[:, :1234] = short
    with torch.no_grad():
        ref = oracle.encoder(padded).numpy()
    assert _rel_l2(got, ref) < 1e-2


def test_decoder_logits_teacher_forced(small):
    dims, engine, oracle, mel = small
    engine.encode(mel)
    tokens = np.random.default_rng(0).integers(0, 1000, size=(3, 7)).astype(np.int32)
    got = engine.decoder_logits(tokens).cpu().numpy()
    with torch.no_grad():
        enc = oracle.encoder(mel)
        ref, _ = oracle.decoder(torch.from_numpy(tokens).long(), oracle.cross_kv(enc))
    ref = ref.numpy()
    assert got.shape == ref.shape
    assert _rel_l2(got, ref) < 2e-2


@pytest.mark.parametrize("n_tokens", [1, 10, 11, 16, 17, 22, 43, 44])
def test_decoder_logits_row_counts(small, n_tokens):
    """3 sequences x n_tokens rows: 3 and 30 rows (the 16-feature kernel, whose last size is 32 rows),
    33, 48, 51, 66, 129 and 132 rows (the 64-feature ring kernel with one, two and three row tiles per
    wave, rows split over several blocks, ragged last tiles): every decode-GEMM variant against the
    oracle."""
    dims, _, oracle, mel = small
    engine = _long_prompt_engine(dims)
    engine.encode(mel)
    tokens = np.random.default_rng(n_tokens).integers(0, 1000, size=(3, n_tokens)).astype(np.int32)
    got = engine.decoder_logits(tokens).cpu().numpy()
    with torch.no_grad():
        enc = oracle.encoder(mel)
        ref, _ = oracle.decoder(torch.from_numpy(tokens).long(), oracle.cross_kv(enc))
    ref = ref.numpy()
    assert got.shape == ref.shape
    assert _rel_l2(got, ref) < 2e-2


@pytest.mark.parametrize("n_tokens", [17, 24, 31, 32, 33, 38, 40, 41])
def test_decoder_logits_up_to_640_rows(small, n_tokens):
    """16 sequences x n_tokens rows = 272, 384, 496, 512, 528, 608 and 640 rows: the 64-feature decode
    kernel with several row groups per layer (what a 512- or 600-window token step runs), ragged and
    full; and 656 rows, the first size that goes through the large tiled kernels instead."""
    from speech_to_text_amd.engine import WhisperEngine
    dims, _, oracle, mel = small
    if "wide" not in _LONG:
        w = _bf16_round(om.random_weights(dims, seed=0, style="peaked"))
        _LONG["wide"] = WhisperEngine(_dims_dict(dims), w, max_batch=16, max_beams=1, max_text_len=48,
                                      max_prompt_len=41)
    engine = _LONG["wide"]
    mel16 = torch.cat([mel, mel.flip(0), mel.flip(1), mel * 0.5, mel.flip(2), mel[:1]], dim=0)
    engine.encode(mel16)
    tokens = np.random.default_rng(n_tokens).integers(0, 1000, size=(16, n_tokens)).astype(np.int32)
    got = engine.decoder_logits(tokens).cpu().numpy()
    with torch.no_grad():
        ref, _ = oracle.decoder(torch.from_numpy(tokens).long(), oracle.cross_kv(oracle.encoder(mel16)))
    assert got.shape == ref.shape and _rel_l2(got, ref.numpy()) < 2e-2


_LONG = {}


def _long_prompt_engine(dims):
    """Same weights as `small`, room for 3 x 48 rows in one decoder pass."""
    if "e" not in _LONG:
        from speech_to_text_amd.engine import WhisperEngine
        w = _bf16_round(om.random_weights(dims, seed=0, style="peaked"))
        _LONG["e"] = WhisperEngine(_dims_dict(dims), w, max_batch=3, max_beams=1, max_text_len=60,
                                   max_prompt_len=48)
    return _LONG["e"]


def test_greedy_tokens(small):
    dims, engine, oracle, mel = small
    engine.encode(mel)
    prompt = np.array([[1, 2, 3]] * 3, dtype=np.int32)
    tokens, lengths, scores = engine.generate(prompt, max_new_tokens=20, eos_token=999, pad_token=999,
                                              early_stopping=False)
    assert tokens.shape == (3, 20)
    with torch.no_grad():
        enc = oracle.encoder(mel)
        ref = om.greedy(oracle, enc=enc, prompt=torch.from_numpy(prompt).long(), max_new_tokens=20,
                        eos=999, pad=999, early_stopping=False)
        # every step: the oracle's argmax, or within the near-tie allowance of it (asserted inside)
        exact, total = P.check_greedy_steps(oracle, enc, prompt, tokens, 999, 999, scores=scores)
    assert total == 60
    P.record_exact("small_greedy_tokens", exact, total)
    assert len(set(tokens.reshape(-1).tolist())) > 6      # the decode is not a degenerate repeat
    if exact == total:                                    # no near-tie met: the free-running decodes agree
        assert np.array_equal(tokens, ref["tokens"].numpy())
        assert np.array_equal(lengths, ref["lengths"].numpy())


def test_greedy_eos_and_suppress_list(small):
    dims, engine, oracle, mel = small
    engine.encode(mel[:2])
    prompt = np.array([[5, 6], [5, 6]], dtype=np.int32)
    free, _, _ = engine.generate(prompt, max_new_tokens=8, eos_token=999, pad_token=999, early_stopping=False)
    eos = int(free[0, 3])                       # make the 4th token of row 0 the EOS
    tokens, lengths, scores = engine.generate(prompt, max_new_tokens=8, eos_token=eos, pad_token=0)
    first = int(np.argmax(free[0] == eos))
    assert lengths[0] == first and (tokens[0, first + 1:] == 0).all() and tokens[0, first] == eos
    with torch.no_grad():
        enc = oracle.encoder(mel[:2])
        P.check_greedy_steps(oracle, enc, prompt, tokens, eos, 0, scores=scores)
    banned = [int(free[1, 0])]
    t2, _, _ = engine.generate(prompt, max_new_tokens=4, eos_token=999, pad_token=999,
                               early_stopping=False, logits_filter=banned)
    assert banned[0] not in t2[1].tolist()


def test_length_temperature(small):
    """`length_temperature` (generation_utils.py:505-507) multiplies the logits by
    (length + 1) ** t before the log-softmax: the greedy ids stay, the scores change."""
    dims, engine, oracle, mel = small
    engine.encode(mel)
    prompt = np.array([[1, 2, 3]] * 3, dtype=np.int32)
    kw = dict(max_new_tokens=12, eos_token=999, pad_token=999, early_stopping=False)
    plain, _, s0 = engine.generate(prompt, **kw)
    tokens, _, s1 = engine.generate(prompt, length_temperature=-0.5, **kw)
    assert np.abs(s0 - s1).max() > 0.5
    with torch.no_grad():
        enc = oracle.encoder(mel)
        P.check_greedy_steps(oracle, enc, prompt, tokens, 999, 999, length_temperature=-0.5, scores=s1)
        n = P.check_beam_steps(engine, oracle, enc, prompt, 3, 5, 999, 999, length_temperature=0.3)
    assert n == 15


def test_beam_search(small):
    """Every transition of every window against the oracle (tests/_parity.py), then the
    properties of the returned sentences."""
    dims, engine, oracle, mel = small
    engine.encode(mel)
    prompt = np.array([[1, 2, 3]] * 3, dtype=np.int32)
    with torch.no_grad():
        enc = oracle.encoder(mel)
        assert P.check_beam_steps(engine, oracle, enc, prompt, 4, 10, 999, 999) == 30
    tokens, lengths, scores = engine.generate(prompt, method="beam", num_beams=4, num_sentences=2,
                                              max_new_tokens=10, eos_token=999, pad_token=999,
                                              early_stopping=False)
    assert tokens.shape == (3, 2, 10)
    assert (scores[:, 0] >= scores[:, 1] - 1e-4).all()
    # the score of each returned sequence equals the oracle's teacher-forced log-probability
    with torch.no_grad():
        g = om.greedy(oracle, enc=enc, prompt=torch.from_numpy(prompt).long(), max_new_tokens=10,
                      eos=999, pad=999, early_stopping=False)
        for b in range(3):
            seq = torch.from_numpy(tokens[b, 0]).long()[None]
            full = torch.cat([torch.from_numpy(prompt[b:b + 1]).long(), seq], dim=1)
            logits, _ = oracle.decoder(full[:, :-1], oracle.cross_kv(enc[b:b + 1]))
            lp = torch.log_softmax(logits[0, 2:], dim=-1)
            want = float(lp.gather(1, seq[0][:, None]).sum())
            assert abs(want - scores[b, 0]) < 0.3


def test_beam_search_ten_beams(small):
    """More beams than a thread keeps in registers during the one-pass selection (8): the
    pass-per-winner selection takes over; every transition against the oracle, an EOS in reach."""
    dims, engine, oracle, mel = small
    engine.reserve(max_beams=10)
    engine.encode(mel)
    prompt = np.array([[1, 2, 3]] * 3, dtype=np.int32)
    free, _, _ = engine.generate(prompt, method="beam", num_beams=10, num_sentences=1, max_new_tokens=6,
                                 eos_token=999, pad_token=999, early_stopping=False)
    eos = int(free[0, 0, 2])
    with torch.no_grad():
        enc = oracle.encoder(mel)
        assert P.check_beam_steps(engine, oracle, enc, prompt, 10, 6, eos, 0) == 18
        assert P.check_beam_steps(engine, oracle, enc, prompt, 8, 6, eos, 0) == 18


def test_beam_search_length_power(small):
    """`length_power` (generation_utils.py:372-374): candidates ranked by score / (length + 1) ** p,
    with an EOS that beams do reach so that lengths differ between beams."""
    dims, engine, oracle, mel = small
    engine.encode(mel)
    prompt = np.array([[1, 2, 3]] * 3, dtype=np.int32)
    free, _, _ = engine.generate(prompt, method="beam", num_beams=4, num_sentences=1, max_new_tokens=8,
                                 eos_token=999, pad_token=999, early_stopping=False)
    eos = int(free[0, 0, 2])
    with torch.no_grad():
        enc = oracle.encoder(mel)
        assert P.check_beam_steps(engine, oracle, enc, prompt, 4, 8, eos, 0, length_power=0.7) == 24
        assert P.check_beam_steps(engine, oracle, enc, prompt, 4, 8, eos, 0, length_power=2.0) == 24


def test_timestamp_filter(small):
    from speech_to_text_amd.engine import TimestampFilter
    dims, engine, oracle, mel = small
    engine.encode(mel)
    prompt = np.array([[1, 2, 3]] * 3, dtype=np.int32)
    filt = TimestampFilter(remove_tokens=[10, 11, 12], remove_tokens_with_space=[10, 11, 12, 13, 900],
                           timestamp_begin_idx=920, eos_token_idx=900)
    tokens, lengths, scores = engine.generate(prompt, max_new_tokens=12, eos_token=900, pad_token=900,
                                              early_stopping=False, logits_filter=filt)
    assert (tokens[:, 0] == 920).all()
    cfg = om.FilterConfig(remove_tokens=(10, 11, 12), remove_tokens_first=(10, 11, 12, 13, 900),
                          timestamp_begin=920, eos=900)
    with torch.no_grad():
        enc = oracle.encoder(mel)
        P.check_greedy_steps(oracle, enc, prompt, tokens, 900, 900, logits_filter=cfg, scores=scores)
    for row in tokens:
        ts = row >= 920
        assert not (ts[:-2] & ts[1:-1] & ts[2:]).any()      # never three timestamps in a row


def test_return_logits(small):
    """`return_logits` (generation_utils.py:266-268, 297): the processed scores of every greedy step,
    with the timestamp rules on so that the filter's -inf entries are part of the comparison; and
    the same tokens as the call without it (which replays a recorded graph instead)."""
    from speech_to_text_amd.engine import TimestampFilter
    dims, engine, oracle, mel = small
    engine.encode(mel)
    prompt = np.array([[1, 2, 3]] * 3, dtype=np.int32)
    filt = TimestampFilter(remove_tokens=[10, 11, 12], remove_tokens_with_space=[10, 11, 12, 13, 900],
                           timestamp_begin_idx=920, eos_token_idx=900)
    kw = dict(max_new_tokens=10, eos_token=900, pad_token=900, early_stopping=False, logits_filter=filt)
    tokens, lengths, scores, logits = engine.generate(prompt, return_logits=True, **kw)
    assert logits.shape == (3, 10, dims.n_vocab) and logits.is_cuda
    plain = engine.generate(prompt, **kw)
    assert np.array_equal(plain[0], tokens) and np.allclose(plain[2], scores, atol=1e-5)
    cfg = om.FilterConfig(remove_tokens=(10, 11, 12), remove_tokens_first=(10, 11, 12, 13, 900),
                          timestamp_begin=920, eos=900)
    with torch.no_grad():
        enc = oracle.encoder(mel)
        P.check_greedy_steps(oracle, enc, prompt, tokens, 900, 900, logits_filter=cfg, scores=scores,
                             engine_logits=logits.cpu().numpy())
    picked = np.take_along_axis(logits.cpu().numpy(), tokens[:, :, None], axis=2)[:, :, 0]
    assert np.allclose(picked.sum(axis=1), scores, atol=1e-3)           # scores = sum of the chosen entries


def test_beam_search_with_timestamp_filter(small):
    """Beam search and the Whisper timestamp rules together (whisper.py:354-416 inside
    generation_utils.py:333-438), 5 beams: every transition against the oracle."""
    from speech_to_text_amd.engine import TimestampFilter
    dims, engine, oracle, mel = small
    engine.encode(mel)
    prompt = np.array([[1, 2, 3]] * 3, dtype=np.int32)
    filt = TimestampFilter(remove_tokens=[10, 11, 12], remove_tokens_with_space=[10, 11, 12, 13, 900],
                           timestamp_begin_idx=920, eos_token_idx=900)
    cfg = om.FilterConfig(remove_tokens=(10, 11, 12), remove_tokens_first=(10, 11, 12, 13, 900),
                          timestamp_begin=920, eos=900)
    with torch.no_grad():
        enc = oracle.encoder(mel)
        n = P.check_beam_steps(engine, oracle, enc, prompt, 5, 12, 900, 900, engine_filter=filt,
                               oracle_filter=cfg)
    assert n == 36


def test_encoder_states_width_256():
    """Width 256: every encoder GEMM has N % 256 == 0 and M = 4500 rows (not a multiple of the
    256-row tile), so the 256 x 256 tile kernel and its row clamping carry the whole pass."""
    from speech_to_text_amd.engine import WhisperEngine
    dims = om.Dims(n_mels=80, n_audio_ctx=1500, n_audio_state=256, n_audio_head=4, n_audio_layer=2,
                   n_vocab=1000, n_text_ctx=64, n_text_state=256, n_text_head=4, n_text_layer=1)
    w = _bf16_round(om.random_weights(dims, seed=3, style="peaked"))
    engine = WhisperEngine(_dims_dict(dims), w, max_batch=3, max_beams=1, max_text_len=16, max_prompt_len=4)
    oracle = om.Whisper(dims, w)
    g = torch.Generator().manual_seed(4)
    mel = (torch.rand(3, 3000, 80, generator=g) * 2 - 1)
    got = engine.encode(mel, return_output=True).cpu().numpy()
    with torch.no_grad():
        enc = oracle.encoder(mel)
    ref = enc.numpy()
    assert _rel_l2(got, ref) < 1e-2
    assert np.abs(got - ref).max() < 0.05 * ref.std()
    # and the decoder on top of it
    prompt = np.array([[1, 2, 3]] * 3, dtype=np.int32)
    toks, lens, scores = engine.generate(prompt, max_new_tokens=8, eos_token=999, pad_token=999,
                                         early_stopping=False)
    with torch.no_grad():
        P.check_greedy_steps(oracle, enc, prompt, toks, 999, 999, scores=scores)


def test_greedy_sampling(small):
    """temperature given = categorical draws (generation_utils.py:546): with the same counter-based
    noise the GPU token must be the oracle's argmax of log-prob + noise at every step, or lose to
    it by less than the bf16 margin; same seed -> same tokens, other seed -> other tokens."""
    dims, engine, oracle, mel = small
    engine.encode(mel)
    prompt = np.array([[1, 2, 3]] * 3, dtype=np.int32)
    kw = dict(max_new_tokens=16, eos_token=999, pad_token=999, early_stopping=False, temperature=0.7)
    tokens, lengths, scores = engine.generate(prompt, seed=1234, **kw)
    again, _, _ = engine.generate(prompt, seed=1234, **kw)
    other, _, _ = engine.generate(prompt, seed=99, **kw)
    plain, _, _ = engine.generate(prompt, **{**kw, "temperature": None})
    assert (tokens == again).all()
    assert (tokens != other).any() and (tokens != plain).any()
    with torch.no_grad():
        enc = oracle.encoder(mel)
        cross = oracle.cross_kv(enc)
        cache, offset, inputs = None, 0, torch.from_numpy(prompt).long()
        lengths_t = torch.full((3,), 3)
        gen = torch.from_numpy(tokens).long()
        total = exact = 0
        want_score = np.zeros(3)
        for t in range(16):
            logits, cache = oracle.decoder(inputs, cross, cache, offset)
            offset += inputs.shape[1]
            lp = om.process_logits(logits[:, -1], lengths_t, gen, t, temperature=0.7)
            for i in range(3):
                key = lp[i] + om.gumbel_noise(1234, t * 3 + i, lp.shape[1])
                best = int(key.argmax())
                total += 1
                exact += best == tokens[i, t]
                assert float(key[best] - key[tokens[i, t]]) < P.row_margin(logits[i, -1]), (i, t)
                want_score[i] += float(lp[i, tokens[i, t]])
            lengths_t = lengths_t + 1
            inputs = gen[:, t][:, None]
    assert exact >= total - 3
    assert np.abs(want_score - scores).max() < 0.02 * 3.5 * 16


def test_beam_sampling(small):
    """Beam search with a temperature: the candidates of a step are categorical draws with
    replacement over the flattened beam scores (generation_utils.py:379-384, 546).  Checked for
    the first two steps against the oracle's scores plus the same noise."""
    dims, engine, oracle, mel = small
    engine.encode(mel)
    nb, v = 4, dims.n_vocab
    prompt = np.array([[1, 2, 3]] * 3, dtype=np.int32)
    kw = dict(method="beam", num_beams=nb, num_sentences=nb, eos_token=999, pad_token=999,
              early_stopping=False, temperature=1.3, seed=77)
    one, _, s1 = engine.generate(prompt, max_new_tokens=1, **kw)
    two, _, s2 = engine.generate(prompt, max_new_tokens=2, **kw)
    assert one.shape == (3, nb, 1) and two.shape == (3, nb, 2)
    with torch.no_grad():
        enc = oracle.encoder(mel)
        p = torch.from_numpy(prompt).long()
        logits, _ = oracle.decoder(p, oracle.cross_kv(enc))
        lp0 = torch.log_softmax(logits[:, -1] / 1.3, dim=-1)
        for b in range(3):
            for d in range(nb):
                key = lp0[b] + om.gumbel_noise(77, 0 * 3 * nb + b * nb + d, v)
                assert float(key.max() - key[one[b, d, 0]]) < P.row_margin(logits[b, -1])
                assert abs(float(lp0[b, one[b, d, 0]]) - s1[b, d]) < P.row_margin(logits[b, -1])
        # second step: the parents are the first step's draws (same seed, same streams)
        for b in range(3):
            full = torch.cat([p[b:b + 1].repeat(nb, 1), torch.from_numpy(one[b]).long()], dim=1)
            logits, _ = oracle.decoder(full, [(k[b:b + 1].repeat(nb, 1, 1, 1), vv[b:b + 1].repeat(nb, 1, 1, 1))
                                              for k, vv in oracle.cross_kv(enc)])
            lp1 = torch.log_softmax(logits[:, -1] / 1.3, dim=-1)
            lws = (lp0[b, one[b, :, 0]][:, None] + lp1).reshape(-1)
            for d in range(nb):
                key = lws + om.gumbel_noise(77, 1 * 3 * nb + b * nb + d, nb * v)
                t0, t1 = int(two[b, d, 0]), int(two[b, d, 1])
                parents = [q for q in range(nb) if int(one[b, q, 0]) == t0]
                assert parents, "the second-step sequence must extend one of the first-step draws"
                got = max(float(key[q * v + t1]) for q in parents)
                assert float(key.max()) - got < P.row_margin(logits[:, -1])
                assert abs(max(float(lws[q * v + t1]) for q in parents) - s2[b, d]) < 2 * P.row_margin(logits[:, -1])


def test_suppress_list_matches_oracle(small):
    """A `logits_filter` given as a list of token ids (`mask_batch_tokens`, generation_utils.py:
    521-522): with all but a handful of columns masked, every token must be one of the survivors
    and the oracle's argmax under the same mask."""
    dims, engine, oracle, mel = small
    engine.encode(mel)
    prompt = np.array([[1, 2, 3]] * 3, dtype=np.int32)
    survivors = {7, 123, 500, 998}
    banned = [i for i in range(dims.n_vocab) if i not in survivors]
    tokens, _, _ = engine.generate(prompt, max_new_tokens=12, eos_token=999, pad_token=999,
                                   early_stopping=False, logits_filter=banned)
    assert set(tokens.reshape(-1).tolist()) <= survivors
    with torch.no_grad():
        enc = oracle.encoder(mel)
        P.check_greedy_steps(oracle, enc, prompt, tokens, 999, 999, logits_filter=banned)
    plain, _, _ = engine.generate(prompt, method="sample", max_new_tokens=4, eos_token=999, pad_token=999,
                                  early_stopping=False)
    ref, _, _ = engine.generate(prompt, max_new_tokens=4, eos_token=999, pad_token=999, early_stopping=False)
    assert (plain == ref).all()          # 'sample' without a temperature is greedy (:826-831)


def test_beam_search_finished_beams(small):
    """Beams that reach EOS: afterwards only the pad token is appended at no cost
    (generation_utils.py:367-371), lengths exclude the EOS, early stopping ends the loop once the
    first `num_sentences` beams of every window are finished (:326-330)."""
    dims, engine, oracle, mel = small
    engine.encode(mel)
    prompt = np.array([[1, 2, 3]] * 3, dtype=np.int32)
    free, _, _ = engine.generate(prompt, method="beam", num_beams=4, num_sentences=1, max_new_tokens=10,
                                 eos_token=999, pad_token=999, early_stopping=False)
    eos = int(free[0, 0, 3])                    # the 4th token of window 0's best beam becomes the EOS
    kw = dict(num_beams=4, num_sentences=2, max_new_tokens=10)
    tokens, lengths, scores = engine.generate(prompt, method="beam", eos_token=eos, pad_token=0,
                                              early_stopping=True, **kw)
    with torch.no_grad():
        enc = oracle.encoder(mel)
        # every transition of every window, finished beams included (pad appended at no cost,
        # lengths frozen): asserted inside
        assert P.check_beam_steps(engine, oracle, enc, prompt, 4, 10, eos, 0) == 30
    for b in range(3):
        for s in range(2):
            row, n = tokens[b, s], int(lengths[b, s])
            hits = np.where(row == eos)[0]
            if len(hits):                       # finished: EOS right after the counted tokens, then pad
                assert hits[0] == n and (row[n + 1:] == 0).all()
            else:
                assert n <= 10
        assert scores[b, 0] >= scores[b, 1] - 1e-4
    # early stopping (generation_utils.py:326-330): the state returned is the one of the first step
    # after which the first `num_sentences` beams of every window are finished, whatever the
    # engine's host-side polling interval.  One window, EOS = the third token of its best beam.
    engine.encode(mel[:1])
    one = prompt[:1]
    kw = dict(method="beam", num_beams=4, num_sentences=2, eos_token=int(free[0, 0, 2]), pad_token=0)
    stop = None
    for t in range(1, 17):
        ft, fl, fs = engine.generate(one, max_new_tokens=t, early_stopping=False, **kw)
        if (ft == kw["eos_token"]).any(axis=2).all():
            stop = t
            break
    assert stop is not None and 1 < stop < 16, "no step at which the window's two best beams are finished"
    for check_every in (1, 8):
        et, el, es = engine.generate(one, max_new_tokens=16, early_stopping=True, check_every=check_every, **kw)
        assert np.array_equal(et[:, :, :stop], ft) and (et[:, :, stop:] == 0).all()
        assert np.array_equal(el, fl) and np.allclose(es, fs, atol=1e-5)


def test_single_row_path_matches_batched_bits_and_oracle(small):
    """One clip decoded alone takes the one-row path of the decode step (every LayerNorm inside the
    projection that consumes it); the same clip decoded with two others takes the separate
    kernels.  The arithmetic is the same operation for operation: identical tokens and scores, and
    one-token logits within the stated tolerance of the oracle."""
    dims, engine, oracle, mel = small
    prompt = np.array([[1, 2, 3]] * 3, dtype=np.int32)
    kw = dict(max_new_tokens=12, eos_token=999, pad_token=999, early_stopping=False)
    engine.encode(mel)
    t3, _, s3 = engine.generate(prompt, **kw)
    engine.encode(mel[:1])
    t1, _, s1 = engine.generate(prompt[:1], **kw)
    assert np.array_equal(t1[0], t3[0]) and s1[0] == s3[0]
    tok = np.array([[7]], dtype=np.int32)
    got = engine.decoder_logits(tok).cpu().numpy()                   # 1 sequence x 1 token = one row
    with torch.no_grad():
        enc = oracle.encoder(mel[:1])
        ref, _ = oracle.decoder(torch.from_numpy(tok).long(), oracle.cross_kv(enc))
    assert _rel_l2(got, ref.numpy()) < 2e-2


def test_reserve_keeps_weights_and_cached_graphs_replay(small):
    """stt_engine_reserve grows the buffers without touching the weights: same tokens before and
    after; a second call with the same configuration replays the cached token-step graph, a
    different configuration records its own."""
    from speech_to_text_amd.engine import WhisperEngine
    dims, _, oracle, mel = small
    w = _bf16_round(om.random_weights(dims, seed=0, style="peaked"))
    engine = WhisperEngine(_dims_dict(dims), w, max_batch=1, max_beams=1, max_text_len=16, max_prompt_len=4)
    prompt = np.array([[1, 2, 3]] * 3, dtype=np.int32)
    kw = dict(max_new_tokens=10, eos_token=999, pad_token=999, early_stopping=False)
    engine.encode(mel[:1])
    first, _, s_first = engine.generate(prompt[:1], **kw)
    again, _, _ = engine.generate(prompt[:1], **kw)                  # cache hit
    assert np.array_equal(first, again)
    engine.reserve(max_batch=3, max_beams=4, max_text_len=40, max_prompt_len=12)
    with pytest.raises(Exception):
        engine.generate(prompt[:1], **kw)                            # the encoded batch is gone
    engine.encode(mel)
    wide, _, s_wide = engine.generate(prompt, **kw)
    assert np.array_equal(wide[0], first[0]) and abs(s_wide[0] - s_first[0]) < 1e-4
    beams, _, _ = engine.generate(prompt, method="beam", num_beams=4, num_sentences=1, **kw)
    wide2, _, _ = engine.generate(prompt, **kw)                      # greedy graph still valid after the beam one
    assert np.array_equal(wide, wide2) and beams.shape == (3, 1, 10)
    long_prompt = np.tile(np.array([[1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12]], dtype=np.int32), [3, 1])
    out, _, sc = engine.generate(long_prompt, **kw)
    with torch.no_grad():
        P.check_greedy_steps(oracle, oracle.encoder(mel), long_prompt, out, 999, 999, scores=sc)
