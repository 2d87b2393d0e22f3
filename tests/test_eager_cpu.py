"""The Python generation loop (speech_to_text_amd/eager.py) on the CPU: its decoder step is served by the
torch oracle instead of the HIP engine, so what is checked here is the loop's own logic - selection,
bookkeeping, beam gathers, accumulated attention weights, returned state, continuation - against the
oracle's restatement of architectures/generation_utils.py (greedy :171-300, beam :302-481) on the same
model.  Same arithmetic on both sides, so ids are equal and numbers agree to float32 rounding."""
import numpy as np
import pytest
import torch

from oracle import model as om
from speech_to_text_amd import eager

EOS, PAD = 299, 298


class OracleEngine(object):
    """The surface of WhisperEngine that eager.generate uses, backed by oracle/model.py."""

    def __init__(self, oracle, enc):
        self.oracle, self.enc = oracle, enc
        self.device = torch.device("cpu")
        d = oracle.dims
        self.dims = {"n_vocab": d.n_vocab, "n_text_layer": d.n_text_layer, "n_text_head": d.n_text_head}
        self._epoch = 0
        self.cache, self.length, self.n_seq, self.per = None, 0, 0, 1
        self.cross = None

    def decode_step(self, tokens, seq_per_window=1, reset=False, parents=None, attention=None):
        tokens = torch.as_tensor(np.asarray(tokens) if not isinstance(tokens, torch.Tensor) else tokens).long()
        if tokens.dim() == 1:
            tokens = tokens[:, None]
        self._epoch += 1
        if reset:
            self.cache, self.length = None, 0
            self.cross = [(k.repeat_interleave(seq_per_window, 0), v.repeat_interleave(seq_per_window, 0))
                          for k, v in self.oracle.cross_kv(self.enc)]
        elif parents is not None:
            p = torch.as_tensor(parents).long()
            self.cache = [(k[p], v[p]) for k, v in self.cache]
        with torch.no_grad():
            out = self.oracle.decoder(tokens, self.cross, self.cache, self.length, return_attention=attention is not None)
        self.cache = out[1]
        self.length += tokens.shape[1]
        self.n_seq, self.per = tokens.shape[0], seq_per_window
        logits = out[0][:, -1].contiguous()
        if attention is None:
            return logits
        layers = self.dims["n_text_layer"]
        first = {"all": 0, "last": layers - 1}[attention]
        own = torch.stack([out[2]["attn_layer_%d" % i] for i in range(first, layers)])
        enc = torch.stack([out[2]["enc_attn_layer_%d" % i] for i in range(first, layers)])
        return logits, own, enc

    def step_state(self):
        return self.n_seq, self.per, self.length

    def export_state(self, layer):
        k, v = self.cache[layer]
        return (k.clone(), v.clone()), tuple(t.clone() for t in self.cross[layer])


@pytest.fixture(scope="module")
def setup():
    dims = om.Dims(n_mels=80, n_audio_ctx=100, n_audio_state=64, n_audio_head=2, n_audio_layer=2,
                   n_vocab=300, n_text_ctx=40, n_text_state=64, n_text_head=2, n_text_layer=2)
    oracle = om.Whisper(dims, om.random_weights(dims, seed=7, style="peaked"))
    mel = torch.randn(3, 200, 80, generator=torch.Generator().manual_seed(2))
    with torch.no_grad():
        enc = oracle.encoder(mel)
    prompt = np.array([[1, 2, 3]] * 3, dtype=np.int32)
    return oracle, enc, prompt


def test_greedy_loop_equals_oracle(setup):
    oracle, enc, prompt = setup
    with torch.no_grad():
        want = om.greedy(oracle, enc=enc, prompt=torch.from_numpy(prompt).long(), max_new_tokens=10, eos=EOS, pad=PAD,
                         early_stopping=False, logits_filter=[5, 7, 11], length_temperature=0.3)
    toks, lens, scores = eager.generate(OracleEngine(oracle, enc), prompt, max_new_tokens=10, eos_token=EOS,
                                        pad_token=PAD, early_stopping=False, logits_filter=[5, 7, 11],
                                        length_temperature=0.3)
    assert np.array_equal(toks, want["tokens"].numpy()) and np.array_equal(lens, want["lengths"].numpy())
    assert np.allclose(scores, want["scores"].numpy(), atol=1e-4)


def test_greedy_early_stop_on_an_eos_the_model_emits(setup):
    oracle, enc, prompt = setup
    free = eager.generate(OracleEngine(oracle, enc), prompt, max_new_tokens=8, eos_token=EOS, pad_token=PAD,
                          early_stopping=False)[0]
    eos = int(free[0, 3])                     # window 0 ends after three tokens
    with torch.no_grad():
        want = om.greedy(oracle, enc=enc, prompt=torch.from_numpy(prompt).long(), max_new_tokens=8, eos=eos, pad=0)
    toks, lens, scores = eager.generate(OracleEngine(oracle, enc), prompt, max_new_tokens=8, eos_token=eos,
                                        pad_token=0, check_every=1)
    assert np.array_equal(toks, want["tokens"].numpy()) and np.array_equal(lens, want["lengths"].numpy())
    assert lens[0] == 3 and (toks[0, 4:] == 0).all()


@pytest.mark.parametrize("length_power", [None, 0.7])
def test_beam_loop_equals_oracle(setup, length_power):
    oracle, enc, prompt = setup
    free = eager.generate(OracleEngine(oracle, enc), prompt, method="beam", num_beams=4, num_sentences=1,
                          max_new_tokens=6, eos_token=EOS, pad_token=PAD, early_stopping=False)[0]
    eos = int(free[1, 0, 2])                  # an EOS beams do reach, so that lengths differ
    kw = dict(num_beams=4, num_sentences=3, max_new_tokens=9, length_power=length_power)
    with torch.no_grad():
        want = om.beam_search(oracle, enc=enc, prompt=torch.from_numpy(prompt).long(), eos=eos, pad=0,
                              early_stopping=False, **kw)
    toks, lens, scores = eager.generate(OracleEngine(oracle, enc), prompt, method="beam", eos_token=eos, pad_token=0,
                                        early_stopping=False, **kw)
    assert toks.shape == (3, 3, 9)
    assert np.array_equal(toks, want["tokens"].numpy()) and np.array_equal(lens, want["lengths"].numpy())
    assert np.allclose(scores, want["scores"].numpy(), atol=1e-4)


def test_callable_filter_gets_the_reference_arguments(setup):
    """generation_utils.py:518-520: f(scores, tokens = generated, state = ..., **kwargs); state.state is None
    on the first step only (models/stt/whisper.py:355, 415)."""
    oracle, enc, prompt = setup
    seen = []

    def only_even(scores, tokens, state, bonus=None, **_):
        seen.append((int(state.t), state.state is None, tuple(tokens.shape), bonus))
        return scores.masked_fill((torch.arange(scores.shape[-1]) % 2 == 1)[None], float("-inf"))

    toks = eager.generate(OracleEngine(oracle, enc), prompt, max_new_tokens=4, eos_token=EOS, pad_token=PAD,
                          early_stopping=False, logits_filter=only_even, filter_kwargs={"bonus": 3})[0]
    assert (toks % 2 == 0).all()
    assert seen == [(t, t == 0, (3, 4), 3) for t in range(4)]


def test_accumulated_attention_is_the_teacher_forced_pass(setup):
    """generation_utils.py:576-610: rows of every step appended, self-attention rows growing by one
    position; the result over prompt + generated[:-1] equals one pass of the decoder over those tokens."""
    oracle, enc, prompt = setup
    extras = {}
    toks = eager.generate(OracleEngine(oracle, enc), prompt, max_new_tokens=5, eos_token=EOS, pad_token=PAD,
                          early_stopping=False, return_attention=True, extras=extras)[0]
    fed = torch.from_numpy(np.concatenate([prompt, toks[:, :-1]], axis=1)).long()
    with torch.no_grad():
        _, _, want = oracle.decoder(fed, oracle.cross_kv(enc), return_attention=True)
    got = extras["attention_weights"]
    assert sorted(got) == sorted(want)
    for key in want:
        assert got[key].shape == want[key].shape
        assert torch.allclose(got[key], want[key], atol=1e-5)
    extras = {}
    eager.generate(OracleEngine(oracle, enc), prompt, max_new_tokens=3, eos_token=EOS, pad_token=PAD,
                   early_stopping=False, return_last_attention=True, extras=extras)
    assert sorted(extras["attention_weights"]) == ["attn_layer_1", "enc_attn_layer_1"]


def test_beam_attention_and_state_follow_the_beam_index(setup):
    oracle, enc, prompt = setup
    extras = {}
    toks = eager.generate(OracleEngine(oracle, enc), prompt, method="beam", num_beams=3, num_sentences=3,
                          max_new_tokens=5, eos_token=EOS, pad_token=PAD, early_stopping=False,
                          return_attention=True, return_state=True, extras=extras)[0]
    fed = torch.from_numpy(np.concatenate([np.repeat(prompt, 3, axis=0), toks.reshape(9, 5)[:, :-1]], axis=1)).long()
    cross = [(k.repeat_interleave(3, 0), v.repeat_interleave(3, 0)) for k, v in oracle.cross_kv(enc)]
    with torch.no_grad():
        _, cache, want = oracle.decoder(fed, cross, return_attention=True)
    for key in want:
        assert torch.allclose(extras["attention_weights"][key], want[key], atol=1e-5)
    state = extras["state"]
    assert len(state) == 2 and list(state) == ["layer_0", "layer_1"] and state.length == 7
    (k, v), (ek, ev) = state["layer_1"]
    assert torch.allclose(k, cache[1][0], atol=1e-5) and torch.allclose(v, cache[1][1], atol=1e-5)
    assert torch.equal(ek, cross[1][0])


def test_continuation_from_a_returned_state(setup):
    oracle, enc, prompt = setup
    engine = OracleEngine(oracle, enc)
    whole = eager.generate(OracleEngine(oracle, enc), prompt, max_new_tokens=9, eos_token=EOS, pad_token=PAD,
                           early_stopping=False)
    extras = {}
    first = eager.generate(engine, prompt, max_new_tokens=5, eos_token=EOS, pad_token=PAD, early_stopping=False,
                           return_state=True, extras=extras)
    state = extras["state"]
    assert state.valid() and state.length == 7
    rest = eager.generate(engine, first[0][:, -1:], max_new_tokens=4, eos_token=EOS, pad_token=PAD,
                          early_stopping=False, initial_state=state)
    assert np.array_equal(np.concatenate([first[0], rest[0]], axis=1), whole[0])
    assert np.allclose(first[2] + rest[2], whole[2], atol=1e-4)
    assert np.array_equal(rest[1], whole[1] - first[1])
    assert not state.valid()
    with pytest.raises(ValueError):
        eager.generate(engine, first[0][:, -1:], max_new_tokens=2, eos_token=EOS, pad_token=PAD, initial_state=state)
    with pytest.raises(RuntimeError):
        state["layer_0"]
    with pytest.raises(ValueError):           # a state continues greedy / sampling loops only
        fresh = {}
        eager.generate(engine, prompt, max_new_tokens=2, eos_token=EOS, pad_token=PAD, return_state=True, extras=fresh)
        eager.generate(engine, prompt[:, :1], method="beam", num_beams=2, max_new_tokens=2, eos_token=EOS,
                       pad_token=PAD, initial_state=fresh["state"])
