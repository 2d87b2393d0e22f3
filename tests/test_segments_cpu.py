"""The wrapper's sequential long-form loop (`Whisper._infer_segments`) against the oracle's
restatement of the reference's (oracle/segments.py) on scripted model outputs: same prompts, same
seeks, same segments.  The runtime is a stub that plays the script back (no GPU)."""
import numpy as np
import pytest

from oracle import segments as osg

TB, EOS = 50364, 50257


def _ts(seconds):
    return TB + int(round(seconds / 0.02))


SCRIPTS = {
    # three closed segments, then text after the last pair that is dropped and decoded again
    "pairs": [[_ts(0), 11, 12, _ts(2.5), _ts(2.5), 13, _ts(7), _ts(7), 14, 15, 16, _ts(29.9), _ts(29.9), 17],
              [_ts(0), 21, _ts(4), _ts(4), 22],
              [_ts(0), 31, 32, _ts(3.3)],
              [40, 41]],
    # no timestamps at all (the reference's default decode: no filter)
    "plain": [[50363, 5, 6, 7], [50363, 8], [9, 10, 11], [12]],
    # a single trailing timestamp: the window lasts until it; and a zero timestamp only
    "single": [[_ts(0), 5, 6, _ts(12.34)], [_ts(0), 7, 8], [1], [2]],
    # ids beyond EOS inside the text (specials) are dropped from `tokens`
    "specials": [[_ts(0), 5, 50300, 6, _ts(1), _ts(1), 50301, 7], [_ts(0), 50360, 8, _ts(9)], [3], [4]],
}


class _ScriptedRuntime(object):
    """Stands where MI355XRuntime stands: called like `compiled_infer`, returns [batch][ids]."""
    dims = {"n_mels": 80}

    def __init__(self, script):
        self.script, self.calls = list(script), []

    def set_tokens(self, **_):
        pass

    def __call__(self, mel, *, tokens, tokens_length, **kwargs):
        assert tuple(mel.shape) == (1, 3000, 80) and kwargs["encoder_output_lengths"] == 1500
        self.calls.append(np.asarray(tokens)[0, :tokens_length].tolist())
        return [self.script[len(self.calls) - 1]]


def _wrapper(script):
    from speech_to_text_amd.stt import whisper as W
    model = W.Whisper.__new__(W.Whisper)          # no engine, no log-mel plan: host logic only
    model.vocabulary = W.WhisperVocabulary()
    model.max_input_length, model.max_output_length, model.pad_mel_value, model.rate = 3000, 256, 0., 16000
    model.mel_fn = type("Mel", (), {"hop_length": 160, "n_mel_channels": 80})()
    model.model = _ScriptedRuntime(script)
    return model


@pytest.mark.parametrize("name", sorted(SCRIPTS))
@pytest.mark.parametrize("conditioned", [True, False])
@pytest.mark.parametrize("lang", ["en", None])
def test_seek_loop_matches_reference_restatement(name, conditioned, lang):
    import torch
    script = SCRIPTS[name]
    if lang is None:                               # the model writes <|lang|><|task|> itself
        script = [[50259, 50359] + s for s in script]
    n_frames = 7321
    model = _wrapper(script)
    mel = torch.zeros(n_frames, 80)
    got = model._infer_segments(mel, lang=lang, condition_on_previous_text=conditioned, verbose=0)
    tok = osg.Tokens(lang_to_idx=model.vocabulary.lang_to_idx)
    seeks = []
    want, calls = osg.infer_segments(n_frames, lambda seek, prompt: (seeks.append(seek), script[len(seeks) - 1])[1],
                                     tok, lang=lang, condition_on_previous_text=conditioned,
                                     decode_text=model.decode_output)
    assert [c[1] for c in calls] == model.model.calls          # same prompts, hence same seeks and history
    assert len(got) == len(want)
    for g, w in zip(got, want):
        assert set(g) == set(w)
        assert g["lang"] == w["lang"] and g["text"] == w["text"]
        assert np.array_equal(g["tokens"], w["tokens"])
        assert g["start"] == w["start"] and g["end"] == w["end"] and g["time"] == w["time"]


def test_pair_closed_at_zero_moves_on():
    """`<|0.00|> text <|0.00|><|0.00|>`: the reference's `seek += 0` never ends; the wrapper moves on by
    one window instead (documented deviation)."""
    import torch
    model = _wrapper([[_ts(0), 5, _ts(0), _ts(0), 6]] * 3)
    segs = model._infer_segments(torch.zeros(6500, 80), lang="en", verbose=0)
    assert len(model.model.calls) == 3 and len(segs) == 3


def test_detect_language_pick():
    from speech_to_text_amd.stt.whisper import LANGUAGES
    z = np.random.default_rng(0).normal(size=51865)
    lang, probs = osg.detect_language(z, range(50259, 50259 + len(LANGUAGES)), LANGUAGES)
    assert lang == LANGUAGES[int(np.argmax(z[50259: 50259 + len(LANGUAGES)]))]
    assert abs(sum(probs.values()) - 1) < 1e-12
