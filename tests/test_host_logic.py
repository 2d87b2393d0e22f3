"""Host-side logic that needs no GPU: audio pre-steps, edit distance, runtime registry,
weight-name normalisation."""
import os

import numpy as np
import pytest
from scipy.io import wavfile

from oracle import frontend as oracle_frontend


def test_audio_presteps_match_oracle(golden_dir):
    from speech_to_text_amd.audio.audio_processing import (WHISPER_TRIM_KWARGS, normalize_audio,
                                                           read_audio, trim_silence_rms)
    rate, pcm = wavfile.read(os.path.join(golden_dir, "audio_test.wav"))
    a = normalize_audio(pcm, max_val=1.)
    assert a.dtype == np.float32 and abs(np.abs(a).max() - 1) < 1e-6
    assert np.array_equal(a, oracle_frontend.normalize_audio(pcm))
    t = trim_silence_rms(a, rate=rate, **WHISPER_TRIM_KWARGS)
    assert np.array_equal(t, oracle_frontend.whisper_prepare_audio(pcm))
    assert len(t) == 63360            # notebook cell-13: 396 frames, end = 3.96 s
    _, via_reader = read_audio(os.path.join(golden_dir, "audio_test.wav"), trim_silence=True,
                               **WHISPER_TRIM_KWARGS)
    assert np.array_equal(via_reader, t)
    for mode in ("start_end", "start", "end"):
        x = trim_silence_rms(a, rate=rate, mode=mode)
        y = oracle_frontend.trim_silence_rms(a, rate=rate, mode=mode)
        assert np.array_equal(x, y)


def test_edit_distance_readme_matrices():
    from speech_to_text_amd.text.metrics import edit_distance
    # README.md:85-108 of the reference
    d, m = edit_distance("cat", "ct", return_matrix=True, normalize=False)
    assert d == 1 and m.tolist() == [[0, 1, 2], [1, 0, 1], [2, 1, 1], [3, 2, 1]]
    d, m = edit_distance("cat", "the ct is", partial=True, return_matrix=True, normalize=False)
    assert d == 1
    assert m.tolist() == [[0] * 10, [1, 1, 1, 1, 1, 0, 1, 1, 1, 1], [2, 2, 2, 2, 2, 1, 1, 2, 2, 2],
                          [3, 2, 3, 3, 3, 2, 1, 2, 3, 3]]
    assert abs(edit_distance("cat", "the ct is", partial=True) - 1 / 3) < 1e-12
    assert edit_distance([1, 2, 3], [1, 2, 3]) == 0
    assert edit_distance("abc", "abd", replacement_cost={"c": {"d": 0.5}}, normalize=False) == 0.5


def test_runtime_registry_and_errors():
    from speech_to_text_amd.runtimes import build_runtime, Runtime, _runtimes
    assert "mi355x" in _runtimes and issubclass(_runtimes["mi355x"], Runtime)
    with pytest.raises(ValueError):
        build_runtime("trt_llm", "/nowhere")


def test_weight_name_normalisation():
    import torch
    from speech_to_text_amd import weights as W
    from speech_to_text_amd.engine import whisper_dims
    dims = dict(whisper_dims("tiny"), n_audio_layer=1, n_text_layer=1, n_vocab=100)
    w = W.random_whisper_weights(dims, 0)
    assert W.dims_from_openai_state_dict(w) == dims
    hf = {}
    back = {"query": "q_proj", "key": "k_proj", "value": "v_proj", "out": "out_proj"}
    for k, v in w.items():
        parts = k.split(".")
        if k.startswith("encoder.conv"):
            hf["model." + k] = v
        elif k.startswith("encoder.ln_post"):
            hf["model.encoder.layer_norm." + parts[-1]] = v
        elif k.startswith("decoder.ln."):
            hf["model.decoder.layer_norm." + parts[-1]] = v
        elif k == "decoder.token_embedding.weight":
            hf["model.decoder.embed_tokens.weight"] = v
            hf["proj_out.weight"] = v
        elif k == "decoder.positional_embedding":
            hf["model.decoder.embed_positions.weight"] = v
        else:
            base = "model.%s.layers.%s." % (parts[0], parts[2])
            mod = parts[3]
            if mod in ("attn", "cross_attn"):
                hf[base + ("self_attn" if mod == "attn" else "encoder_attn") + "." + back[parts[4]] + "." + parts[5]] = v
            elif mod == "attn_ln":
                hf[base + "self_attn_layer_norm." + parts[4]] = v
            elif mod == "cross_attn_ln":
                hf[base + "encoder_attn_layer_norm." + parts[4]] = v
            elif mod == "mlp_ln":
                hf[base + "final_layer_norm." + parts[4]] = v
            else:
                hf[base + ("fc1." if parts[4] == "0" else "fc2.") + parts[5]] = v
    conv = W.from_hf_state_dict(hf)
    assert set(conv) == set(w)
    assert all(torch.equal(conv[k], w[k]) for k in w)
    keras = {k: (v.numpy().T if (k.endswith(".weight") and v.dim() == 2 and "embedding" not in k)
                 else (np.transpose(v.numpy(), (2, 1, 0)) if v.dim() == 3 else v.numpy())) for k, v in w.items()}
    rt = W.from_keras_layout(keras)
    assert all(np.array_equal(rt[k], w[k].numpy()) for k in w)


def test_vocabulary_ids():
    from speech_to_text_amd.stt.whisper import WhisperVocabulary, LANGUAGES
    v = WhisperVocabulary()
    assert len(LANGUAGES) == 99 and v.lang_to_idx["en"] == 50259 and v.lang_to_idx["fr"] == 50265
    assert v.transcribe_token_idx == 50359 and v.timestamp_begin_idx == 50364


def test_logits_masks():
    """The cases of the reference's `test_logits_filtering` (tests/test_utils_text.py:258-279):
    column lists and column ranges set to -inf, on numpy arrays and on torch tensors."""
    import torch
    from speech_to_text_amd.text.tokens_processing import mask_batch_tokens, mask_slice_tokens, mask_tokens
    logits = np.random.default_rng(0).normal(size=(5, 25)).astype(np.float32)
    keep = logits.copy()
    for wrap, unwrap in ((lambda a: a, lambda a: a), (torch.from_numpy, lambda a: a.numpy())):
        for i in range(0, 25, 5):
            want = logits.copy(); want[:, i:] = -np.inf
            assert np.array_equal(unwrap(mask_slice_tokens(wrap(logits), i, remove_after=True)), want)
            want = logits.copy(); want[:, :i] = -np.inf
            assert np.array_equal(unwrap(mask_slice_tokens(wrap(logits), i, remove_after=False)), want)
            indexes = [i, 24, 1]
            want = logits.copy(); want[:, indexes] = -np.inf
            assert np.array_equal(unwrap(mask_batch_tokens(wrap(logits), indexes)), want)
        want = logits.copy(); want[1, 3] = want[4, 0] = -np.inf
        assert np.array_equal(unwrap(mask_tokens(wrap(logits), [[1, 3], [4, 0]])), want)
    assert np.array_equal(logits, keep)


def test_reference_callable_filter_is_translated():
    """The reference passes its timestamp rules as a callable bound to the model
    (models/stt/whisper.py:414-416): recognised by name, translated from the model's attributes;
    unknown callables are refused."""
    import functools
    import types
    from speech_to_text_amd.engine import TimestampFilter

    def logits_filter(self, scores, tokens, state, **_):        # stands for the reference's free function
        raise AssertionError("never called on the host")

    def timestamp_filter(self, scores, tokens, to_remove, state, max_initial_timestamp=1, **_):
        raise AssertionError("never called on the host")

    model = types.SimpleNamespace(remove_tokens=np.array([5, 6]), remove_tokens_with_space=np.array([220, 50257, 5, 6]),
                                  timestamp_begin_idx=50364, eos_token_idx=50257)
    for fn in (functools.partial(logits_filter, model), types.MethodType(logits_filter, model),
               functools.partial(logits_filter, self=model)):
        f = TimestampFilter.from_callable(fn)
        assert f.remove_tokens.tolist() == [5, 6] and f.remove_tokens_with_space.tolist() == [220, 50257, 5, 6]
        assert (f.timestamp_begin_idx, f.eos_token_idx, f.max_initial_timestamp) == (50364, 50257, 1)
    f = TimestampFilter.from_callable(functools.partial(timestamp_filter, model, to_remove=[7], max_initial_timestamp=0))
    assert f.remove_tokens.tolist() == [7] and f.max_initial_timestamp == 0
    assert TimestampFilter.from_callable(lambda scores, **kw: scores) is None
    assert TimestampFilter.from_callable(functools.partial(logits_filter, object())) is None


def test_large_v3_special_token_ids():
    """The large-v3 tokenizer has one more language token than whisper-base's: every id from
    <|translate|> on is one higher and the timestamps start at 50365 (n_vocab 51866)."""
    from speech_to_text_amd.stt.whisper import WhisperVocabulary
    base, v3 = WhisperVocabulary(), WhisperVocabulary(vocab_size=50365)
    assert (base.translate_token_idx, base.transcribe_token_idx, base.timestamp_begin_idx) == (50358, 50359, 50364)
    assert (v3.translate_token_idx, v3.transcribe_token_idx, v3.timestamp_begin_idx) == (50359, 50360, 50365)
    assert v3.lang_to_idx["yue"] == 50358 and len(v3.lang_to_idx) == 100 and v3.lang_to_idx["en"] == 50259
    assert v3.start_of_prev_token_idx == 50362 and v3.notimestamps_idx == 50364
    assert v3.eos_token_idx == base.eos_token_idx == 50257 and v3.sos_token_idx == 50258


def test_stft_configuration_round_trip(tmp_path):
    """The reference's own test of its mel classes (tests/test_utils_audio.py:89-103): a saved
    configuration loads back into an equal object, of the class its `class_name` states."""
    from speech_to_text_amd.audio.stft import MelSTFT, WhisperSTFT
    for fn in (WhisperSTFT(), WhisperSTFT(n_mel_channels=128)):
        path = str(tmp_path / "WhisperSTFT.json")
        fn.save(path)
        assert os.path.exists(path)
        back = MelSTFT.load_from_file(path)
        assert type(back) is WhisperSTFT and back == fn and back.get_config() == fn.get_config()
        assert MelSTFT.create(path) == fn
    assert WhisperSTFT() != WhisperSTFT(n_mel_channels=128)
    # the reference's length estimate, ceil(max(filter_length, n) / hop_length) (utils/audio/stft.py:122-127):
    # an upper bound used for padding, not the frame count of the transform (404 frames for 64 880 samples)
    fn = WhisperSTFT()
    assert [fn.get_mel_length(n) for n in (399, 400, 559, 560, 64880)] == [3, 3, 4, 4, 406]
    assert fn.get_audio_length(406) == 64960
