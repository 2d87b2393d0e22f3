"""A model directory as the reference writes it (pretrained_models/<name>/saving/{*.weights.h5,
tokenizer.json, mel_fn.json}, models/interfaces/base_model.py:126-136) read without Keras: the
converted weights, the reference-format tokenizer file and the mel configuration.  The fixture
under tests/golden/keras_model_dir is synthetic (tools/convert_keras_weights.py --make-synthetic):
no real reference checkpoint is available here, so the variable layout is the one the reference's
layer classes declare, not one read back from a file the reference wrote (parity unpinned)."""
import os
import shutil
import subprocess

import numpy as np
import pytest

H5_PYTHON = "/opt/conda/bin/python3.9"          # the one interpreter of this image that has h5py
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture
def model_dir(golden_dir):
    return os.path.join(golden_dir, "keras_model_dir")


def test_converted_weights_load_with_openai_names(model_dir):
    from speech_to_text_amd.runtimes.mi355x_runtime import _load_state_dict
    from speech_to_text_amd import weights as W
    sd = _load_state_dict(model_dir)
    dims = W.dims_from_openai_state_dict(sd)
    assert dims["n_mels"] == 80 and dims["n_audio_state"] == dims["n_text_state"] == 128
    assert dims["n_audio_layer"] == dims["n_text_layer"] == 1 and dims["n_vocab"] == 512 and dims["n_text_ctx"] == 32
    assert tuple(sd["encoder.conv1.weight"].shape) == (128, 80, 3)
    assert tuple(sd["decoder.blocks.0.mlp.0.weight"].shape) == (512, 128)
    assert "decoder.blocks.0.cross_attn.key.weight" in sd and "decoder.blocks.0.cross_attn.key.bias" not in sd


def test_unconverted_directory_names_the_converter(model_dir, tmp_path):
    from speech_to_text_amd.runtimes.mi355x_runtime import _load_state_dict
    os.makedirs(tmp_path / "saving")
    shutil.copy(os.path.join(model_dir, "saving", "ckpt-0001.weights.h5"), tmp_path / "saving")
    with pytest.raises(FileNotFoundError, match="convert_keras_weights"):
        _load_state_dict(str(tmp_path))


@pytest.mark.skipif(not os.path.exists(H5_PYTHON), reason="no interpreter with h5py")
def test_converter_round_trip(model_dir, tmp_path):
    """Synthetic checkpoint written in the Keras layout -> converter -> the tensors it was written
    from, and the committed fixture is what the converter makes of the committed checkpoint."""
    tool = os.path.join(REPO, "tools", "convert_keras_weights.py")
    ckpt = str(tmp_path / "ckpt-0007.weights.h5")
    subprocess.run([H5_PYTHON, tool, "--make-synthetic", ckpt], check=True, capture_output=True, timeout=120)
    made, truth = np.load(str(tmp_path / "weights.npz")), np.load(ckpt.replace(".weights.h5", ".truth.npz"))
    assert sorted(made.files) == sorted(truth.files) and len(made.files) == 49
    for k in made.files:
        assert np.array_equal(made[k], truth[k]), k
    out = str(tmp_path / "again.npz")
    subprocess.run([H5_PYTHON, tool, model_dir, "--out", out], check=True, capture_output=True, timeout=120)
    again, committed = np.load(out), np.load(os.path.join(model_dir, "saving", "weights.npz"))
    for k in committed.files:
        assert np.array_equal(again[k], committed[k]), k


def test_reference_tokenizer_file(model_dir):
    from speech_to_text_amd.stt.whisper import WhisperVocabulary
    v = WhisperVocabulary.from_file(os.path.join(model_dir, "saving", "tokenizer.json"))
    names = v.tokenizer.index
    assert v.eos_token_idx == 400 and v.sos_token_idx == 401 and v.timestamp_begin_idx == v.vocab_size == 418
    assert list(v.lang_to_idx) == "en zh de es ru ko fr ja pt tr".split() and v.lang_to_idx["en"] == 402
    assert v.transcribe_token_idx == names["<|transcribe|>"] and v.notimestamps_idx == 417
    space = chr(0x120)                              # byte-level spelling of ' '
    ids = [names[space + "the"], names[space + "cat"], names["s"], names[space + "caf"], names["Ã©"], names["."]]
    assert v.decode(ids + [v.eos_token_idx, 450]) == " the cats café."
    assert v.token_id(" ") == names[space] and v.token_id(" the") == names[space + "the"]
    assert v.token_id("never seen") == []
    non_speech = v.non_speech_token_indexes
    for t in ("(", space + "(", "[", "]", '"', space + "-", space + "'"):
        assert names[t] in non_speech
    assert names["the"] not in non_speech


def test_mel_configuration_file(model_dir):
    from speech_to_text_amd.audio.stft import MelSTFT, WhisperSTFT
    fn = MelSTFT.create(os.path.join(model_dir, "saving", "mel_fn.json"))
    assert isinstance(fn, WhisperSTFT) and fn.n_mel_channels == 80 and fn.hop_length == 160
    assert fn.get_config()["periodic"] is True
