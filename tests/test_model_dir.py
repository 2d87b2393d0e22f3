"""A model directory as the reference writes it (pretrained_models/<name>/saving/{*.weights.h5,
config_models.json, tokenizer.json, mel_fn.json}, models/interfaces/base_model.py:126-136) read
in-process, without Keras or h5py: the checkpoint, the serialised model configuration, the
reference-format tokenizer file and the mel configuration.  The fixture under
tests/golden/keras_model_dir is synthetic (tools/convert_keras_weights.py --make-synthetic wrote the
checkpoint through h5py; tests/golden/keras_model_dir_expected.npz is what h5py's own reader makes of
it): no real reference checkpoint is available here, so the variable layout is the one the
reference's layer classes declare, not one read back from a file the reference wrote (parity
unpinned)."""
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


def test_directory_holds_only_what_the_reference_writes(model_dir):
    assert sorted(os.listdir(os.path.join(model_dir, "saving"))) == [
        "ckpt-0001.weights.h5", "config_models.json", "mel_fn.json", "tokenizer.json"]


def test_checkpoint_loads_with_openai_names(model_dir, golden_dir):
    from speech_to_text_amd.runtimes.mi355x_runtime import _load_state_dict, _load_weights
    from speech_to_text_amd import weights as W
    sd = _load_state_dict(model_dir)
    expected = np.load(os.path.join(golden_dir, "keras_model_dir_expected.npz"))
    assert sorted(sd) == sorted(expected.files)
    for k in expected.files:
        assert np.array_equal(sd[k].numpy(), expected[k]), k
    assert _load_weights(model_dir)[1] is True               # config_models.json: 'gelu' = the tanh form
    dims = W.dims_from_openai_state_dict(sd)
    assert dims["n_mels"] == 80 and dims["n_audio_state"] == dims["n_text_state"] == 128
    assert dims["n_audio_layer"] == dims["n_text_layer"] == 1 and dims["n_vocab"] == 512 and dims["n_text_ctx"] == 32
    assert tuple(sd["encoder.conv1.weight"].shape) == (128, 80, 3)
    assert tuple(sd["decoder.blocks.0.mlp.0.weight"].shape) == (512, 128)
    assert "decoder.blocks.0.cross_attn.key.weight" in sd and "decoder.blocks.0.cross_attn.key.bias" not in sd


def test_model_configuration_is_read_and_checked(model_dir, tmp_path):
    """config_models.json: sizes and the GELU form; a size the checkpoint contradicts is an error,
    an exact-erf activation selects the other kernel form, the newest checkpoint wins."""
    import json
    from speech_to_text_amd import keras_checkpoint as kc
    from speech_to_text_amd.runtimes.mi355x_runtime import MI355XRuntime
    cfg = kc.model_config(model_dir)
    assert cfg["gelu_tanh"] is True
    assert cfg["dims"] == dict(n_mels=80, n_audio_state=128, n_audio_layer=1, n_audio_head=2, n_audio_ctx=1500,
                               n_vocab=512, n_text_state=128, n_text_layer=1, n_text_head=2, n_text_ctx=32)
    assert MI355XRuntime.load_engine(model_dir).gelu_tanh is True
    copy = tmp_path / "model"
    shutil.copytree(model_dir, copy)
    path = copy / "saving" / "config_models.json"
    config = json.load(open(path))
    config["model"]["config"]["decoder_ffn_activation"] = {"class_name": "gelu", "config": {"approximate": False}}
    config["model"]["config"]["encoder_ffn_activation"] = "gelu_exact"
    config["model"]["config"]["encoder_activation"] = "gelu_exact"
    json.dump(config, open(path, "w"))
    assert MI355XRuntime.load_engine(str(copy)).gelu_tanh is False
    assert MI355XRuntime.load_engine(str(copy), gelu="tanh").gelu_tanh is True
    config["model"]["config"]["decoder_num_layers"] = 3
    json.dump(config, open(path, "w"))
    with pytest.raises(ValueError, match="n_text_layer"):
        MI355XRuntime.load_engine(str(copy))
    config["model"]["config"]["decoder_num_layers"] = 1
    config["model"]["config"]["decoder_ffn_activation"] = "relu"
    json.dump(config, open(path, "w"))
    with pytest.raises(ValueError, match="GELU"):
        MI355XRuntime.load_engine(str(copy))
    shutil.copy(copy / "saving" / "ckpt-0001.weights.h5", copy / "saving" / "ckpt-0003.weights.h5")
    assert kc.latest_checkpoint(str(copy)).endswith("ckpt-0003.weights.h5")
    # with the manager's own state file: the checkpoint it would restore (checkpoint_manager.py:88-98)
    state = {"counter": 4, "loaded": 0, "best_checkpoint": {},
             "checkpoints": [{"epoch": 1, "step": 10, "counter": 1}, {"epoch": 3, "step": 30, "counter": 3}]}
    json.dump(state, open(copy / "saving" / "checkpoint.json", "w"))
    assert kc.latest_checkpoint(str(copy)).endswith("ckpt-0001.weights.h5")
    state["loaded"] = 1
    json.dump(state, open(copy / "saving" / "checkpoint.json", "w"))
    assert kc.latest_checkpoint(str(copy)).endswith("ckpt-0003.weights.h5")
    state["loaded"] = "best"
    json.dump(state, open(copy / "saving" / "checkpoint.json", "w"))
    assert kc.latest_checkpoint(str(copy)).endswith("ckpt-0003.weights.h5")        # no best.weights.h5: the newest
    shutil.copy(copy / "saving" / "ckpt-0001.weights.h5", copy / "saving" / "best.weights.h5")
    assert kc.latest_checkpoint(str(copy)).endswith("best.weights.h5")
    assert kc.load_model_dir(str(tmp_path)) is None


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
    again, committed = np.load(out), np.load(os.path.join(os.path.dirname(model_dir), "keras_model_dir_expected.npz"))
    for k in committed.files:
        assert np.array_equal(again[k], committed[k]), k
    # the in-process reader on the freshly written file: same arrays as h5py's
    from speech_to_text_amd import keras_checkpoint as kc
    mine = kc.to_openai(kc.read_vars(ckpt))
    assert sorted(mine) == sorted(truth.files)
    for k in truth.files:
        assert np.array_equal(mine[k], truth[k]), k


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
