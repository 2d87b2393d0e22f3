"""The in-process HDF5 reader (speech_to_text_amd/hdf5_lite.py) against files h5py wrote
(tests/golden/make_hdf5_fixtures.py) and against the Keras-layout checkpoint fixture: every array
equal to what went in, bit for bit, with its shape and type."""
import os

import numpy as np
import pytest

from speech_to_text_amd.hdf5_lite import Hdf5Error, Hdf5File


@pytest.fixture(scope="module")
def variants(golden_dir):
    folder = os.path.join(golden_dir, "hdf5_variants")
    return folder, np.load(os.path.join(folder, "expected.npz"))


@pytest.mark.parametrize("tag", ["default", "chunked", "latest"])
def test_every_dataset_reads_back(variants, tag):
    """default: symbol-table groups (one with 40 children: a B-tree over several symbol nodes),
    contiguous and compact data, scalar and empty shapes, float16/32/64, integers, big endian.
    chunked: ragged edge chunks, deflate, deflate + shuffle, a chunk B-tree with two levels.
    latest: version-2 object headers and link messages (h5py libver = 'latest')."""
    folder, expected = variants
    want = {k.split(":", 1)[1]: expected[k] for k in expected.files if k.startswith(tag + ":")}
    with Hdf5File(os.path.join(folder, tag + ".h5")) as f:
        assert f.names() == sorted(want)
        assert dict(f.datasets()) == {k: v.shape for k, v in want.items()}
        for name, ref in want.items():
            got = f.read(name)
            assert got.shape == ref.shape and got.dtype == ref.dtype.newbyteorder("="), name
            assert np.array_equal(got, ref), name
            assert got.dtype.isnative and got.flags.writeable


def test_unknown_names_and_foreign_files(variants, tmp_path):
    folder, _ = variants
    with Hdf5File(os.path.join(folder, "default.h5")) as f:
        assert "layers/dense/vars/0" in f and "layers/dense" not in f
        with pytest.raises(KeyError):
            f.read("layers/dense/vars/7")
    other = tmp_path / "not_hdf5.h5"
    other.write_bytes(b"PK\x03\x04" + b"\0" * 4096)
    with pytest.raises(Hdf5Error, match="not an HDF5 file"):
        Hdf5File(str(other))
    cut = tmp_path / "cut.h5"
    cut.write_bytes(open(os.path.join(folder, "default.h5"), "rb").read()[:3000])
    with pytest.raises(Hdf5Error):
        with Hdf5File(str(cut)) as f:
            for name in f.names():
                f.read(name)


def test_keras_checkpoint_fixture(golden_dir):
    """The Keras-layout checkpoint of the model-directory fixture: 49 variables under
    `<layer path>/vars/<i>`, equal to the arrays h5py's own reader gave the converter."""
    from speech_to_text_amd import keras_checkpoint as kc
    model_dir = os.path.join(golden_dir, "keras_model_dir")
    layers = kc.read_vars(kc.latest_checkpoint(model_dir))
    assert sum(len(v) for v in layers.values()) == 49
    assert "encoder/transformer_layers/transformer_layer/attention/wq" in layers
    weights = kc.to_openai(layers)
    committed = np.load(os.path.join(golden_dir, "keras_model_dir_expected.npz"))
    assert sorted(weights) == sorted(committed.files)
    for k in committed.files:
        assert np.array_equal(weights[k], committed[k]), k
