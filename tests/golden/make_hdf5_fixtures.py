#!/usr/bin/env python3
"""Writes tests/golden/hdf5_variants/*.h5 with h5py (run where it exists: in this image
`/opt/conda/bin/python3.9 tests/golden/make_hdf5_fixtures.py`) and expected.npz with the arrays that
went in: the cases the in-process reader (speech_to_text_amd/hdf5_lite.py) is checked against in
tests/test_hdf5_lite.py.  Seeded, so the files can be regenerated bit for bit."""
import os

import h5py
import numpy as np

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "hdf5_variants")
rng = np.random.default_rng(5)
expected = {}


def put(f, tag, name, data, **kw):
    f.create_dataset(name, data=data, **kw)
    expected[tag + ":" + name] = np.asarray(data)


# h5py defaults (what keras.Model.save_weights produces): symbol-table groups, contiguous datasets.
# 40 children in one group force a group B-tree with several symbol nodes.
with h5py.File(os.path.join(HERE, "default.h5"), "w") as f:
    put(f, "default", "layers/dense/vars/0", rng.standard_normal((7, 5)).astype(np.float32))
    put(f, "default", "layers/dense/vars/1", rng.standard_normal(5).astype(np.float32))
    put(f, "default", "scalar", np.float32(2.5))
    put(f, "default", "empty_rows", np.zeros((0, 4), np.float32))
    put(f, "default", "types/f64", rng.standard_normal((3, 3)))
    put(f, "default", "types/f16", rng.standard_normal(9).astype(np.float16))
    put(f, "default", "types/i32", rng.integers(-1000, 1000, 11).astype(np.int32))
    put(f, "default", "types/u8", rng.integers(0, 255, 13).astype(np.uint8))
    put(f, "default", "types/big_endian", rng.standard_normal(6).astype(">f4"))
    for i in range(40):
        put(f, "default", "wide/layer_%d/vars/0" % i, rng.standard_normal(3).astype(np.float32))
    f["layers"].attrs["note"] = "attributes are skipped"

# chunked storage, with and without the deflate / shuffle filters, ragged edge chunks
with h5py.File(os.path.join(HERE, "chunked.h5"), "w") as f:
    a = rng.standard_normal((37, 19)).astype(np.float32)
    put(f, "chunked", "plain", a, chunks=(8, 8))
    put(f, "chunked", "gzip", a, chunks=(16, 19), compression="gzip", compression_opts=4)
    put(f, "chunked", "gzip_shuffle", a, chunks=(10, 7), compression="gzip", shuffle=True)
    put(f, "chunked", "many", rng.standard_normal(5000).astype(np.float32), chunks=(16,))

# the newest file format h5py can write: version-2 object headers, link messages
with h5py.File(os.path.join(HERE, "latest.h5"), "w", libver="latest") as f:
    put(f, "latest", "encoder/norm/vars/0", rng.standard_normal(8).astype(np.float32))
    put(f, "latest", "encoder/norm/vars/1", rng.standard_normal(8).astype(np.float32))
    put(f, "latest", "decoder/table/vars/0", rng.standard_normal((4, 8)).astype(np.float32))

np.savez(os.path.join(HERE, "expected.npz"), **expected)
print("wrote", sorted(os.listdir(HERE)))
