"""The shared library loads and exports every symbol the header declares (no GPU needed)."""
import os
import re

from speech_to_text_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols():
    text = open(os.path.join(ROOT, "include", "stt_mi355x.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(stt_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    import ctypes
    assert os.path.exists(_lib.LIB_PATH), "build with make -C speech_to_text_amd/csrc"
    lib = ctypes.CDLL(_lib.LIB_PATH)
    declared = _header_symbols()
    assert len(declared) >= 9
    for name in declared:
        assert hasattr(lib, name), "missing export: " + name
    assert sorted(_lib.exported_symbols()) == declared


def test_host_only_entry_points():
    lib = _lib.load()
    assert lib.stt_abi_version() >= 1
    assert lib.stt_logmel_num_frames(1) == 2
    assert lib.stt_logmel_num_frames(400) == 2
    assert lib.stt_logmel_num_frames(64880) == 405
    assert lib.stt_logmel_num_frames(57_600_000) == 360000
    assert lib.stt_logmel_workspace_bytes(57_600_000) >= 4 * 4 * 1458
