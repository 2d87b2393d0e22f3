"""ctypes binding of libstt_mi355x.so (the C ABI declared in include/stt_mi355x.h).

The library is built in-tree by ``speech_to_text_amd/csrc/Makefile``.  There is
no fallback: if it is missing, loading raises.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("STT_MI355X_LIB", os.path.join(_HERE, "libstt_mi355x.so"))

_c = ctypes
_lib = None
ABI_VERSION = 5          # include/stt_mi355x.h: bumped on any signature or structure change


class SttError(RuntimeError):
    def __init__(self, code, message):
        super().__init__("libstt_mi355x error %d: %s" % (code, message))
        self.code = code


def _declare(lib):
    vp, i32, i64, sz = _c.c_void_p, _c.c_int, _c.c_int64, _c.c_size_t
    fp = _c.POINTER(_c.c_float)
    sigs = {
        "stt_last_error": (_c.c_char_p, []),
        "stt_abi_version": (i32, []),
        "stt_logmel_plan_create": (i32, [i32, fp, fp, _c.POINTER(vp)]),
        "stt_logmel_plan_destroy": (None, [vp]),
        "stt_logmel_num_frames": (i64, [i64]),
        "stt_logmel_workspace_bytes": (sz, [i64]),
        "stt_logmel_run": (i32, [vp, vp, i64, vp, vp, sz, vp]),
        "stt_logmel_static_max_samples": (i64, []),
        "stt_logmel_run_static": (i32, [vp, vp, i64, vp, vp]),
        "stt_logmel_stage1": (i32, [vp, vp, i64, i64, i64, i64, i64, vp, vp, vp, sz, vp]),
        "stt_logmel_stage2": (i32, [vp, vp, i64, vp, vp, vp]),
        "stt_pcm_workspace_bytes": (sz, []),
        "stt_pcm_normalize": (i32, [vp, i32, i64, vp, vp, sz, vp]),
        "stt_pcm_block_peaks": (i32, [vp, i64, i32, vp, vp]),
    }
    for name, (res, args) in sigs.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    return sigs


def load():
    """Load the shared library once; raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                "%s not found: build it with `make -C %s` (hipcc, gfx950)"
                % (LIB_PATH, os.path.join(_HERE, "csrc")))
        # torch ships its own HIP runtime; it has to be the one already in the process when this
        # library's libamdhip64 dependency is resolved, or the two runtimes disagree about devices
        import torch  # noqa: F401
        lib = ctypes.CDLL(LIB_PATH)
        _declare(lib)
        if lib.stt_abi_version() != ABI_VERSION:        # the ctypes structures here mirror one version
            raise ImportError("%s has ABI version %d, this package binds version %d: rebuild it with "
                              "`make -C %s`" % (LIB_PATH, lib.stt_abi_version(), ABI_VERSION,
                                                os.path.join(_HERE, "csrc")))
        _lib = lib
    return _lib


def check(code):
    if code != 0:
        raise SttError(code, load().stt_last_error().decode("utf-8", "replace"))


def exported_symbols():
    """Names this binding expects the library to export (= include/stt_mi355x.h)."""
    from .engine import ENGINE_SYMBOLS
    return sorted(list(_declare(ctypes.CDLL(LIB_PATH)).keys()) + list(ENGINE_SYMBOLS))
