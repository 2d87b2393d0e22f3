"""Base class of runtime plugins.

Inside the reference (yui-mhcp/speech_to_text on ``sys.path``) this IS the reference's own
``utils.keras.runtimes.runtime.Runtime``, so ``MI355XRuntime`` is a genuine subclass of the class
``build_runtime`` and ``BaseModel`` know.  Stand-alone (the reference not importable: its package
pulls in keras at import time) a minimal class with the same contract takes its place
(utils/keras/runtimes/runtime.py:19-42): the constructor resolves ``path`` to an engine through a
per-class cache filled by ``load_engine``, and instances are callable.
"""


def _reference_runtime():
    try:
        from utils.keras.runtimes.runtime import Runtime as ref      # the reference's package layout
    except Exception:                                                # not importable here: keras absent
        return None
    ok = all(hasattr(ref, name) for name in ("load_engine", "_engines", "__call__"))
    return ref if ok else None


class _StandaloneRuntime(object):
    """`Runtime(path, engine = None, reload = False, **kw)`: `self.engine` is `engine` if given,
    else the class-wide cached `load_engine(path, **kw)` (loaded again when `reload`)."""
    _engines = {}

    def __init__(self, path, *, engine=None, reload=False, **kwargs):
        self.path = path
        self.engine = engine if engine is not None else self._cached_engine(path, reload, kwargs)

    @classmethod
    def _cached_engine(cls, path, reload, kwargs):
        cache = cls._engines
        if reload or path not in cache:
            cache[path] = cls.load_engine(path, **kwargs)
        return cache[path]

    @staticmethod
    def load_engine(path, **kwargs):
        raise NotImplementedError("a runtime plugin provides load_engine(path, **kwargs)")

    def __call__(self, *args, **kwargs):
        raise NotImplementedError("a runtime plugin provides __call__")

    def __repr__(self):
        return "<%s path=%s>" % (type(self).__name__, self.path)


Runtime = _reference_runtime() or _StandaloneRuntime
