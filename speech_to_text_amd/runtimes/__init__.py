"""Runtime registry with the reference's ``build_runtime`` contract
(utils/keras/runtimes/__init__.py:23-45): unknown names raise ``ValueError``."""
from .runtime import Runtime
from .mi355x_runtime import MI355XRuntime, InferenceOutput


def build_runtime(runtime, path, *args, **kwargs):
    if runtime not in _runtimes:
        raise ValueError('Unsupported runtime !\n  Accepted : {}\n  Got : {}'.format(
            tuple(_runtimes.keys()), runtime
        ))
    return _runtimes[runtime](path, *args, **kwargs)


_runtimes = {
    'mi355x': MI355XRuntime,
}
