"""MI355X-native Whisper inference path behind the `models.stt` API of yui-mhcp/speech_to_text."""
__all__ = ["stt", "transcribe", "translate", "search"]


def __getattr__(name):
    if name in ("transcribe", "translate", "search", "Whisper"):
        from . import stt
        return getattr(stt, name)
    raise AttributeError(name)
