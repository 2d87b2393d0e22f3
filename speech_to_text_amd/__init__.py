"""MI355X-native Whisper inference path behind the `models.stt` API of yui-mhcp/speech_to_text."""
__all__ = ["stt", "transcribe", "translate", "stream", "search"]


def __getattr__(name):
    if name in ("transcribe", "translate", "stream", "search", "Whisper"):
        from . import stt
        return getattr(stt, name)
    raise AttributeError(name)
