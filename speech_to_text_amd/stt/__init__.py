from .whisper import Whisper, WhisperVocabulary, transcribe, translate, stream, search, get_model
