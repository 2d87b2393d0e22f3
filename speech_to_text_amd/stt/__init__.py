from .whisper import Whisper, WhisperVocabulary, transcribe, translate, search, get_model
