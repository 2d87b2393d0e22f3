"""CPU restatement of the host-side seek loop and timestamp segmentation of the reference's
`Whisper._infer_segments` (models/stt/whisper.py:204-331) and of the language pick of
`detect_language` (:174-202).

TEST INFRASTRUCTURE ONLY, like the rest of oracle/: statement by statement after the cited lines,
with the model call abstracted into `decode(seek, prompt) -> generated ids` so that a test can
script the tokens or record them from the real engine.  PARITY UNPINNED: the reference holds no
test or fixture for this function (its only known answer, notebook cell-13's `end: 3.96`, is
pinned through the frame count in tests/test_oracle_frontend.py).
"""
import numpy as np


class Tokens:
    """The ids `_infer_segments` reads from the model object (whisper.py:80-131)."""

    def __init__(self, sos=50258, eos=50257, transcribe=50359, translate=50358, start_of_prev=50361,
                 timestamp_begin=50364, lang_to_idx=None):
        self.sos, self.eos, self.transcribe, self.translate = sos, eos, transcribe, translate
        self.start_of_prev, self.timestamp_begin = start_of_prev, timestamp_begin
        self.lang_to_idx = lang_to_idx or {}
        self.idx_to_lang = {v: k for k, v in self.lang_to_idx.items()}


def get_inference_tokens(tok, lang=None, task=None):
    """whisper.py:333-338."""
    return [tok.sos, tok.lang_to_idx[lang], tok.translate if task == "translate" else tok.transcribe] \
        if lang else [tok.sos]


def infer_segments(n_frames, decode, tok, *, lang=None, task=None, condition_on_previous_text=True,
                   max_length=256, seek=0, max_input_length=3000, hop_length=160, rate=16000,
                   decode_text=None, max_windows=1000):
    """whisper.py:218-331.  Returns (segments, calls) where calls = [(seek, prompt ids)]."""
    def sample_time(frames):                      # base_audio_model.py:149-153
        return 0. if frames == 0 else frames * hop_length / rate
    time_precision = sample_time(2)
    decode_text = decode_text or (lambda ids: " ".join(str(int(t)) for t in ids if t < tok.eos))
    input_stride = 2
    all_tokens, segments, calls = [], [], []
    while seek < n_frames:
        assert len(calls) < max_windows, "seek does not advance"
        segment_length = min(max_input_length, n_frames - seek)
        tokens = get_inference_tokens(tok, lang=lang, task=task)
        if condition_on_previous_text and len(all_tokens):
            tokens = np.array([tok.start_of_prev] + all_tokens[-(max_length // 2 - 1):] + tokens, "int32")
        else:
            tokens = np.array(tokens, "int32")
        calls.append((int(seek), tokens.tolist()))
        tokens = np.array(decode(int(seek), tokens), dtype=np.int32)
        if lang is None:
            lang = tok.idx_to_lang.get(int(tokens[0]), None)
            tokens = tokens[2:]
        timestamp_offset = sample_time(seek)
        timestamp_tokens = tokens >= tok.timestamp_begin
        consecutive = np.where(np.logical_and(timestamp_tokens[:-1], timestamp_tokens[1:]))[0] + 1
        if len(consecutive) > 0:
            last_slice = 0
            for current_slice in consecutive:
                sliced = tokens[last_slice: current_slice]
                start_pos = sliced[0] - tok.timestamp_begin
                end_pos = sliced[-1] - tok.timestamp_begin
                sliced = sliced[1:-1]
                segments.append({"start": timestamp_offset + start_pos * time_precision,
                                 "end": timestamp_offset + end_pos * time_precision,
                                 "text": decode_text(sliced), "tokens": sliced[sliced < tok.eos], "lang": lang})
                last_slice = current_slice
            last_pos = tokens[last_slice - 1] - tok.timestamp_begin
            seek += int(last_pos) * input_stride
            all_tokens.extend(tokens[: last_slice + 1].tolist())
        else:
            duration = sample_time(segment_length)
            timestamps = tokens[timestamp_tokens]
            if len(timestamps) > 0 and timestamps[-1] != tok.timestamp_begin:
                duration = float(timestamps[-1] - tok.timestamp_begin) * time_precision
            tokens = tokens[tokens < tok.eos]
            segments.append({"start": timestamp_offset, "end": timestamp_offset + duration,
                             "text": decode_text(tokens), "tokens": tokens, "lang": lang})
            seek += max_input_length              # len(segment) of the padded window
            all_tokens.extend(tokens.tolist())
    for segment in segments:
        segment["time"] = segment["end"] - segment["start"]
    return segments, calls


def detect_language(logits_row, language_indexes, languages):
    """whisper.py:184-202: softmax over the language-token logits of the step after <|sot|>."""
    z = np.asarray(logits_row, dtype=np.float64)[list(language_indexes)]
    p = np.exp(z - z.max())
    p = p / p.sum()
    return languages[int(np.argmax(p))], dict(zip(languages, p.tolist()))
