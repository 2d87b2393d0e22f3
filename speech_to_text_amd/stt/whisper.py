"""``Whisper`` speech-to-text wrapper over the MI355X runtime, with the method names, arguments
and result dictionaries of the reference's ``models/stt/whisper.py`` / ``base_stt.py``:
``predict`` / ``infer`` / ``_infer_segments`` / ``detect_language`` / ``get_inference_tokens``
plus ``search`` (README "Search and partial alignment").

Host code (Python) like the reference's wrapper; everything numeric runs in the runtime.
"""
import os

import numpy as np
import torch

from ..audio.audio_processing import WHISPER_TRIM_KWARGS, read_audio
from ..audio.stft import WhisperSTFT
from ..engine import TimestampFilter
from ..runtimes import build_runtime
from ..text.metrics import edit_distance, process_model_output

# language codes in the order of the tokenizer's language tokens (whisper.py:418-518)
LANGUAGES = ("en zh de es ru ko fr ja pt tr pl ca nl ar sv it id hi fi vi he uk el ms cs ro da hu ta "
             "no th ur hr bg lt la mi ml cy sk te fa lv bn sr az sl kn et mk br eu is hy ne mn bs kk "
             "sq sw gl mr pa si km sn yo so af oc ka be tg sd gu am yi lo uz fo ht ps tk nn mt sa lb "
             "my bo tl mg as tt haw ln ha ba jw su").split()


class _ReferenceTokenizer(object):
    """Decoding and single-token lookups from the reference's own tokenizer file: token strings in
    byte-level BPE spelling (`vocab`) and the byte <-> character table (`byte_encoder`).  Enough
    for what the wrapper needs (ids -> text, ' ' and symbol lookups); encoding free text is the
    caller's business, as the prompt ids are."""

    def __init__(self, config):
        self.vocab = list(config['vocab'])
        self.index = {t: i for i, t in enumerate(self.vocab)}
        table = config.get('byte_encoder') or {}
        self.to_char = {int(k): v for k, v in table.items()}
        self.to_byte = {v: k for k, v in self.to_char.items()}

    def decode(self, ids):
        text = ''.join(self.vocab[i] for i in ids if 0 <= i < len(self.vocab))
        if not self.to_byte:
            return text
        return bytearray(self.to_byte.get(c, ord('?')) for c in text).decode('utf-8', errors='replace')

    def encode(self, text, add_special_tokens=False):
        spelled = ''.join(self.to_char.get(b, '') for b in text.encode('utf-8')) if self.to_char else text
        hit = self.index.get(spelled)
        return type('Encoding', (), {'ids': [hit] if hit is not None else []})()


class WhisperVocabulary(object):
    """Special-token ids of the `openai/whisper-base` tokenizer the reference always loads
    (utils/text/tokenizer.py:870-872): 50364 entries, timestamps start right after.  An optional
    `tokenizers.Tokenizer` (from a local tokenizer.json) provides text decoding and the
    non-speech token list; without it token ids are returned as text."""

    def __init__(self, tokenizer=None, vocab_size=50364):
        self.tokenizer = tokenizer
        self.vocab_size = vocab_size
        self.eos_token_idx = 50257
        self.sos_token_idx = 50258
        self.lang_to_idx = {l: 50259 + i for i, l in enumerate(LANGUAGES)}
        self.idx_to_lang = {i: l for l, i in self.lang_to_idx.items()}
        self.translate_token_idx = 50358
        self.transcribe_token_idx = 50359
        self.start_of_lm_idx = 50360
        self.start_of_prev_token_idx = 50361
        self.nospeech_token_idx = 50362
        self.notimestamps_idx = 50363
        self.timestamp_begin_idx = vocab_size
        if vocab_size == 50365:
            # the large-v3 tokenizer: one more language token (<|yue|>) in front of the task tokens
            # moves every id after it up by one (n_vocab 51866 = 50365 + 1501 timestamps)
            self.lang_to_idx['yue'] = 50358
            self.idx_to_lang[50358] = 'yue'
            self.translate_token_idx, self.transcribe_token_idx = 50359, 50360
            self.start_of_lm_idx, self.start_of_prev_token_idx = 50361, 50362
            self.nospeech_token_idx, self.notimestamps_idx = 50363, 50364
        names = getattr(tokenizer, 'index', None)
        if names and '<|endoftext|>' in names and '<|startoftranscript|>' in names:
            # ids from the file instead of the whisper-base constants (a large-v3 tokenizer has one
            # more language, which shifts every id after it)
            self.eos_token_idx, self.sos_token_idx = names['<|endoftext|>'], names['<|startoftranscript|>']
            self.lang_to_idx = {l: names['<|%s|>' % l] for l in LANGUAGES + ['yue'] if '<|%s|>' % l in names}
            self.idx_to_lang = {i: l for l, i in self.lang_to_idx.items()}
            for attr, token in (('translate_token_idx', '<|translate|>'), ('transcribe_token_idx', '<|transcribe|>'),
                                ('start_of_lm_idx', '<|startoflm|>'), ('start_of_prev_token_idx', '<|startofprev|>'),
                                ('nospeech_token_idx', '<|nospeech|>'), ('notimestamps_idx', '<|notimestamps|>')):
                if token in names:
                    setattr(self, attr, names[token])
            self.vocab_size = self.timestamp_begin_idx = names.get('<|0.00|>', len(names))

    @classmethod
    def from_file(cls, tokenizer_json):
        """A Hugging Face `tokenizer.json`, or the `saving/tokenizer.json` the reference writes
        (`Tokenizer.save`, utils/text/tokenizer.py:671-706: a `vocab` list and a `byte_encoder`)."""
        import json
        with open(tokenizer_json, encoding='utf-8') as f:
            head = json.load(f)
        if isinstance(head.get('vocab'), list):
            return cls(_ReferenceTokenizer(head))
        from tokenizers import Tokenizer
        return cls(Tokenizer.from_file(tokenizer_json))

    def token_id(self, text):
        if self.tokenizer is None:
            return None
        ids = self.tokenizer.encode(text, add_special_tokens=False).ids
        return ids[0] if len(ids) == 1 else ids

    @property
    def special_token_indexes(self):
        return [self.sos_token_idx, self.start_of_lm_idx, self.start_of_prev_token_idx,
                self.nospeech_token_idx, self.notimestamps_idx]

    @property
    def non_speech_token_indexes(self):
        """OpenAI's non-speech symbol set (whisper.py:132-153); needs the tokenizer."""
        if self.tokenizer is None:
            return ()
        symbols = list("\"#()*+/:;<=>@[\\]^_`{|}~「」『』")
        symbols += "<< >> <<< >>> -- --- -( -[ (' (\" (( )) ((( ))) [[ ]] {{ }} ♪♪ ♪♪♪".split()
        misc = set("♩♪♫♬♭♮♯")
        result = set()
        for t in (" -", " '"):
            i = self.token_id(t)
            if isinstance(i, int): result.add(i)
        for symbol in symbols + list(misc):
            for toks in (self.token_id(symbol), self.token_id(" " + symbol)):
                if isinstance(toks, int):
                    result.add(toks)
                elif symbol in misc and toks:
                    result.add(toks[0])
        return tuple(sorted(result))

    def decode(self, tokens):
        tokens = [int(t) for t in tokens if int(t) < self.eos_token_idx]
        if self.tokenizer is None:
            return ' '.join(str(t) for t in tokens)
        return self.tokenizer.decode(tokens)


class Whisper(object):
    """`Whisper(pretrained = 'large-v3', runtime = 'mi355x', path = ...)`."""

    def __init__(self, lang='multi', pretrained='base', *, runtime='mi355x', path=None,
                 vocabulary=None, max_batch=1, **kwargs):
        self.lang = 'en' if pretrained and '.en' in pretrained else lang
        self.pretrained = pretrained
        self.rate = 16000
        self.max_input_length = 3000
        self.max_output_length = 256
        self.pad_mel_value = 0.
        if vocabulary is None and isinstance(path, str) and os.path.isfile(os.path.join(path, 'saving', 'tokenizer.json')):
            vocabulary = WhisperVocabulary.from_file(os.path.join(path, 'saving', 'tokenizer.json'))
        self.vocabulary = vocabulary or WhisperVocabulary()
        if path is None:
            import warnings
            warnings.warn("Whisper(pretrained = {!r}) without `path`: no checkpoint can be downloaded here, "
                          "using SEEDED RANDOM weights - transcripts will be meaningless. Pass path = <weights> "
                          "(or path = 'random:<size>' to silence this)".format(pretrained), stacklevel=2)
            path = 'random:{}'.format(pretrained)
        self.model = build_runtime(runtime, path, max_batch=max_batch,
                                   **{k: v for k, v in kwargs.items() if k in ('gelu', 'reload')})
        if vocabulary is None and self.model.dims['n_vocab'] == 51866:
            # no tokenizer given and large-v3 weights: the ids of that checkpoint's own tokenizer
            # rather than whisper-base's, which the reference always loads (utils/text/tokenizer.py:
            # 870-872) and which is off by one from <|translate|> on for this model
            self.vocabulary = WhisperVocabulary(vocab_size=50365)
        n_mels = self.model.dims['n_mels']
        mel_file = os.path.join(path, 'saving', 'mel_fn.json') if isinstance(path, str) else ''
        if os.path.isfile(mel_file):                  # written by the reference's MelSTFT.save (stft.py:164-182)
            from ..audio.stft import MelSTFT
            self.mel_fn = MelSTFT.create(mel_file)
            if self.mel_fn.n_mel_channels != n_mels:
                raise ValueError('{} has {} mel channels, the weights expect {}'.format(
                    mel_file, self.mel_fn.n_mel_channels, n_mels))
        else:
            self.mel_fn = WhisperSTFT(n_mel_channels=n_mels)
        self.trim_kwargs = dict(trim_silence=True, **WHISPER_TRIM_KWARGS)
        self._device_audio = None
        self.model.set_tokens(sos_token=self.sos_token_idx, eos_token=self.eos_token_idx,
                              pad_token=self.eos_token_idx)
        t = list(self.vocabulary.non_speech_token_indexes) + list(self.vocabulary.special_token_indexes)
        self.remove_tokens = np.array(t, 'int32')
        space = self.vocabulary.token_id(' ')
        self.remove_tokens_with_space = np.array(
            ([space] if isinstance(space, int) else []) + [self.eos_token_idx] + t, 'int32')

    # ---- token properties (whisper.py:80-131)
    sos_token_idx = property(lambda self: self.vocabulary.sos_token_idx)
    eos_token_idx = property(lambda self: self.vocabulary.eos_token_idx)
    transcribe_token_idx = property(lambda self: self.vocabulary.transcribe_token_idx)
    translate_token_idx = property(lambda self: self.vocabulary.translate_token_idx)
    start_of_prev_token_idx = property(lambda self: self.vocabulary.start_of_prev_token_idx)
    timestamp_begin_idx = property(lambda self: self.vocabulary.timestamp_begin_idx)
    languages = property(lambda self: list(self.vocabulary.lang_to_idx.keys()))
    language_indexes = property(lambda self: list(self.vocabulary.idx_to_lang.keys()))
    is_encoder_decoder = True
    compiled_infer = property(lambda self: self.model)       # base_model.py:365-374

    @property
    def time_precision(self):
        return 2 * self.mel_fn.hop_length / self.rate          # 0.02 s

    def _get_sample_time(self, frames):
        return 0. if frames == 0 else frames * self.mel_fn.hop_length / self.rate

    @property
    def logits_filter(self):
        """The reference's opt-in timestamp filter (whisper.py:354-416) as a device-side filter."""
        return TimestampFilter(self.remove_tokens, self.remove_tokens_with_space,
                               self.timestamp_begin_idx, self.eos_token_idx)

    # ---- inputs
    def get_audio(self, data, device_pipeline=False, **kwargs):
        """Normalised, silence-trimmed clip: a float32 numpy array from the host pre-steps (what the
        reference runs, utils/audio/audio_io.py:249-264), or with `device_pipeline = True` a CUDA
        tensor from the same steps run on the GPU (audio/device_pipeline.py: the clip crosses the
        link once, as int16)."""
        if device_pipeline:
            if self._device_audio is None:
                from ..audio.device_pipeline import DeviceAudioPipeline
                self._device_audio = DeviceAudioPipeline(rate=self.rate)
            return self._device_audio(data, **{**self.trim_kwargs, **kwargs})[1]
        _, audio = read_audio(data, rate=self.rate, **{**self.trim_kwargs, **kwargs})
        return audio

    def get_input(self, audio, pad_or_trim=True, **kwargs):
        """Raw audio / filename -> mel [frames, n_mels] (CUDA tensor)."""
        if isinstance(audio, dict):
            audio = audio.get('audio', audio.get('filename'))
        if hasattr(audio, 'shape') and len(audio.shape) >= 2 and audio.shape[-1] == self.mel_fn.n_mel_channels:
            mel = torch.as_tensor(audio)
        else:
            keep = {k: v for k, v in kwargs.items() if k in ('normalize', 'trim_silence', 'mode', 'threshold',
                                                            'min_silence', 'replace_by', 'min_voice_time',
                                                            'device_pipeline', 'rate')}
            mel = self.mel_fn(self.get_audio(audio, **keep))[0]
        return self.pad_or_trim(mel) if pad_or_trim else mel

    def pad_or_trim(self, mel):
        if mel.shape[0] > self.max_input_length:
            return mel[: self.max_input_length]
        if mel.shape[0] < self.max_input_length:
            return torch.nn.functional.pad(mel, (0, 0, 0, self.max_input_length - mel.shape[0]),
                                           value=self.pad_mel_value)
        return mel

    def get_inference_tokens(self, lang=None, task=None, **_):
        if not lang:
            return [self.sos_token_idx]
        return [self.sos_token_idx, self.vocabulary.lang_to_idx[lang],
                self.translate_token_idx if task == 'translate' else self.transcribe_token_idx]

    def decode_output(self, tokens):
        return self.vocabulary.decode(tokens)

    # ---- language detection (whisper.py:174-202)
    def detect_language(self, audio, **kwargs):
        mel = self.get_input(audio, pad_or_trim=True)
        if mel.dim() == 2:
            mel = mel[None]
        tokens = np.full((mel.shape[0], 1), self.sos_token_idx, dtype=np.int32)
        enc = self.model.encoder(mel)
        pred = self.model.decoder(tokens, encoder_output=enc)
        idx = torch.as_tensor(self.language_indexes, device=pred.device)
        probs = torch.softmax(pred[0, -1, :][idx], dim=-1).cpu().numpy()
        return self.languages[int(np.argmax(probs))], {l: float(p) for l, p in zip(self.languages, probs)}

    # ---- sequential long-form decoding (behaviour of whisper.py:204-331)
    def _prompt_for(self, lang, history, max_length, **kwargs):
        """[<|startofprev|>, last max_length // 2 - 1 tokens of `history`] + the task prompt."""
        task = self.get_inference_tokens(lang=lang, **kwargs)
        if not history:
            return np.asarray(task, dtype=np.int32)
        tail = history[-(max_length // 2 - 1):]
        return np.asarray([self.start_of_prev_token_idx] + list(tail) + task, dtype=np.int32)

    def _decode_window(self, window, prompt, **kwargs):
        """One 30 s window through the runtime -> generated ids of the best sentence (no prompt, no EOS)."""
        out = self.compiled_infer(window[None], tokens=prompt[None], tokens_length=len(prompt), **kwargs)
        ids = process_model_output(out)[0] if hasattr(out, 'tokens') else out[0]
        while len(ids) and isinstance(ids[0], (list, np.ndarray)):      # [beam][token] -> best beam
            ids = ids[0]
        return np.asarray(ids, dtype=np.int32)

    def _split_at_timestamps(self, ids, offset_s, window_s, lang):
        """Segments of one window's ids, the number of mel frames they account for (None: the whole
        window) and how many leading ids join the history for the next prompt.

        Two timestamp tokens in a row close one segment and open the next, so the ids are cut
        after the first of every such pair; every piece then reads <start> text ... <end>.  What
        follows the last pair is dropped and decoded again: the next window starts at the last
        closed timestamp.  Without any pair the window is one segment that lasts until its last
        timestamp (if any, and not the zero one) or the end of the window."""
        t0, step = self.timestamp_begin_idx, self.time_precision
        stamp = ids >= t0
        pairs = np.flatnonzero(stamp[1:] & stamp[:-1]) + 1
        if pairs.size == 0:
            text = ids[ids < self.eos_token_idx]
            marks = ids[stamp]
            length_s = window_s if not (marks.size and marks[-1] != t0) else float(marks[-1] - t0) * step
            seg = {'start': offset_s, 'end': offset_s + length_s, 'text': self.decode_output(text),
                   'tokens': text, 'lang': lang}
            return [seg], None, text.tolist()
        segments = []
        for piece in np.split(ids[: pairs[-1]], pairs[:-1]):
            body = piece[1:-1]
            segments.append({'start': offset_s + (piece[0] - t0) * step, 'end': offset_s + (piece[-1] - t0) * step,
                             'text': self.decode_output(body), 'tokens': body[body < self.eos_token_idx],
                             'lang': lang})
        closed_at = int(ids[pairs[-1] - 1] - t0)
        return segments, closed_at * 2, ids[: pairs[-1] + 1].tolist()      # 2 mel frames per timestamp unit

    def _infer_segments(self, mel, *, lang=None, verbose=True, force_detect_language=False,
                        condition_on_previous_text=True, segment_processing=None, **kwargs):
        kwargs['encoder_output_lengths'] = 1500
        max_length = kwargs.setdefault('max_length', self.max_output_length)
        seek, total = kwargs.pop('seek', 0), len(mel)
        history, segments = [], []
        while seek < total:
            valid = min(self.max_input_length, total - seek)
            window = self.pad_or_trim(mel[seek: seek + valid])
            if lang is None and force_detect_language:
                lang, _ = self.detect_language(window)
            prompt = self._prompt_for(lang, history if condition_on_previous_text else (), **kwargs)
            ids = self._decode_window(window, prompt, **kwargs)
            if lang is None:                 # prompt was <|sot|> alone: the model wrote language and task
                lang = self.vocabulary.idx_to_lang.get(int(ids[0])) if len(ids) else None
                ids = ids[2:]
            new, frames, kept = self._split_at_timestamps(ids, self._get_sample_time(seek),
                                                          self._get_sample_time(valid), lang)
            for seg in new:
                seg['time'] = seg['end'] - seg['start']
                if segment_processing is not None:
                    segment_processing(seg, segment=window)
            segments += new
            history += kept
            # a pair closed at 0.00 s would leave `seek` where it is for ever (the reference's loop,
            # whisper.py:296-299, does not terminate on such an output): move on by the whole window
            seek += valid if not frames else frames
        return segments

    # ---- independent windows, batched (and shardable across GPUs)
    def infer_windows(self, mel, *, lang='en', task=None, batch_size=32, shard=None, replicas=1, **kwargs):
        """Transcribes the 30 s windows of `mel` independently of each other (the mode the
        reference reaches with condition_on_previous_text = False and no timestamp seek,
        whisper.py:301-323), `batch_size` windows per engine call.  `shard = (rank, world)`
        keeps the windows w with w % world == rank.

        `replicas = 2` runs the batches alternately on two engines built from the same weights,
        each on its own host thread and stream, the second one phase late: while one engine decodes
        a batch (bound by HBM: the cross-attention K/V) the other encodes the next (bound by the
        matrix cores), and the two passes share the GPU.  Results are the same as with one."""
        n_frames = len(mel)
        starts = list(range(0, n_frames, self.max_input_length))
        mine = [w for w in range(len(starts)) if shard is None or w % shard[1] == shard[0]]
        prompt = np.array(self.get_inference_tokens(lang=lang, task=task), 'int32')
        kwargs.setdefault('max_length', self.max_output_length)
        batches = [mine[i: i + batch_size] for i in range(0, len(mine), batch_size)]
        results = {}

        def run(ws, **extra):
            batch = torch.stack([self.pad_or_trim(mel[starts[w]: starts[w] + self.max_input_length]) for w in ws])
            out = self.compiled_infer(batch, tokens=np.tile(prompt[None], [len(ws), 1]),
                                      tokens_length=len(prompt), **kwargs, **extra)
            toks = process_model_output(out)
            for w, t in zip(ws, toks):
                while len(t) and isinstance(t[0], (list, np.ndarray)):
                    t = t[0]
                t = np.asarray(t, dtype=np.int32)
                seg_len = min(self.max_input_length, n_frames - starts[w])
                results[w] = {'start': self._get_sample_time(starts[w]),
                              'end': self._get_sample_time(starts[w] + seg_len),
                              'tokens': t[t < self.eos_token_idx], 'lang': lang,
                              'text': self.decode_output(t)}

        if replicas < 2 or len(batches) < 2 or not isinstance(mel, torch.Tensor) or not mel.is_cuda:
            for ws in batches:
                run(ws)
            return results

        import threading
        torch.cuda.synchronize(mel.device)
        first_encoded = threading.Event()
        box, errors = {}, []

        def worker(r):
            try:
                with torch.cuda.device(mel.device), torch.cuda.stream(torch.cuda.Stream(mel.device)):
                    if r == 1:
                        first_encoded.wait()
                        if 'event' in box:
                            box['event'].synchronize()
                    for i, ws in enumerate(batches[r::2]):
                        if r == 0 and i == 0:
                            run(ws, replica=r, on_encoded=lambda ev: (box.__setitem__('event', ev), first_encoded.set()))
                        else:
                            run(ws, replica=r)
            except BaseException as e:        # surface worker failures in the caller
                errors.append(e)
            finally:
                first_encoded.set()

        threads = [threading.Thread(target=worker, args=(r,)) for r in range(2)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        if errors:
            raise errors[0]
        return results

    def infer(self, audio, *, condition_on_previous_text=True, independent_windows=False, **kwargs):
        infos = audio.copy() if isinstance(audio, dict) else {}
        mel = self.get_input(audio, pad_or_trim=False, **kwargs)
        run_kwargs = {k: v for k, v in kwargs.items() if k not in
                      ('normalize', 'trim_silence', 'mode', 'threshold', 'min_silence', 'replace_by',
                       'min_voice_time', 'device_pipeline', 'rate')}
        if independent_windows:
            res = self.infer_windows(mel, **run_kwargs)
            segments = [res[w] for w in sorted(res)]
            for seg in segments:
                seg['time'] = seg['end'] - seg['start']
        else:
            segments = self._infer_segments(mel, condition_on_previous_text=condition_on_previous_text,
                                            **run_kwargs)
        infos.update({'text': ' '.join(seg['text'] for seg in segments), 'segments': segments})
        if 'lang' not in infos and segments and 'lang' in segments[0]:
            infos['lang'] = segments[0]['lang']
        if isinstance(audio, str):
            infos['filename'] = audio
        return infos

    def predict(self, audios, **kwargs):
        if not isinstance(audios, (list, tuple)):
            audios = [audios]
        return [self.infer(a, **kwargs) for a in audios]

    def stream(self, stream, *, callback=None, **kwargs):
        """`predict` over inputs that arrive one at a time, each result handed to `callback` instead of
        being collected (models/stt/base_stt.py:391-396 over models/interfaces/base_model.py:714:
        `predict(..., return_output = False, return_results = False)` fed by the reference's `Stream`,
        which runs in the caller's thread here: `max_workers = 0`).  `stream` may be an iterable or
        generator, a queue (anything with `.get()`; a `None` item ends it) or a callable returning the
        next input (`None` ends it).  The frontend is run once on 0.5 s and 1 s of silence first, as the
        reference does, so that the first real clip does not pay for plan creation.  Returns the number of
        inputs processed."""
        for length in (self.rate // 2, self.rate):
            self.get_input({'audio': np.zeros((length,), dtype='float32'), 'rate': self.rate},
                           normalize=False, trim_silence=False)
        def until_none(next_item):              # (`iter(f, None)` would compare arrays with None)
            while True:
                item = next_item()
                if item is None:
                    return
                yield item

        if hasattr(stream, 'get') and not isinstance(stream, dict):
            items = until_none(stream.get)
        elif callable(stream):
            items = until_none(stream)
        else:
            items = iter(stream)
        count = 0
        for item in items:
            result = self.infer(item, **kwargs)
            count += 1
            if callback is not None:
                callback(result)
        return count

    # ---- keyword search by partial alignment (README "Search and partial alignment")
    def search(self, keyword, audio, threshold=0.8, **kwargs):
        """Transcribes with timestamps and scores `keyword` against every segment's text with
        edit_distance(partial = True); returns the segments whose similarity 1 - distance reaches
        `threshold`, each with the character span of the best alignment."""
        kwargs.setdefault('logits_filter', self.logits_filter)
        result = audio if isinstance(audio, dict) and 'segments' in audio else self.infer(audio, **kwargs)
        hits = []
        for seg in result['segments']:
            text = seg['text']
            if not text:
                continue
            dist, matrix = edit_distance(keyword, text, partial=True, return_matrix=True, normalize=True)
            score = 1. - dist
            if score >= threshold:
                end = int(np.argmin(matrix[-1, 1:])) + 1
                hits.append({'start': seg['start'], 'end': seg['end'], 'text': text, 'score': score,
                             'span': (max(0, end - len(keyword)), end)})
        return {'keyword': keyword, 'matches': hits, 'text': result['text']}


_models = {}


def get_model(lang=None, model=None, **kwargs):
    if isinstance(model, Whisper):
        return model
    key = model or 'base'
    if key not in _models:
        _models[key] = Whisper(pretrained=key, **kwargs)
    return _models[key]


def transcribe(audios, *, lang=None, model=None, **kwargs):
    """models/stt/__init__.py:45-46."""
    return get_model(lang=lang, model=model).predict(audios, lang=lang, **kwargs)


def translate(audios, *, lang=None, model=None, **kwargs):
    """models/stt/__init__.py:48-51."""
    return get_model(lang=lang, model=model).predict(audios, lang=lang, task='translate', **kwargs)


def stream(stream, *, lang=None, model=None, **kwargs):
    """models/stt/__init__.py:53-55."""
    return get_model(lang=lang, model=model).stream(stream, lang=lang, **kwargs)


def search(keyword, audios, *, lang=None, model=None, **kwargs):
    m = get_model(lang=lang, model=model)
    if not isinstance(audios, (list, tuple)):
        audios = [audios]
    return [m.search(keyword, a, lang=lang, **kwargs) for a in audios]
