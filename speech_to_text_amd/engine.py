"""Python handle on the native Whisper engine (stt_engine_* / stt_encode / stt_generate of
include/stt_mi355x.h).  torch only owns the device tensors handed in and out."""
import ctypes as C
import os

import numpy as np
import torch

from . import _lib

DIM_KEYS = ("n_mels", "n_audio_ctx", "n_audio_state", "n_audio_head", "n_audio_layer",
            "n_vocab", "n_text_ctx", "n_text_state", "n_text_head", "n_text_layer")

# OpenAI Whisper checkpoints ship these sizes (architectures/transformers/whisper_arch.py:181-189,
# 226-234 read them from `dims`)
WHISPER_DIMS = {
    "tiny": dict(n_mels=80, n_audio_state=384, n_audio_head=6, n_audio_layer=4, n_vocab=51865),
    "base": dict(n_mels=80, n_audio_state=512, n_audio_head=8, n_audio_layer=6, n_vocab=51865),
    "small": dict(n_mels=80, n_audio_state=768, n_audio_head=12, n_audio_layer=12, n_vocab=51865),
    "medium": dict(n_mels=80, n_audio_state=1024, n_audio_head=16, n_audio_layer=24, n_vocab=51865),
    "large-v3": dict(n_mels=128, n_audio_state=1280, n_audio_head=20, n_audio_layer=32, n_vocab=51866),
}


def whisper_dims(name):
    d = dict(WHISPER_DIMS[name])
    d.update(n_audio_ctx=1500, n_text_ctx=448, n_text_state=d["n_audio_state"],
             n_text_head=d["n_audio_head"], n_text_layer=d["n_audio_layer"])
    return d


class ModelConfig(C.Structure):
    _fields_ = [(k, C.c_int32) for k in DIM_KEYS] + [("gelu_tanh", C.c_int32)]


class GenerateParams(C.Structure):
    _fields_ = [
        ("method", C.c_int32), ("num_beams", C.c_int32), ("num_sentences", C.c_int32),
        ("max_steps", C.c_int32), ("eos_token", C.c_int32), ("pad_token", C.c_int32),
        ("early_stopping", C.c_int32), ("temperature", C.c_float),
        ("has_length_temperature", C.c_int32), ("length_temperature", C.c_float),
        ("has_length_power", C.c_int32), ("length_power", C.c_float),
        ("suppress", C.POINTER(C.c_int32)), ("n_suppress", C.c_int32),
        ("timestamp_filter", C.c_int32), ("suppress_first", C.POINTER(C.c_int32)),
        ("n_suppress_first", C.c_int32), ("timestamp_begin", C.c_int32),
        ("max_initial_timestamp", C.c_int32), ("check_every", C.c_int32),
        ("sample", C.c_int32), ("seed", C.c_uint64),
        ("max_steps_per_window", C.POINTER(C.c_int32)), ("d_logits_out", C.c_void_p),
    ]


class TimestampFilter(object):
    """Arguments of the reference's opt-in `logits_filter` (models/stt/whisper.py:354-416)."""

    def __init__(self, remove_tokens, remove_tokens_with_space, timestamp_begin_idx, eos_token_idx,
                 max_initial_timestamp=1):
        self.remove_tokens = np.asarray(remove_tokens, np.int32)
        self.remove_tokens_with_space = np.asarray(remove_tokens_with_space, np.int32)
        self.timestamp_begin_idx = int(timestamp_begin_idx)
        self.eos_token_idx = int(eos_token_idx)
        self.max_initial_timestamp = int(max_initial_timestamp)

    def as_callable(self):
        """The same rules as a function of torch tensors, for the loop that selects in Python
        (eager.py): `f(scores [n, V], tokens = generated [n, steps], state = LoopState)`.  Written as
        whole-tensor masks, so nothing is read back to the host; restates
        models/stt/whisper.py:354-416."""
        first = torch.from_numpy(self.remove_tokens_with_space.astype(np.int64))
        later = torch.from_numpy(self.remove_tokens.astype(np.int64))
        begin, eos, max_init = self.timestamp_begin_idx, self.eos_token_idx, self.max_initial_timestamp

        def rules(scores, tokens, state, **_):
            ninf = float("-inf")
            ids = torch.arange(scores.shape[-1], device=scores.device)[None]
            if state.state is None:
                ban = ids < begin
                if max_init > 0:
                    ban = ban | (ids >= begin + max_init)
                return scores.index_fill(1, first.to(scores.device), ninf).masked_fill(ban, ninf)
            t = int(state.t)
            scores = scores.index_fill(1, later.to(scores.device), ninf)
            last = tokens[:, t - 1] >= begin
            before = tokens[:, t - 2] >= begin if t >= 2 else torch.ones_like(last)
            scores = scores.masked_fill((last & ~before)[:, None] & (ids < eos), ninf)
            scores = scores.masked_fill((last & before)[:, None] & (ids >= begin), ninf)
            logp = torch.log_softmax(scores, dim=-1)
            stamp = torch.logsumexp(logp[:, begin:], dim=-1) > logp[:, :begin].max(dim=-1).values
            return scores.masked_fill(stamp[:, None] & (ids < begin), ninf)
        return rules

    @classmethod
    def from_callable(cls, fn):
        """The reference enables its timestamp rules by handing `generation_utils.infer` a callable
        (architectures/generation_utils.py:518-520): the free function `logits_filter(self, scores,
        tokens, state)` of models/stt/whisper.py:414-416 bound to the model, as a
        `functools.partial` or a bound method (or `timestamp_filter`, :354-412, with `to_remove`
        given).  Such a callable is recognised by its name and translated from the model object it
        carries (`remove_tokens`, `remove_tokens_with_space`, `timestamp_begin_idx`,
        `eos_token_idx`); the rules themselves run on the device.  Anything else returns None:
        arbitrary Python cannot run inside the device loop."""
        import functools
        owner, extra, func = None, {}, fn
        if isinstance(fn, functools.partial):
            func, extra = fn.func, dict(fn.keywords or {})
            owner = fn.args[0] if fn.args else extra.pop('self', None)
        elif getattr(fn, '__self__', None) is not None and hasattr(fn, '__func__'):
            func, owner = fn.__func__, fn.__self__
        name = getattr(func, '__name__', '')
        if owner is None or name not in ('logits_filter', 'timestamp_filter'):
            return None
        try:
            begin, eos = int(owner.timestamp_begin_idx), int(owner.eos_token_idx)
            if name == 'timestamp_filter':
                later = first = np.asarray(extra['to_remove'], np.int32).reshape(-1)
            else:
                later = np.asarray(owner.remove_tokens, np.int32).reshape(-1)
                first = np.asarray(owner.remove_tokens_with_space, np.int32).reshape(-1)
        except (AttributeError, KeyError):
            return None
        return cls(later, first, begin, eos, max_initial_timestamp=int(extra.get('max_initial_timestamp', 1)))


class WhisperEngine(object):
    def __init__(self, dims, weights, max_batch=1, max_beams=1, max_text_len=256, max_prompt_len=4,
                 gelu_tanh=True, device=None):
        self._lib = _lib.load()
        _declare(self._lib)
        self.dims = {k: int(dims[k]) for k in DIM_KEYS}
        self.device = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
        self.max_batch, self.max_beams = int(max_batch), int(max_beams)
        self.max_text_len, self.max_prompt_len = int(max_text_len), int(max_prompt_len)
        cfg = ModelConfig(gelu_tanh=1 if gelu_tanh else 0, **self.dims)
        self._h = C.c_void_p()
        with torch.cuda.device(self.device):
            _lib.check(self._lib.stt_engine_create(C.byref(cfg), self.max_batch, self.max_beams,
                                                   self.max_text_len, self.max_prompt_len,
                                                   C.byref(self._h)))
            for name, value in weights.items():
                arr = value.detach().cpu().float().numpy() if isinstance(value, torch.Tensor) else value
                arr = np.ascontiguousarray(arr, dtype=np.float32)
                _lib.check(self._lib.stt_engine_set_weight(
                    self._h, name.encode(), arr.ctypes.data_as(C.POINTER(C.c_float)), arr.size))
            _lib.check(self._lib.stt_engine_finalize(self._h))
        self._batch = 0
        self._epoch = 0          # bumped by every call that changes the decoder state (see eager.EngineState)

    def __del__(self):
        try:
            if self._h:
                self._lib.stt_engine_destroy(self._h)
                self._h = None
        except Exception:
            pass

    def reserve(self, max_batch=0, max_beams=0, max_text_len=0, max_prompt_len=0):
        """Grow the capacity (weights stay where they are); the encoded batch is dropped.  When the
        new buffers do not fit the card the call raises and the engine is back at its previous
        capacity (or, if even that no longer fits, unusable until a reserve succeeds: `usable()`)."""
        self._batch = 0
        self._epoch += 1
        with torch.cuda.device(self.device):
            _lib.check(self._lib.stt_engine_reserve(self._h, int(max_batch), int(max_beams),
                                                    int(max_text_len), int(max_prompt_len)))
        self.max_batch, self.max_beams = max(self.max_batch, int(max_batch)), max(self.max_beams, int(max_beams))
        self.max_text_len = max(self.max_text_len, int(max_text_len))
        self.max_prompt_len = max(self.max_prompt_len, int(max_prompt_len))
        self._batch = 0

    def usable(self):
        """False after an allocation failure that left the engine without buffers."""
        return bool(self._lib.stt_engine_usable(self._h))

    def probe_kernel(self, kernel=0, layer=0):
        """One launch of a hot kernel on the current stream (stt_engine_probe_kernel): for timing."""
        with torch.cuda.device(self.device):
            _lib.check(self._lib.stt_engine_probe_kernel(self._h, int(kernel), int(layer), self._stream()))

    def _stream(self):
        return torch.cuda.current_stream(self.device).cuda_stream

    def encode(self, mel, return_output=False):
        """mel: [B, frames <= 3000, n_mels] float32 (numpy or tensor).  Keeps the encoder state in
        the engine; optionally returns the encoder output [B, 1500, d] (float32 CUDA tensor)."""
        if not isinstance(mel, torch.Tensor):
            mel = torch.from_numpy(np.ascontiguousarray(np.asarray(mel, dtype=np.float32)))
        mel = mel.to(device=self.device, dtype=torch.float32).contiguous()
        if mel.dim() == 2:
            mel = mel[None]
        b, frames, m = mel.shape
        if m != self.dims["n_mels"]:
            raise ValueError("expected %d mel channels, got %d" % (self.dims["n_mels"], m))
        if frames > 3000:
            mel = mel[:, :3000].contiguous()
            frames = 3000
        out = None
        self._epoch += 1
        with torch.cuda.device(self.device):
            if return_output:
                out = torch.empty((b, 1500, self.dims["n_audio_state"]), dtype=torch.float32, device=self.device)
            _lib.check(self._lib.stt_encode(self._h, mel.data_ptr(), b, frames,
                                            out.data_ptr() if out is not None else None, self._stream()))
        self._batch = b
        return out

    def decoder_logits(self, tokens):
        """Teacher-forced logits [B, S, V] for tokens [B, S] over the last encoded batch."""
        tokens = np.ascontiguousarray(np.asarray(tokens, dtype=np.int32))
        if tokens.ndim == 1:
            tokens = tokens[None]
        b, s = tokens.shape
        with torch.cuda.device(self.device):
            out = torch.empty((b, s, self.dims["n_vocab"]), dtype=torch.float32, device=self.device)
            _lib.check(self._lib.stt_decoder_forward(
                self._h, tokens.ctypes.data_as(C.POINTER(C.c_int32)), b, s, out.data_ptr(), self._stream()))
        return out

    def decode_step(self, tokens, seq_per_window=1, reset=False, parents=None, attention=None):
        """One cached decoder step (stt_decode_step): `tokens` int32 [n_seq, n_tokens] (CUDA tensor or
        array) are appended to the sequences' histories and the logits of the last position come
        back as a float32 CUDA tensor [n_seq, V].  `reset` starts the histories afresh (the prompt);
        `parents` int32 [n_seq] re-links sequence i to the history of sequence parents[i] first
        (beam search).  Selection is the caller's: see speech_to_text_amd/eager.py.
        `attention`: None, 'last', 'all' or a first decoder layer: the call then returns
        (logits, self_attention [layers, n_seq, heads, n_tokens, positions so far],
        cross_attention [layers, n_seq, heads, n_tokens, 1500]) for those layers
        (stt_decode_step_attention)."""
        if not isinstance(tokens, torch.Tensor):
            tokens = torch.from_numpy(np.ascontiguousarray(np.asarray(tokens, dtype=np.int32)))
        tokens = tokens.to(device=self.device, dtype=torch.int32).contiguous()
        if tokens.dim() == 1:
            tokens = tokens[:, None]
        n_seq, n_tok = tokens.shape
        if parents is not None:
            if not isinstance(parents, torch.Tensor):
                parents = torch.from_numpy(np.ascontiguousarray(np.asarray(parents, dtype=np.int32)))
            parents = parents.to(device=self.device, dtype=torch.int32).contiguous()
            if parents.numel() != n_seq:
                raise ValueError("parents must name one sequence per sequence")
        self._epoch += 1
        with torch.cuda.device(self.device):
            out = torch.empty((n_seq, self.dims["n_vocab"]), dtype=torch.float32, device=self.device)
            if attention is None:
                _lib.check(self._lib.stt_decode_step(
                    self._h, tokens.data_ptr(), n_seq, int(seq_per_window), n_tok, 1 if reset else 0,
                    parents.data_ptr() if parents is not None else None, out.data_ptr(), self._stream()))
                return out
            n_layer, heads = self.dims["n_text_layer"], self.dims["n_text_head"]
            first = {"all": 0, "last": n_layer - 1}.get(attention, attention)
            if not isinstance(first, int) or not 0 <= first < n_layer:
                raise ValueError("attention must be None, 'last', 'all' or a decoder layer, got {!r}".format(attention))
            total = (0 if reset else self.step_state()[2]) + n_tok
            own = torch.empty((n_layer - first, n_seq, heads, n_tok, total), dtype=torch.float32, device=self.device)
            enc = torch.empty((n_layer - first, n_seq, heads, n_tok, 1500), dtype=torch.float32, device=self.device)
            _lib.check(self._lib.stt_decode_step_attention(
                self._h, tokens.data_ptr(), n_seq, int(seq_per_window), n_tok, 1 if reset else 0,
                parents.data_ptr() if parents is not None else None, out.data_ptr(), first,
                own.data_ptr(), enc.data_ptr(), self._stream()))
        return out, own, enc

    def step_state(self):
        """(sequences, sequences per window, cached positions) of the current decode_step history."""
        n, per, length = C.c_int(0), C.c_int(0), C.c_int(0)
        _lib.check(self._lib.stt_engine_step_state(self._h, C.byref(n), C.byref(per), C.byref(length)))
        return n.value, per.value, length.value

    def export_state(self, layer):
        """((k, v), (enc_k, enc_v)) of decoder layer `layer`, float32 CUDA tensors [n, heads, length,
        64]: the self-attention cache of the decode_step history and the encoder-attention K/V of
        the encoded batch - the layer's entry of the state the reference's loop returns
        (architectures/transformers/transformer_arch.py:440-475)."""
        n, _, length = self.step_state()
        heads = self.dims["n_text_head"]
        with torch.cuda.device(self.device):
            k = torch.empty((n, heads, length, 64), dtype=torch.float32, device=self.device)
            v = torch.empty_like(k)
            ek = torch.empty((self._batch, heads, 1500, 64), dtype=torch.float32, device=self.device)
            ev = torch.empty_like(ek)
            _lib.check(self._lib.stt_engine_export_state(self._h, int(layer), 0, k.data_ptr(), v.data_ptr(), self._stream()))
            _lib.check(self._lib.stt_engine_export_state(self._h, int(layer), 1, ek.data_ptr(), ev.data_ptr(), self._stream()))
        return (k, v), (ek, ev)

    def generate(self, prompt, *, method="greedy", num_beams=1, num_sentences=1, max_new_tokens,
                 eos_token, pad_token, early_stopping=True, temperature=None, length_temperature=None,
                 length_power=None, logits_filter=None, check_every=8, sample=None, seed=None,
                 return_logits=False, filter_kwargs=None, max_new_tokens_per_window=None,
                 return_attention=False, return_last_attention=False, return_state=False,
                 initial_state=None, extras=None):
        """`temperature` given (and not 0) means, as in the reference (generation_utils.py:538-548),
        that tokens are drawn from the categorical distribution of the processed scores instead of
        taking the argmax / top-k; `sample = False` keeps the rescaling but not the draw.  `seed`
        makes the draws repeatable (default: fresh entropy per call, like the reference's
        unseeded `keras.random.categorical`).  `return_logits` appends a CUDA tensor
        [batch, (num_sentences,) max_new_tokens, n_vocab] with the processed scores of every step to
        the result (generation_utils.py:266-268, 297, 422-428); rows of steps after an early stop are
        -inf.  `logits_filter`: a list of token ids, a TimestampFilter, the reference's Whisper filter
        bound to a model (all three run inside the device loop), or any other callable
        `f(scores, tokens = ..., state = ..., **filter_kwargs)` on CUDA tensors, which - like
        `return_logits` under beam search - runs the loop in Python over `decode_step` (eager.py).
        `max_new_tokens_per_window` [batch]: a window ends after that many new tokens, exactly as a call
        of its own with that `max_new_tokens` would (device loop only); with early stopping on, windows
        that have ended - by EOS or by their limit - cost no attention traffic any more.
        `return_attention` / `return_last_attention` / `return_state` / `initial_state`: the reference's
        options of the same names (generation_utils.py:84-87, 293-300); they run the loop in Python
        too and leave their results in the dict passed as `extras` ('attention_weights', 'state')."""
        if return_attention or return_last_attention or return_state or initial_state is not None:
            from . import eager
            if callable(logits_filter) and not isinstance(logits_filter, TimestampFilter):
                logits_filter = TimestampFilter.from_callable(logits_filter) or logits_filter
            if isinstance(logits_filter, TimestampFilter):
                logits_filter = logits_filter.as_callable()
            return eager.generate(
                self, prompt, method=method, num_beams=num_beams, num_sentences=num_sentences,
                max_new_tokens=max_new_tokens, eos_token=eos_token, pad_token=pad_token,
                early_stopping=early_stopping, temperature=temperature,
                length_temperature=length_temperature, length_power=length_power,
                logits_filter=logits_filter, return_logits=return_logits, seed=seed, check_every=check_every,
                filter_kwargs=filter_kwargs, return_attention=return_attention,
                return_last_attention=return_last_attention, return_state=return_state,
                initial_state=initial_state, extras=extras)
        prompt = np.ascontiguousarray(np.asarray(prompt, dtype=np.int32))
        if prompt.ndim == 1:
            prompt = prompt[None]
        b, s = prompt.shape
        beam = method in ("beam", "beam_search")
        if not beam and method not in ("greedy", "sample"):      # generation_utils.py:826-831
            raise ValueError("unsupported inference method: {}".format(method))
        nsent = int(num_sentences) if beam else 1
        p = GenerateParams()
        p.method = 1 if beam else 0
        p.num_beams = int(num_beams) if beam else 1
        p.num_sentences = nsent
        p.max_steps = int(max_new_tokens)
        p.eos_token, p.pad_token = int(eos_token), int(pad_token)
        p.early_stopping = 1 if early_stopping else 0
        p.temperature = float(temperature) if temperature else 0.0
        p.has_length_temperature = 0 if length_temperature is None else 1
        p.length_temperature = float(length_temperature or 0.0)
        p.has_length_power = 0 if length_power is None else 1
        p.length_power = float(length_power or 0.0)
        p.check_every = int(check_every)
        p.sample = int(bool(temperature) if sample is None else bool(sample))
        if p.sample:
            p.seed = int(seed) & (2 ** 64 - 1) if seed is not None else int.from_bytes(os.urandom(8), "little")
        keep = []
        if callable(logits_filter) and not isinstance(logits_filter, TimestampFilter):
            # the reference's own Whisper filter is recognised and runs inside the device loop; any
            # other callable gets the scores of every step as a CUDA tensor (eager.py)
            translated = TimestampFilter.from_callable(logits_filter)
            if translated is None:
                from . import eager
                return eager.generate(
                    self, prompt, method=method, num_beams=num_beams, num_sentences=num_sentences,
                    max_new_tokens=max_new_tokens, eos_token=eos_token, pad_token=pad_token,
                    early_stopping=early_stopping, temperature=temperature,
                    length_temperature=length_temperature, length_power=length_power,
                    logits_filter=logits_filter, return_logits=return_logits, seed=seed,
                    check_every=check_every, filter_kwargs=filter_kwargs)
            logits_filter = translated
        if return_logits and beam:
            # per-step scores gathered by beam index (generation_utils.py:422-428): eager loop,
            # with the timestamp rules as a torch function when they are asked for
            from . import eager
            if isinstance(logits_filter, TimestampFilter):
                logits_filter = logits_filter.as_callable()
            return eager.generate(
                self, prompt, method=method, num_beams=num_beams, num_sentences=num_sentences,
                max_new_tokens=max_new_tokens, eos_token=eos_token, pad_token=pad_token,
                early_stopping=early_stopping, temperature=temperature,
                length_temperature=length_temperature, length_power=length_power,
                logits_filter=logits_filter, return_logits=True, seed=seed, check_every=check_every,
                filter_kwargs=filter_kwargs)
        if isinstance(logits_filter, TimestampFilter):
            a, bb = logits_filter.remove_tokens, logits_filter.remove_tokens_with_space
            keep += [a, bb]
            p.timestamp_filter = 1
            p.suppress, p.n_suppress = a.ctypes.data_as(C.POINTER(C.c_int32)), a.size
            p.suppress_first, p.n_suppress_first = bb.ctypes.data_as(C.POINTER(C.c_int32)), bb.size
            p.timestamp_begin = logits_filter.timestamp_begin_idx
            p.max_initial_timestamp = logits_filter.max_initial_timestamp
            p.eos_token = int(eos_token)
        elif logits_filter is not None:
            a = np.ascontiguousarray(np.asarray(logits_filter, dtype=np.int32).reshape(-1))
            keep.append(a)
            p.suppress, p.n_suppress = a.ctypes.data_as(C.POINTER(C.c_int32)), a.size
        if max_new_tokens_per_window is not None:
            caps = np.ascontiguousarray(np.asarray(max_new_tokens_per_window, dtype=np.int32).reshape(-1))
            if caps.size != b:
                raise ValueError("max_new_tokens_per_window: one value per window")
            keep.append(caps)
            p.max_steps_per_window = caps.ctypes.data_as(C.POINTER(C.c_int32))
        tokens = np.empty((b, nsent, p.max_steps), dtype=np.int32)
        lengths = np.empty((b, nsent), dtype=np.int32)
        scores = np.empty((b, nsent), dtype=np.float32)
        logits = None
        self._epoch += 1
        with torch.cuda.device(self.device):
            if return_logits:
                logits = torch.full((b, p.max_steps, self.dims["n_vocab"]), float("-inf"),
                                    dtype=torch.float32, device=self.device)
                p.d_logits_out = logits.data_ptr()
            _lib.check(self._lib.stt_generate(
                self._h, C.byref(p), prompt.ctypes.data_as(C.POINTER(C.c_int32)), b, s,
                tokens.ctypes.data_as(C.POINTER(C.c_int32)), lengths.ctypes.data_as(C.POINTER(C.c_int32)),
                scores.ctypes.data_as(C.POINTER(C.c_float)), self._stream()))
        if not beam:
            if return_logits:
                return tokens[:, 0], lengths[:, 0], scores[:, 0], logits
            return tokens[:, 0], lengths[:, 0], scores[:, 0]
        return tokens, lengths, scores


_declared = False


def _declare(lib):
    global _declared
    if _declared:
        return
    vp, i32, i64 = C.c_void_p, C.c_int, C.c_int64
    lib.stt_engine_create.restype = i32
    lib.stt_engine_create.argtypes = [C.POINTER(ModelConfig), i32, i32, i32, i32, C.POINTER(vp)]
    lib.stt_engine_destroy.restype = None
    lib.stt_engine_destroy.argtypes = [vp]
    lib.stt_engine_set_weight.restype = i32
    lib.stt_engine_set_weight.argtypes = [vp, C.c_char_p, C.POINTER(C.c_float), i64]
    lib.stt_engine_finalize.restype = i32
    lib.stt_engine_finalize.argtypes = [vp]
    lib.stt_engine_reserve.restype = i32
    lib.stt_engine_reserve.argtypes = [vp, i32, i32, i32, i32]
    lib.stt_engine_usable.restype = i32
    lib.stt_engine_usable.argtypes = [vp]
    lib.stt_decode_step.restype = i32
    lib.stt_decode_step.argtypes = [vp, vp, i32, i32, i32, i32, vp, vp, vp]
    lib.stt_decode_step_attention.restype = i32
    lib.stt_decode_step_attention.argtypes = [vp, vp, i32, i32, i32, i32, vp, vp, i32, vp, vp, vp]
    lib.stt_engine_export_state.restype = i32
    lib.stt_engine_export_state.argtypes = [vp, i32, i32, vp, vp, vp]
    lib.stt_engine_step_state.restype = i32
    lib.stt_engine_step_state.argtypes = [vp, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]
    lib.stt_engine_probe_kernel.restype = i32
    lib.stt_engine_probe_kernel.argtypes = [vp, i32, i32, vp]
    lib.stt_encode.restype = i32
    lib.stt_encode.argtypes = [vp, vp, i32, i32, vp, vp]
    lib.stt_decoder_forward.restype = i32
    lib.stt_decoder_forward.argtypes = [vp, C.POINTER(C.c_int32), i32, i32, vp, vp]
    lib.stt_generate.restype = i32
    lib.stt_generate.argtypes = [vp, C.POINTER(GenerateParams), C.POINTER(C.c_int32), i32, i32,
                                 C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_float), vp]
    _declared = True


ENGINE_SYMBOLS = ("stt_engine_create", "stt_engine_destroy", "stt_engine_set_weight",
                  "stt_engine_finalize", "stt_engine_reserve", "stt_engine_usable", "stt_engine_probe_kernel", "stt_decode_step",
                  "stt_decode_step_attention", "stt_engine_export_state", "stt_engine_step_state", "stt_encode", "stt_decoder_forward", "stt_generate")
