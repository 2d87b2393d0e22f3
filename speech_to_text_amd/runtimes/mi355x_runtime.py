"""``Runtime`` implementation that runs Whisper on the MI355X engine.

It satisfies what ``models/stt/whisper.py`` asks of ``self.model`` when a runtime is plugged in
through ``BaseModel(runtime = ...)`` (models/interfaces/base_model.py:179-189, 365-374):

* ``__call__(mel, tokens = ..., tokens_length = ..., max_length = ... | max_new_tokens = ...,
  method = ..., num_beams = ..., ...)`` returning an object with ``tokens`` / ``lengths`` /
  ``scores`` that ``process_model_output`` accepts (whisper.py:247-251);
* ``set_tokens`` (models/stt/base_stt.py:68), ``is_enc_dec`` and a non-None ``decoder``
  (models/interfaces/base_text_model.py:45-48);
* ``encoder(mel)`` / ``decoder(tokens, encoder_output = ...)`` for language detection
  (whisper.py:174-186).

``path`` is a directory or file holding the weights: ``*.safetensors`` / ``*.pt`` / ``*.npz`` with
OpenAI or Hugging Face Whisper names, a model directory the reference wrote (``saving/*.weights.h5``
+ ``saving/config_models.json``, read in-process), or the string ``random:<size>[:seed[:plain|peaked]]`` for seeded
synthetic weights (no checkpoint can be downloaded offline).
"""
import collections
import os
import threading

import numpy as np
import torch

from .runtime import Runtime
from ..engine import WhisperEngine, TimestampFilter, whisper_dims
from .. import weights as W
from .. import keras_checkpoint as kc

InferenceOutput = collections.namedtuple(
    "InferenceOutput", ["tokens", "lengths", "scores", "logits", "state", "attention_weights"]
)


def _load_state_dict(path):
    return _load_weights(path)[0]


def _load_weights(path):
    """(OpenAI-named tensors, GELU form or None) of a weight file, a plain weight directory, or a
    model directory written by the reference (pretrained_models/<name>/{config.json, saving/...},
    models/interfaces/base_model.py:126-136): there the newest `saving/*.weights.h5` is read
    in-process (keras_checkpoint.py) and `config_models.json` says which GELU the model uses."""
    if os.path.isdir(path):
        for folder in (path, os.path.join(path, 'saving')):
            if not os.path.isdir(folder):
                continue
            for name in sorted(os.listdir(folder)):
                if name.endswith(('.safetensors', '.pt', '.pth', '.npz')):
                    return _load_weights(os.path.join(folder, name))[0], kc.model_config(path)['gelu_tanh']
        found = kc.load_model_dir(path)
        if found is None:
            raise FileNotFoundError('no weight file (*.safetensors, *.pt, *.npz, *.weights.h5) in {}'.format(path))
        weights, config = found
        return {k: torch.from_numpy(v) for k, v in weights.items()}, config['gelu_tanh']
    if path.endswith('.weights.h5'):
        weights = kc.to_openai(kc.read_vars(path))
        return {k: torch.from_numpy(v) for k, v in weights.items()}, kc.model_config(os.path.dirname(path))['gelu_tanh']
    return _load_file(path), None


def _load_file(path):
    if path.endswith('.safetensors'):
        from safetensors.torch import load_file
        sd = load_file(path)
    elif path.endswith('.npz'):
        with np.load(path, allow_pickle=False) as f:
            sd = {k: torch.from_numpy(f[k]) for k in f.files}
    else:
        sd = torch.load(path, map_location='cpu', weights_only=True)
        if 'model_state_dict' in sd:
            sd = sd['model_state_dict']
    if any(k.startswith('model.encoder.') for k in sd):
        return W.from_hf_state_dict(sd)
    return W.from_openai_state_dict(sd)


class _Engine(object):
    """What ``load_engine`` caches per path: dims + weights; device engines are built lazily for
    the capacity a call needs."""

    def __init__(self, dims, weights, gelu_tanh=True):
        self.dims, self.weights, self.gelu_tanh = dims, weights, gelu_tanh
        self._built = {}      # replica -> WhisperEngine
        self._locks = collections.defaultdict(threading.RLock)
        self._guard = threading.Lock()

    def lock(self, replica=0):
        """Calls on one engine must not overlap (the reference's `Stream` may call a runtime from
        a worker thread, utils/stream.py:161-167): every use of a replica happens under its lock."""
        with self._guard:
            return self._locks[replica]

    # Up to this many sequences (windows x beams) the engine is sized for the text side's worst
    # case; beyond it from the calls it has seen (rounded up), because the self-attention cache
    # grows with sequences x positions: large-v3 needs 73 MB per sequence at 448 positions, i.e.
    # 37 GB for 512 greedy windows against 11 GB for the 132 positions such a call really uses.
    WORST_CASE_SEQUENCES = 8

    def get(self, batch, beams, text_len, prompt_len, replica=0):
        """Engine `replica` (0 unless a caller pipelines batches over several engines).

        Few sequences (the reference's own way of calling: one window, or one window x a few
        beams): built once for every position of the decoder context and a prompt of half of it
        plus the task tokens, which is what the long-form loop grows to window by window
        (models/stt/whisper.py:238-243) - so neither a longer prompt nor `encoder()` followed by
        `decoder()` ever re-allocates.  Batched calls: sized from the call, text positions rounded
        up to 64 and prompt tokens to 8.  Whatever grows later goes through stt_engine_reserve:
        buffers are re-allocated (the encoded batch is lost), the weights stay on the device."""
        ctx = min(int(self.dims['n_text_ctx']), 512)
        if batch * beams <= self.WORST_CASE_SEQUENCES:
            text_len = ctx
            prompt_len = max(prompt_len, min(ctx - 1, ctx // 2 + 3))
        else:
            text_len = min(ctx, (max(text_len, 2) + 63) // 64 * 64)
            prompt_len = min(text_len - 1, (max(prompt_len, 1) + 7) // 8 * 8)
        eng = self._built.get(replica)
        if eng is None:
            eng = WhisperEngine(self.dims, self.weights, max_batch=batch, max_beams=beams,
                                max_text_len=text_len, max_prompt_len=prompt_len, gelu_tanh=self.gelu_tanh)
            self._built[replica] = eng
        elif (batch > eng.max_batch or beams > eng.max_beams or text_len > eng.max_text_len
              or prompt_len > eng.max_prompt_len):
            try:
                eng.reserve(batch, beams, text_len, prompt_len)
            except Exception:
                # out of memory: the native engine went back to the capacity that fitted, or
                # holds no buffers at all - in both cases its encoded batch is gone
                eng._batch = 0
                if not eng.usable():
                    del self._built[replica]
                raise
        return eng


class MI355XRuntime(Runtime):
    _engines = {}

    def __init__(self, path, *, max_batch=1, **kwargs):
        super().__init__(path, **kwargs)
        self.is_enc_dec = True
        self.sos_token = -1
        self.eos_token = -1
        self.pad_token = -1
        self._max_batch = max_batch

    # -- Runtime interface -------------------------------------------------
    @staticmethod
    def load_engine(path, *, gelu=None, **kwargs):
        """`gelu`: 'tanh' or 'erf'; by default what the model directory's `config_models.json` states,
        else 'tanh' (the form the name 'gelu' resolves to in Keras 3, which the reference runs)."""
        if isinstance(path, str) and path.startswith('random:'):
            parts = path.split(':')
            dims = whisper_dims(parts[1])
            seed = int(parts[2]) if len(parts) > 2 else 0
            weights = W.random_whisper_weights(dims, seed, style=parts[3] if len(parts) > 3 else "plain")
            stated = None
        else:
            weights, stated = _load_weights(path)
            dims = W.dims_from_openai_state_dict(weights)
        if gelu is None:
            gelu = 'tanh' if stated in (None, True) else 'erf'
        elif gelu not in ('tanh', 'erf'):
            raise ValueError("gelu must be 'tanh' or 'erf', got {!r}".format(gelu))
        return _Engine(dims, weights, gelu_tanh=(gelu == 'tanh'))

    def set_tokens(self, sos_token=None, eos_token=None, pad_token=None):
        if sos_token not in (-1, None): self.sos_token = sos_token
        if eos_token not in (-1, None): self.eos_token = eos_token
        if pad_token not in (-1, None): self.pad_token = pad_token

    @property
    def dims(self):
        return self.engine.dims

    @property
    def vocab_size(self):
        return self.engine.dims['n_vocab']

    def __call__(self, inputs, *, tokens=None, tokens_length=None, encoder_output_lengths=None,
                 max_length=None, max_new_tokens=None, method='greedy', num_beams=10,
                 num_sentences=1, temperature=None, length_temperature=None, length_power=None,
                 logits_filter=None, early_stopping=True, seed=None, replica=0, on_encoded=None,
                 return_logits=False, return_state=False, return_attention=False,
                 return_last_attention=False, initial_state=None, **other):
        """inputs: [B, <=3000, n_mels] mel windows; tokens: [B, S] prompt ids.  Unknown keyword
        arguments (lang, verbose, ...) are ignored, as the reference's runtimes do.
        A `temperature` switches token selection from argmax / top-k to categorical draws, as in
        the reference (generation_utils.py:538-548); `seed` makes the draws repeatable.
        `replica` selects one of several engines built from the same weights and `on_encoded(event)`
        is called once the encoder pass is enqueued (the event completes when it has run): both
        exist so that a caller can decode one batch while another engine encodes the next.
        `return_logits` fills `InferenceOutput.logits` with the processed scores of every step,
        [B, (num_sentences,) steps, V] on the GPU (generation_utils.py:266-268, 297, 422-428).
        `logits_filter` may be any callable `f(scores, tokens = ..., state = ..., **kwargs)`: the
        reference's own Whisper filter runs inside the device loop, anything else (and
        `return_logits` under beam search) runs the loop in Python, one cached decoder step per token,
        with the scores as CUDA tensors and the remaining keyword arguments of this call passed on
        to the filter as the reference does (generation_utils.py:360-366, 518-520).
        `return_attention` / `return_last_attention` fill `InferenceOutput.attention_weights` with the
        dict {'attn_layer_<i>', 'enc_attn_layer_<i>'} of every / the last decoder layer, `return_state`
        fills `.state` with a mapping layer name -> ((k, v), (enc_k, enc_v)) (eager.EngineState), and
        `initial_state` = such a state, untouched since, continues from it: `inputs` may then be None
        (the windows are the ones already encoded) and `tokens` are only the tokens that follow
        (generation_utils.py:84-87, 186, 293-300).  These run the loop in Python as well."""
        if initial_state is not None:
            return self._continue(initial_state, tokens, tokens_length, max_length, max_new_tokens, method,
                                  temperature, length_temperature, logits_filter, early_stopping, seed,
                                  return_logits, return_state, return_attention, return_last_attention, other)
        if isinstance(inputs, torch.Tensor):
            mel = inputs
        else:
            mel = torch.from_numpy(np.ascontiguousarray(np.asarray(inputs, dtype=np.float32)))
        if mel.dim() == 2:
            mel = mel[None]
        b = mel.shape[0]
        if tokens is None:
            if self.sos_token in (-1, None):
                raise ValueError('no prompt given and no sos token set')
            tokens = np.full((b, 1), self.sos_token, dtype=np.int32)
        tokens = np.asarray(tokens.cpu() if isinstance(tokens, torch.Tensor) else tokens, dtype=np.int32)
        if tokens.ndim == 1:
            tokens = np.tile(tokens[None], [b, 1])
        s = int(tokens_length) if tokens_length is not None else tokens.shape[1]
        tokens = np.ascontiguousarray(tokens[:, :s])
        if self.eos_token in (-1, None):
            raise ValueError('call set_tokens(eos_token = ..., pad_token = ...) first')
        # generation_utils.py:794-801
        if max_length is None:
            max_length = s + (256 if max_new_tokens is None else int(max_new_tokens))
        max_steps = int(max_length) - s
        if max_steps < 1:
            raise ValueError('prompt of {} tokens leaves no room under max_length = {}'.format(s, max_length))
        beam = method in ('beam', 'beam_search')
        pad = self.pad_token if self.pad_token not in (-1, None) else self.eos_token
        with self.engine.lock(replica):
            eng = self.engine.get(max(b, self._max_batch), int(num_beams) if beam else 1, s + max_steps, s,
                                  replica=replica)
            self.engine._state_owner = None        # the encoded batch of an earlier encoder() call is gone
            extras = {}
            eng.encode(mel)
            if on_encoded is not None:
                ev = torch.cuda.Event()
                ev.record()
                on_encoded(ev)
            out = eng.generate(
                tokens, method=method, num_beams=num_beams, num_sentences=num_sentences,
                max_new_tokens=max_steps, eos_token=self.eos_token, pad_token=pad,
                early_stopping=early_stopping, temperature=temperature,
                length_temperature=length_temperature, length_power=length_power,
                logits_filter=logits_filter, seed=seed, return_logits=bool(return_logits),
                filter_kwargs=other, return_attention=bool(return_attention),
                return_last_attention=bool(return_last_attention), return_state=bool(return_state),
                extras=extras)
        toks, lengths, scores = out[:3]
        return InferenceOutput(tokens=toks, lengths=lengths, scores=scores,
                               logits=out[3] if return_logits else None, state=extras.get('state'),
                               attention_weights=extras.get('attention_weights'))

    def _continue(self, state, tokens, tokens_length, max_length, max_new_tokens, method, temperature,
                  length_temperature, logits_filter, early_stopping, seed, return_logits, return_state,
                  return_attention, return_last_attention, other):
        """A loop that starts from the state an earlier call returned (`initial_state`): nothing is
        encoded and no earlier token is run again; `max_length` counts the positions of the state."""
        from ..eager import EngineState
        if not isinstance(state, EngineState):
            raise ValueError('initial_state must be the `state` of an earlier InferenceOutput of this runtime')
        if tokens is None:
            raise ValueError('initial_state needs the tokens that follow it')
        tokens = np.asarray(tokens.cpu() if isinstance(tokens, torch.Tensor) else tokens, dtype=np.int32)
        if tokens.ndim == 1:
            tokens = np.tile(tokens[None], [state.n_seq, 1])
        s = int(tokens_length) if tokens_length is not None else tokens.shape[1]
        tokens = np.ascontiguousarray(tokens[:, :s])
        if max_length is None:
            max_length = state.length + s + (256 if max_new_tokens is None else int(max_new_tokens))
        max_steps = int(max_length) - s - state.length
        if max_steps < 1:
            raise ValueError('state of {} positions and {} new tokens leave no room under max_length = {}'.format(
                state.length, s, max_length))
        eng = state.engine
        if state.length + s + max_steps > eng.max_text_len:
            raise ValueError('the engine was sized for {} positions; ask for the whole length in the first call '
                             '(max_length)'.format(eng.max_text_len))
        pad = self.pad_token if self.pad_token not in (-1, None) else self.eos_token
        extras = {}
        with self.engine.lock():
            out = eng.generate(
                tokens, method=method, max_new_tokens=max_steps, eos_token=self.eos_token, pad_token=pad,
                early_stopping=early_stopping, temperature=temperature, length_temperature=length_temperature,
                logits_filter=logits_filter, seed=seed, return_logits=bool(return_logits), filter_kwargs=other,
                return_attention=bool(return_attention), return_last_attention=bool(return_last_attention),
                return_state=bool(return_state), initial_state=state, extras=extras)
        toks, lengths, scores = out[:3]
        return InferenceOutput(tokens=toks, lengths=lengths, scores=scores,
                               logits=out[3] if return_logits else None, state=extras.get('state'),
                               attention_weights=extras.get('attention_weights'))

    # -- callables used by detect_language --------------------------------
    def encoder(self, mel, training=False, **_):
        """Encoder output [B, 1500, d] (float32 CUDA tensor).  The engine also keeps the encoded
        batch (and its cross-attention K/V) for a following `decoder` call."""
        if not isinstance(mel, torch.Tensor):
            mel = torch.from_numpy(np.ascontiguousarray(np.asarray(mel, dtype=np.float32)))
        if mel.dim() == 2:
            mel = mel[None]
        with self.engine.lock():
            eng = self.engine.get(max(mel.shape[0], self._max_batch), 1, 8, 4)
            out = eng.encode(mel, return_output=True)
            self._mark_state_owner()
            self._encoded = (out.data_ptr(), tuple(out.shape), out._version, mel)
            return out

    def decoder(self, tokens, encoder_output=None, training=False, **_):
        """Logits [B, S, V] of a teacher-forced pass (whisper.py:181-183).  `encoder_output` must be
        what the last `encoder` call on this runtime returned, untouched: the engine decodes
        against the K/V it derived from it.  If another call has used the engine in between (the
        lock is not held across the two calls) the windows are encoded again first; a tensor the
        engine never produced is refused."""
        tokens = np.asarray(tokens.cpu() if isinstance(tokens, torch.Tensor) else tokens, dtype=np.int32)
        if tokens.ndim == 1:
            tokens = tokens[None]
        with self.engine.lock():
            eng = self.engine.get(max(tokens.shape[0], self._max_batch), 1, max(8, tokens.shape[1] + 1),
                                  max(4, tokens.shape[1]))
            if encoder_output is not None:
                known = getattr(self, '_encoded', None)
                same = (known is not None and isinstance(encoder_output, torch.Tensor)
                        and encoder_output.data_ptr() == known[0] and tuple(encoder_output.shape) == known[1]
                        and encoder_output._version == known[2])
                if not same:
                    raise ValueError('encoder_output is not the (unmodified) tensor the last encoder() call '
                                     'returned: the engine decodes against its own copy of the encoder state')
                # (growing the engine for a longer token list drops its encoded batch)
                if self._state_owner() is not self or eng._batch != known[1][0]:
                    eng.encode(known[3])
            self._mark_state_owner()
            return eng.decoder_logits(tokens)

    def _state_owner(self):
        return getattr(self.engine, '_state_owner', None)

    def _mark_state_owner(self):
        self.engine._state_owner = self
