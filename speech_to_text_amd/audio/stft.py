"""MI355X drop-in for the reference's ``WhisperSTFT`` (utils/audio/stft.py:317-360).

Same constructor arguments, ``__call__`` contract (1-D or [1, n] audio in, a
[1, frames, n_mel_channels] float32 mel out), ``get_config`` / ``save`` /
``MelSTFT.create`` behaviour.  The arithmetic runs in ``libstt_mi355x.so``;
torch is only used to own device memory and the stream.
"""
import json
import os
import threading

import numpy as np
import torch

from .. import _lib
from .mel_basis import hann_window, mel_filterbank


def _in_samples(value, sampling_rate):
    """Window / hop / filter lengths may be given in samples (> 1) or as a fraction of a second."""
    return int(value * sampling_rate) if value <= 1. else value


class MelSTFT(object):
    """What a mel frontend is configured with, under the attribute and file-key names of the
    reference's class of the same name (utils/audio/stft.py:23-75, 146-182), so that a `mel_fn.json`
    it saved builds the same object here and the other way round.  No arithmetic lives here: the
    subclass with a kernel behind it is `WhisperSTFT`."""

    # file keys, in the order they are written; each is also the attribute that holds the value
    _FIELDS = ('n_mel_channels', 'sampling_rate', 'win_length', 'hop_length', 'filter_length',
               'mel_fmin', 'mel_fmax', 'pre_emph', 'normalize_mode')
    _registry = {}

    def __init_subclass__(cls, **kwargs):
        super().__init_subclass__(**kwargs)
        MelSTFT._registry[cls.__name__] = cls

    def __init__(self, sampling_rate, n_mel_channels=80, *, win_length=1024, hop_length=256,
                 filter_length=1024, mel_fmin=0.0, mel_fmax=8000.0, normalize_mode=None,
                 pre_emph=0., **kwargs):
        if normalize_mode not in (None, 'per_feature', 'all_feature'):
            raise ValueError("normalize_mode must be None, 'per_feature' or 'all_feature', got {!r}".format(normalize_mode))
        self.sampling_rate, self.n_mel_channels = sampling_rate, n_mel_channels
        self.win_length = _in_samples(win_length, sampling_rate)
        self.hop_length = _in_samples(hop_length, sampling_rate)
        self.filter_length = _in_samples(filter_length, sampling_rate)
        self.mel_fmin, self.mel_fmax = mel_fmin, mel_fmax
        self.pre_emph, self.normalize_mode = pre_emph, normalize_mode
        # [1, filter_length // 2 + 1, n_mel_channels], the layout the reference keeps its basis in
        self.mel_basis = mel_filterbank(self.sampling_rate, self.filter_length, self.n_mel_channels,
                                        self.mel_fmin, self.mel_fmax)[None]

    rate = property(lambda self: self.sampling_rate)

    def get_mel_length(self, audio_length):
        """Frames of a clip of `audio_length` samples, clips shorter than one filter being padded
        to it (stft.py:122-124)."""
        return -(-max(audio_length, self.filter_length) // self.hop_length)

    def get_audio_length(self, mel_length):
        return self.hop_length * mel_length

    def get_config(self):
        config = {'class_name': type(self).__name__}
        config.update((key, getattr(self, key)) for key in self._FIELDS)
        return config

    def __eq__(self, other):
        return isinstance(other, MelSTFT) and other.get_config() == self.get_config()

    __hash__ = None

    def save(self, filename):
        """Writes the configuration as JSON (`.json` appended when missing); returns the path."""
        path = filename if filename.endswith('.json') else filename + '.json'
        with open(path, 'w') as f:
            json.dump(self.get_config(), f, indent=4)
        return path

    save_to_file = save

    @classmethod
    def load_from_file(cls, filename):
        return MelSTFT.create(filename)

    @staticmethod
    def create(class_name, *args, **kwargs):
        """`create('WhisperSTFT', ...)` builds that class; `create('<file>.json')` builds what the file
        describes (its `class_name` key names the class, the other keys are constructor arguments)."""
        known = MelSTFT._registry
        if class_name in known:
            return known[class_name](*args, **kwargs)
        if isinstance(class_name, str) and os.path.isfile(class_name):
            with open(class_name) as f:
                stored = json.load(f)
            return MelSTFT.create(**stored)
        raise ValueError("unknown mel frontend {!r}: not one of {} and not a configuration file".format(
            class_name, sorted(known)))


MelSTFT._registry['MelSTFT'] = MelSTFT


class WhisperSTFT(MelSTFT):
    """Whisper log-mel on the GPU.  Only the Whisper geometry (400 / 160 / 400, periodic
    Hann, no pre-emphasis) has a kernel; other settings raise at construction."""

    def __init__(self, sampling_rate=16000, n_mel_channels=80, *, win_length=400, hop_length=160,
                 filter_length=400, mel_fmin=0.0, mel_fmax=8000.0, window='hann', periodic=True,
                 device=None, **kwargs):
        super().__init__(sampling_rate=sampling_rate, n_mel_channels=n_mel_channels,
                         win_length=win_length, hop_length=hop_length, filter_length=filter_length,
                         mel_fmin=mel_fmin, mel_fmax=mel_fmax, **kwargs)
        if (self.win_length, self.hop_length, self.filter_length) != (400, 160, 400):
            raise ValueError("the MI355X frontend is built for win/hop/filter = 400/160/400, got "
                             "{}/{}/{}".format(self.win_length, self.hop_length, self.filter_length))
        if window != 'hann' or self.pre_emph or self.normalize_mode is not None:
            raise ValueError("only window='hann', pre_emph=0, normalize_mode=None are supported")
        self.window = window
        self.periodic = periodic
        self._window = hann_window(self.win_length, periodic).astype(np.float32)
        self._device = torch.device(device) if device is not None else None
        self._plans = {}
        self._plans_lock = threading.Lock()
        self._lib = _lib.load()

    def get_config(self):
        config = super().get_config()
        config.update({'window': self.window, 'periodic': self.periodic, 'to_magnitude': True})
        return config

    # -- device plumbing ---------------------------------------------------
    def _plan(self, device, stream=None):
        """The native plan for (device, stream).  A plan owns the counters its kernels synchronise
        through, so calls on one plan must not overlap on the GPU: every stream gets its own
        (threads that share a stream are ordered by that stream)."""
        index = device.index if device.index is not None else torch.cuda.current_device()
        if stream is None:
            stream = torch.cuda.current_stream(index).cuda_stream
        key = (index, int(stream))
        with self._plans_lock:
            if key not in self._plans:
                import ctypes
                handle = ctypes.c_void_p()
                win = np.ascontiguousarray(self._window, dtype=np.float32)
                basis = np.ascontiguousarray(self.mel_basis[0], dtype=np.float32)
                with torch.cuda.device(index):
                    _lib.check(self._lib.stt_logmel_plan_create(
                        int(self.n_mel_channels),
                        win.ctypes.data_as(ctypes.POINTER(ctypes.c_float)),
                        basis.ctypes.data_as(ctypes.POINTER(ctypes.c_float)),
                        ctypes.byref(handle)))
                self._plans[key] = handle
            return self._plans[key]

    def __del__(self):
        try:
            for handle in self._plans.values():
                self._lib.stt_logmel_plan_destroy(handle)
        except Exception:
            pass

    def _as_device_audio(self, audio):
        if isinstance(audio, torch.Tensor):
            t = audio
        else:
            t = torch.from_numpy(np.ascontiguousarray(np.asarray(audio, dtype=np.float32)))
        device = self._device or (t.device if t.is_cuda else torch.device('cuda', torch.cuda.current_device()))
        t = t.to(device=device, dtype=torch.float32).contiguous()
        if t.dim() == 2:
            if t.shape[0] != 1:
                raise ValueError("one clip per call: expected shape [n] or [1, n], got {}".format(tuple(t.shape)))
            t = t[0]
        if t.dim() != 1 or t.numel() == 0:
            raise ValueError("audio must be a non-empty 1-D signal")
        if t.data_ptr() % 16:
            t = t.clone()
        return t

    def __call__(self, audio, **kwargs):
        """audio: [n] or [1, n] samples in [-1, 1].  Returns a CUDA tensor
        [1, frames, n_mel_channels] (stft.py:96-120)."""
        return self.mel_spectrogram(audio)

    def mel_spectrogram(self, audio):
        pcm = self._as_device_audio(audio)
        n = pcm.numel()
        frames = self._lib.stt_logmel_num_frames(n)
        ws_bytes = self._lib.stt_logmel_workspace_bytes(n)
        with torch.cuda.device(pcm.device):
            mel = torch.empty((1, frames, self.n_mel_channels), dtype=torch.float32, device=pcm.device)
            ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=pcm.device)
            stream = torch.cuda.current_stream(pcm.device).cuda_stream
            _lib.check(self._lib.stt_logmel_run(
                self._plan(pcm.device), pcm.data_ptr(), n, mel.data_ptr(), ws.data_ptr(), ws_bytes,
                stream))
        return mel


    # -- a clip computed in pieces (one per GPU): frames [frame_begin, +frame_count) ------------
    def piece_spectrogram(self, pcm_piece, sample_origin, n_total, frame_begin, frame_count):
        """First half of the log-mel for one piece of a clip: returns (z, piece_max, state) where
        z = (log10(mel) + 4) / 4 for the piece's frames, piece_max is a 1-element CUDA tensor with
        the piece's maximum of log10(mel), and state must be handed to `finish_piece` together
        with the clip-wide maximum (the max over every piece's piece_max)."""
        pcm = self._as_device_audio(pcm_piece)
        with torch.cuda.device(pcm.device):
            ws_bytes = self._lib.stt_logmel_workspace_bytes(max(int(frame_count), 3) * 160)
            mel = torch.empty((int(frame_count), self.n_mel_channels), dtype=torch.float32, device=pcm.device)
            ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=pcm.device)
            piece_max = torch.empty((1,), dtype=torch.float32, device=pcm.device)
            stream = torch.cuda.current_stream(pcm.device).cuda_stream
            _lib.check(self._lib.stt_logmel_stage1(
                self._plan(pcm.device), pcm.data_ptr(), int(sample_origin), pcm.numel(), int(n_total),
                int(frame_begin), int(frame_count), mel.data_ptr(), piece_max.data_ptr(), ws.data_ptr(),
                ws_bytes, stream))
        return mel, piece_max, ws

    def finish_piece(self, mel, clip_max, state):
        with torch.cuda.device(mel.device):
            clip_max = clip_max.to(device=mel.device, dtype=torch.float32).contiguous()
            stream = torch.cuda.current_stream(mel.device).cuda_stream
            _lib.check(self._lib.stt_logmel_stage2(
                self._plan(mel.device), mel.data_ptr(), mel.shape[0], clip_max.data_ptr(),
                state.data_ptr(), stream))
        return mel

