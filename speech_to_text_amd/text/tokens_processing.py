"""Host-side token helpers with the reference's names (utils/text/tokens_processing.py:20-81).

Inside the generation loop the same masks are applied by the fused selection kernel (a list of
token ids = `mask_batch_tokens`, the timestamp rules' ranges = `mask_slice_tokens`); these
functions are for callers that post-process logits themselves, e.g. the output of
`MI355XRuntime.decoder`.  They take and return numpy arrays or torch tensors (any device) and, like
the reference's, never modify their argument.
"""
import numpy as np

from .metrics import process_model_output   # noqa: F401  (tokens_processing.py:20-39)


def _copy(logits):
    return logits.clone() if hasattr(logits, 'clone') else np.array(logits, copy=True)


def mask_tokens(logits, indices, value=float('-inf')):
    """`logits[indices[:, 0], indices[:, 1]] = value` on a copy."""
    out = _copy(logits)
    idx = np.asarray(indices.cpu() if hasattr(indices, 'cpu') else indices, dtype=np.int64).reshape(-1, 2)
    out[idx[:, 0], idx[:, 1]] = value
    return out


def mask_batch_tokens(logits, indices, value=float('-inf')):
    """`logits[:, indices] = value` on a copy."""
    out = _copy(logits)
    idx = np.asarray(indices.cpu() if hasattr(indices, 'cpu') else indices, dtype=np.int64).reshape(-1)
    out[:, idx] = value
    return out


def mask_slice_tokens(logits, index, remove_after, value=float('-inf')):
    """`logits[:, index:] = value` (`remove_after`) or `logits[:, :index] = value`, on a copy."""
    out = _copy(logits)
    index = int(index)
    if remove_after:
        out[:, index:] = value
    else:
        out[:, :index] = value
    return out
