"""Weighted Levenshtein distance with partial alignment - the building block of the README's
keyword ``search`` - and the unpacking of a generation result into token lists.

Same signatures and results as the reference's `edit_distance` (utils/text/metrics.py:85-159)
and `process_model_output` (utils/text/tokens_processing.py:20-39).  Host code: the strings are
a few hundred symbols long.
"""
import numpy as np


def _as_symbols(data):
    """bytes -> str, int sequences -> the string of their decimal forms joined by blanks (what the
    reference's dispatcher does before the distance, metrics.py:243-249)."""
    if isinstance(data, np.ndarray):
        data = data.tolist()
    if isinstance(data, bytes):
        data = data.decode()
    if isinstance(data, (list, tuple)) and data and isinstance(data[0], (int, np.integer)):
        data = " ".join(map(str, data))
    return data


def edit_distance(hypothesis, truth, partial=False, deletion_cost={}, insertion_cost={},
                  replacement_cost={}, default_del_cost=1, default_insert_cost=1,
                  default_replace_cost=1, normalize=True, return_matrix=False, **kwargs):
    """D[i, j] = cost of turning hypothesis[:i] into truth[:j]; with `partial` the first row is
    free (the hypothesis may start anywhere in `truth`) and the result is the minimum of the last
    row (it may end anywhere).  Normalised by len(truth), or len(hypothesis) when `partial`.

    A row is filled without a Python loop over columns: with c[j] the cost of inserting truth[:j]
    and v[j] the best of "delete" / "match or replace" for column j (v[0] = D[i, 0]), the insert
    recurrence D[i, j] = min(v[j], D[i, j-1] + ins[j-1]) unrolls to
    D[i, j] = c[j] + min_{k <= j} (v[k] - c[k]), a running minimum.  (Exact for costs that are
    sums of powers of two, such as the defaults; other costs agree to rounding.)"""
    hyp, ref = _as_symbols(hypothesis), _as_symbols(truth)
    rows, cols = len(hyp), len(ref)
    ref_symbols = np.array(list(ref)) if isinstance(ref, str) else np.asarray(ref)
    ins_prefix = np.concatenate([[0.], np.cumsum([insertion_cost.get(t, default_insert_cost) for t in ref],
                                                  dtype=np.float64)])
    table = np.zeros((rows + 1, cols + 1))
    if not partial:
        table[0] = ins_prefix
    for i, symbol in enumerate(hyp, start=1):
        above = table[i - 1]
        swap = replacement_cost.get(symbol, {})
        change = np.fromiter((swap.get(t, default_replace_cost) for t in ref), dtype=np.float64, count=cols)
        change[ref_symbols == symbol] = 0.
        drop = deletion_cost.get(symbol, default_del_cost)
        candidates = np.empty(cols + 1)
        candidates[0] = above[0] + drop
        np.minimum(above[1:] + drop, above[:-1] + change, out=candidates[1:])
        table[i] = np.minimum.accumulate(candidates - ins_prefix) + ins_prefix
    distance = table[-1, 1:].min() if partial else table[-1, -1]
    if normalize:
        distance = distance / (rows if partial else cols)
    return (distance, table) if return_matrix else distance


def process_model_output(output, offset=None, lengths=None):
    """Token ids without prompt and padding: `tokens[...][offset : length]` for every leading index
    of `lengths`, as nested lists of arrays (one level per dimension of `lengths`: batch, then
    beams).  `output` is an object with `tokens` / `lengths` (/ `offset`) or the token array
    itself; without lengths the tokens come back untouched."""
    tokens = getattr(output, "tokens", output)
    if lengths is None:
        lengths = getattr(output, "lengths", None)
    if lengths is None:
        return tokens
    if offset is None:
        offset = getattr(output, "offset", None)
    lengths = np.asarray(lengths)
    if offset is None:
        first = np.zeros(lengths.shape, dtype=np.int64)
    else:
        first = np.asarray(offset)
        first = np.broadcast_to(first.reshape(first.shape + (1,) * (lengths.ndim - first.ndim)), lengths.shape)

    def unpack(prefix, depth):
        if depth == lengths.ndim:
            return np.asarray(tokens[prefix][int(first[prefix]): int(lengths[prefix])])
        return [unpack(prefix + (k,), depth + 1) for k in range(lengths.shape[depth])]

    return unpack((), 0)
