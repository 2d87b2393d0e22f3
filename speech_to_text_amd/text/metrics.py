"""Weighted Levenshtein distance with partial alignment: the building block of the README's
keyword ``search`` (utils/text/metrics.py:85-159).  Host code: the strings are a few hundred
symbols long."""
import numpy as np


def _normalize(data):
    if isinstance(data, np.ndarray): data = data.tolist()
    if isinstance(data, bytes): data = data.decode()
    if isinstance(data, (list, tuple)) and len(data) and isinstance(data[0], (int, np.integer)):
        data = ' '.join(str(d) for d in data)
    return data


def edit_distance(hypothesis, truth, partial=False, deletion_cost={}, insertion_cost={},
                  replacement_cost={}, default_del_cost=1, default_insert_cost=1,
                  default_replace_cost=1, normalize=True, return_matrix=False, **kwargs):
    hypothesis, truth = _normalize(hypothesis), _normalize(truth)
    n, m = len(hypothesis), len(truth)
    matrix = np.zeros((n + 1, m + 1))
    del_costs = np.array([0] + [deletion_cost.get(h, default_del_cost) for h in hypothesis])
    ins_costs = np.array([insertion_cost.get(t, default_insert_cost) for t in truth])
    matrix[:, 0] = np.cumsum(del_costs)
    if not partial:
        matrix[0, 1:] = np.cumsum(ins_costs)
    truth_arr = np.array(list(truth)) if isinstance(truth, str) else np.asarray(truth)
    for i in range(1, n + 1):
        h = hypothesis[i - 1]
        sub = np.array([replacement_cost.get(h, {}).get(t, default_replace_cost) for t in truth])
        diag = matrix[i - 1, :-1] + sub * (truth_arr != h)
        up = matrix[i - 1, 1:] + del_costs[i]
        best = np.minimum(up, diag)
        row = matrix[i]
        for j in range(1, m + 1):
            left = row[j - 1] + ins_costs[j - 1]
            row[j] = best[j - 1] if best[j - 1] < left else left
    distance = matrix[-1, -1] if not partial else np.min(matrix[-1, 1:])
    if normalize:
        distance = distance / m if not partial else distance / n
    return distance if not return_matrix else (distance, matrix)


def process_model_output(output, offset=None, lengths=None):
    """tokens[b][(beam)][offset : length] as arrays (utils/text/tokens_processing.py:20-39)."""
    if hasattr(output, 'lengths') or hasattr(output, 'offset'):
        lengths = np.asarray(output.lengths)
        if not hasattr(output, 'offset'):
            offset = np.zeros_like(lengths)
        else:
            offset = np.asarray(output.offset)
            if lengths.ndim == 2 and offset.ndim == 1:
                offset = np.tile(offset[:, None], [1, lengths.shape[1]])
        output = output.tokens
    elif lengths is None:
        return output.tokens
    if lengths.ndim:
        return [process_model_output(out, off, length) for out, off, length in zip(output, offset, lengths)]
    return np.asarray(output[int(offset): int(lengths)])
