"""Validation metrics of the recipe (reference pretrain.py:216-249): word error rate over ``" \\n = \\n "``-joined bars of
space-joined Kern symbols (the reference calls jiwer.wer -- third-party, absent here: restated as word-level Levenshtein distance
/ reference length, jiwer's definition), and macro-F1 of key / time-signature ids (sklearn.metrics.f1_score, as the reference)."""
import numpy as np

from .spec import EOS

BAR_JOIN = " \n = \n "


def word_error_rate(reference, hypothesis):
    """(substitutions + deletions + insertions) / number of reference words, on whitespace-split words."""
    r, h = reference.split(), hypothesis.split()
    if not r:
        return float(len(h) > 0)
    prev = list(range(len(h) + 1))
    for i, rw in enumerate(r, 1):
        cur = [i] + [0] * len(h)
        for j, hw in enumerate(h, 1):
            cur[j] = min(prev[j] + 1, cur[j - 1] + 1, prev[j - 1] + (rw != hw))
        prev = cur
    return prev[-1] / len(r)


def unpad(ids):
    """Token row -> ids before the first <eos> (reference pretrain.py:245-249)."""
    ids = np.asarray(ids)
    hit = np.nonzero(ids == EOS)[0]
    return ids[: hit[0]] if len(hit) else ids


def ids_to_text(rows, inv_map):
    return BAR_JOIN.join(" ".join(inv_map[int(i)] for i in row) for row in rows)


def corpus_wer(pred, target, inv_map):
    """pred/target: dict id -> list of per-bar id lists.  Returns (mean WER over clips, per-clip dict)."""
    per = {k: word_error_rate(ids_to_text(target[k], inv_map), ids_to_text(pred[k], inv_map)) for k in pred}
    return (sum(per.values()) / max(len(per), 1)), per


def corpus_f1(pred, target):
    from sklearn.metrics import f1_score
    per = {k: float(f1_score(target[k], pred[k], average="macro")) for k in pred}
    return (sum(per.values()) / max(len(per), 1)), per
