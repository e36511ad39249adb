"""Kern tokenizer for the transcription decoder (host side, pure Python).

Drop-in for the one class of the reference's ``data_processing/humdrum.py`` that the
training hot path needs: ``LabelsMultiple`` (reference ``data_processing/humdrum.py:70-131``).
The rest of that file (score munging built on music21 / humextra) is offline data
preparation and is out of scope.

The 173-symbol table is *generated* from the structure of the ``**kern`` alphabet
(rhythm values, pitch spellings by octave, structural marks) instead of being listed;
``tests/test_tokenizer.py`` pins it against the table and the known-answer vectors that
``tests/golden/make_golden.py`` dumped from the reference implementation.
"""
import re

_LETTERS = "cdefgab"


def _octave_names(reps, upper, first=None, last=None):
    """Spellings of one octave: for every letter flat, natural, sharp (kern: ``-``/``#``)."""
    names = []
    for letter in _LETTERS:
        stem = (letter.upper() if upper else letter) * reps
        names.extend([stem + "-", stem, stem + "#"])
    lo = names.index(first) if first is not None else 0
    hi = names.index(last) + 1 if last is not None else len(names)
    return names[lo:hi]


def _base_labels():
    rhythm = []
    for base in (1, 2, 4, 8, 16, 32, 64):            # binary values, plain and dotted
        rhythm.extend([str(base), str(base) + "."])
    rhythm.extend(str(v) for v in (3, 6, 12, 24, 48, 96))   # triplet family
    pitches = ["BBB#"]
    pitches += _octave_names(2, True, first="CC")      # CC .. BB#   (CC- only in the extension)
    pitches += _octave_names(1, True)                  # C- .. B#
    pitches += _octave_names(1, False)                 # c- .. b#
    pitches += _octave_names(2, False)                 # cc- .. bb#
    pitches += _octave_names(3, False)                 # ccc- .. bbb#
    pitches += _octave_names(4, False, last="ffff")    # cccc- .. ffff
    marks = ["r", ".", "[", "_", "]", ";", "\t", "\n", "<b>"]
    control = ["<sos>", "<eos>", "<pad>"]
    return rhythm + pitches + marks + control


def _extension_labels():
    rhythm = ["128", "20", "40", "176", "112"]
    low = _octave_names(3, True, first="CCC", last="BBB")   # CCC .. BBB (no CCC-, no BBB#)
    return rhythm + low + ["CC-"]


_NOTE_RE = re.compile(r"(\[?)(\d+\.*)([a-gA-Gr]{1,4}[\-#]*)(;?)([\]_]?)")


class LabelsMultiple(object):
    """148 (base) / 173 (extended) symbol vocabulary; same surface as the reference class.

    ``labels`` (list), ``labels_map`` (symbol -> id), ``labels_map_inv`` (id -> symbol),
    ``encode(text) -> [ids]``, ``decode(ids) -> [symbols]``.
    """

    def __init__(self, extended=False):
        self.labels = _base_labels()
        if extended:
            self.labels = self.labels + _extension_labels()
        self.labels_map = {sym: i for i, sym in enumerate(self.labels)}
        self.labels_map_inv = {i: sym for i, sym in enumerate(self.labels)}

    def encode(self, chars):
        """``**kern`` text -> token ids.

        Lines are separated by ``\\n`` tokens, spines by ``\\t`` tokens, notes of a chord by
        ``<b>``; a multi-character note is split into [tie-open][duration][pitch][fermata]
        [tie-close/continue].  A note that does not parse raises (as the reference does).
        """
        ids = self.labels_map
        out = []
        for line in chars.splitlines():
            for chord in line.split("\t"):
                for note in chord.split(" "):
                    if len(note) == 1:
                        out.append(ids[note])
                    else:
                        m = _NOTE_RE.fullmatch(note)
                        if m is None:
                            raise Exception(f"Item {note} in {line} does not match")
                        out.extend(ids[g] for g in m.groups() if g)
                    out.append(ids["<b>"])
                if out[-1] == ids["<b>"]:
                    out.pop()
                out.append(ids["\t"])
            out[-1] = ids["\n"]
        out.pop()
        return out

    def decode(self, tokens):
        syms = [self.labels_map_inv.get(t) for t in tokens]
        return [" " if s == "<b>" else s for s in syms if s]
