"""Kern tokenizer vs known-answer vectors dumped from the reference's LabelsMultiple
(tests/golden/tokenizer_kat.json, written by tests/golden/make_golden.py; reference
data_processing/humdrum.py:70-131)."""
import json
import os

import pytest

from data_processing.humdrum import LabelsMultiple


@pytest.fixture(scope="module")
def kat(golden_dir):
    return json.load(open(os.path.join(golden_dir, "tokenizer_kat.json")))


def test_label_table(kat):
    lab = LabelsMultiple(extended=True)
    assert lab.labels == kat["labels_extended"]
    assert len(LabelsMultiple(extended=False).labels) == kat["n_base"] == 148
    assert len(lab.labels) == 173
    assert (lab.labels_map["<sos>"], lab.labels_map["<eos>"], lab.labels_map["<pad>"]) == (145, 146, 147)
    assert (lab.labels_map["\t"], lab.labels_map["\n"], lab.labels_map["<b>"]) == (142, 143, 144)
    for i, s in ((0, "1"), (19, "96"), (20, "BBB#"), (148, "128"), (172, "CC-")):
        assert lab.labels_map_inv[i] == s


def test_known_answers(kat):
    lab = LabelsMultiple(extended=True)
    for case in kat["kats"]:
        if "raises" in case:
            with pytest.raises(Exception):
                lab.encode(case["text"])
        else:
            ids = lab.encode(case["text"])
            assert ids == case["ids"], case["text"]
            if "decoded" in case:
                assert lab.decode(ids) == case["decoded"]


def test_survey_vectors():
    lab = LabelsMultiple(extended=True)
    assert lab.encode("4c") == [4, 63]
    assert lab.encode("4c\t8e 8g\n4r") == [4, 63, 142, 6, 69, 144, 6, 75, 143, 4, 136]
    assert lab.encode("[2.CC#_ 4ee-;]\t.\n16ffff") == [138, 3, 22, 139, 144, 4, 89, 141, 140, 142, 137, 143, 8, 135]
    assert "".join(lab.decode(lab.encode("8.r\t4c 4e 4g"))) == "8.r\t4c 4e 4g"
    with pytest.raises(Exception):
        lab.encode("4h")
