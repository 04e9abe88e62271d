"""recnet_amd.metrics (Python-3 BLEU / ROUGE-L / CIDEr) against the scores of the reference's own coco-caption scorers
on a seeded corpus (tests/golden/metrics.json, made by tests/golden/make_golden_metrics.py)."""
import json
import os

import numpy as np
import pytest

from recnet_amd import metrics as M

G = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "metrics.json")))
GTS = {k: G["gts"][k] for k in G["ids"]}
RES = {k: G["res"][k] for k in G["ids"]}


def test_bleu_matches_reference_scorer():
    b, per = M.bleu(GTS, RES)
    np.testing.assert_allclose(b, G["bleu"], rtol=1e-12)
    np.testing.assert_allclose(per, G["bleu_per_id"], rtol=1e-12, atol=1e-300)


def test_rouge_matches_reference_scorer():
    r, per = M.rouge_l(GTS, RES)
    assert abs(r - G["rouge"]) <= 1e-12
    np.testing.assert_allclose(per, G["rouge_per_id"], rtol=1e-12)


def test_cider_matches_reference_scorer():
    c, per = M.cider(GTS, RES)
    assert abs(c - G["cider"]) <= 1e-10
    np.testing.assert_allclose(per, G["cider_per_id"], rtol=1e-9, atol=1e-12)


def test_known_answers_and_errors():
    i = G["ids"].index("exact")
    assert abs(M.rouge_l(GTS, RES)[1][i] - 1.0) < 1e-12 and M.bleu(GTS, RES)[1][3][i] > 0.99
    s = M.score_all(GTS, RES)
    assert set(s) == {"Bleu_1", "Bleu_2", "Bleu_3", "Bleu_4", "CIDEr", "ROUGE_L"}
    with pytest.raises(ValueError):
        M.bleu({"a": ["x"]}, {"b": ["x"]})
    with pytest.raises(ValueError):
        M.cider({"a": ["x"]}, {"a": ["x", "y"]})
    assert M.indices_to_sentence([5, 6, 2, 7], {5: "a", 6: "dog", 7: "runs"}) == "a dog"
