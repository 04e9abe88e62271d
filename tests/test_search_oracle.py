"""Pin the CPU restatement of eval.py's greedy / beam search against the reference's own outputs."""
import numpy as np
import pytest
import torch

from oracle import search_oracle as S
from tests import golden_util as GU

CASES = ["search_small", "search_small_b", "search_eos", "search_stop", "search_gru", "search_gru_b", "search_gru_stop"]


def load_search_case(name):
    g = GU.load(name)
    B, F, D, V, E, H, A = [int(x) for x in g["meta_dims"]]
    g["_cell"] = GU.cells_of(g)[0]
    P = GU.formula_params(GU.decoder_shapes(V, E, H, A, D, g["_cell"]), int(g["meta_seed"]))
    sc = float(g["meta_scale"])
    P["out.weight"] = P["out.weight"] * sc
    P["out.bias"] = P["out.bias"] * sc
    P["out.bias"][2] += float(g["meta_eos_bias"])
    P["out.bias"][0] += float(g["meta_pad_bias"])
    return g, P, torch.from_numpy(g["enc"])


@pytest.mark.parametrize("name", CASES)
def test_greedy_matches_reference(name):
    g, P, enc = load_search_case(name)
    with torch.no_grad():
        out = S.greedy_search(P, enc, cell=g["_cell"])
    assert np.array_equal(out, g["greedy"])


@pytest.mark.parametrize("bw", [1, 3, 5])
@pytest.mark.parametrize("name", CASES)
def test_beam_matches_reference(name, bw):
    g, P, enc = load_search_case(name)
    with torch.no_grad():
        out = S.beam_search(P, enc, bw, cell=g["_cell"])
    assert np.array_equal(out, g["beam%d" % bw])
