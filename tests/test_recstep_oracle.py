"""The oracle's per-step reconstructor functions against goldens produced by the reference's own modules in the
reference's own time loops (tests/golden/make_golden_recstep.py; train.py:82-94, 112-123)."""
import numpy as np
import pytest
import torch

from oracle import recnet_oracle as O
from tests import golden_util as GU

CASES = ["recstep_global_eval", "recstep_global_train", "recstep_local_eval", "recstep_local_train",
         "recstep_global_gru", "recstep_local_gru"]


def make_hiddens(T, B, H, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.tanh(torch.randn(T, 1, B, H, generator=g)) * 0.6


def case_inputs(name):
    g = GU.load(name)
    B, T, F, H, R, RA = [int(x) for x in g["meta_dims"]]
    kind = "global" if "global" in name else "local"
    cell = "GRU" if int(g["meta_gru"]) else "LSTM"
    P = GU.formula_params(GU.rec_shapes(kind, H, R, RA, cell), int(g["meta_seed"]))
    hid = make_hiddens(T, B, H, int(g["meta_seed"]) + 50)
    return g, (B, T, F, H, R, RA), kind, cell, P, hid


@pytest.mark.parametrize("name", CASES)
def test_oracle_step_functions_match_reference_loops(name):
    g, (B, T, F, H, R, RA), kind, cell, P, hid = case_inputs(name)
    drop = O.Dropper("hash", seed=int(g["meta_drop_seed"])) if int(g["meta_train_mode"]) else O.Dropper("eval")
    hidden = O.zero_hidden(B, R, cell)
    n = T if kind == "global" else F
    for t in range(n):
        if kind == "global":
            out, hidden = O.global_rec_step(P, hid[t], hidden, hid, cell=cell, drop=drop, t=t)
        else:
            out, hidden = O.local_rec_step(P, hidden, hid, cell=cell, drop=drop, t=t)
        h = (hidden[0] if cell == "LSTM" else hidden)[0]
        assert np.abs(out.numpy() - g["out"][t]).max() <= 2e-6, (name, t)
        assert np.abs(h.numpy() - g["h"][t]).max() <= 2e-6, (name, t)
        if cell == "LSTM":
            assert np.abs(hidden[1][0].numpy() - g["c"][t]).max() <= 2e-6, (name, t)
