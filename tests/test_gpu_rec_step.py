"""The reconstructors' per-step API on the HIP path: the reference's own time loops (train.py:82-94 and 112-123),
verbatim, against this package's GlobalReconstructor / LocalReconstructor modules; results held to goldens the
reference's modules produced in the same loops (tests/golden/make_golden_recstep.py)."""
import numpy as np
import pytest
import torch

import recnet_amd as R
from tests.test_recstep_oracle import CASES, case_inputs

pytestmark = pytest.mark.gpu


class _C:            # the globals the reference's loops read (config.TrainConfig attributes)
    device = "cuda"


@pytest.mark.parametrize("prec", ["f32", "bf16"])
@pytest.mark.parametrize("name", CASES)
def test_reference_time_loops_over_package_modules(name, prec):
    g, (B, T, F, H, Rr, RA), kind, cell, P, hid = case_inputs(name)
    C = _C
    C.batch_size, C.reconstructor_hidden_size, C.encoder_output_len, C.reconstructor_model = B, Rr, F, cell
    C.reconstructor_n_layers = 1
    if kind == "global":
        reconstructor = R.GlobalReconstructor(model_name=cell, n_layers=1, decoder_hidden_size=H, hidden_size=Rr, dropout=0.5,
                                              decoder_dropout=0.5, caption_max_len=30, precision=prec)
    else:
        reconstructor = R.LocalReconstructor(model_name=cell, n_layers=1, decoder_hidden_size=H, hidden_size=Rr, dropout=0.5,
                                             decoder_dropout=0.5, attn_size=RA, precision=prec)
    reconstructor.load_state_dict(P)
    reconstructor = reconstructor.to(C.device)
    reconstructor.dropout_seed = int(g["meta_drop_seed"])
    reconstructor.train(bool(int(g["meta_train_mode"])))
    decoder_hiddens = hid.to(C.device)
    outs, hs, cs = [], [], []
    # ---- train.py:82-89 / 112-119, as written
    if C.reconstructor_model == "LSTM":
        reconstructor_hidden = (
            torch.zeros(C.reconstructor_n_layers, C.batch_size, C.reconstructor_hidden_size).to(C.device),
            torch.zeros(C.reconstructor_n_layers, C.batch_size, C.reconstructor_hidden_size).to(C.device))
    else:
        reconstructor_hidden = torch.zeros(C.reconstructor_n_layers, C.batch_size, C.reconstructor_hidden_size)
        reconstructor_hidden = reconstructor_hidden.to(C.device)
    if kind == "global":
        decoder_len = decoder_hiddens.size(0)                                                # train.py:92
        for t in range(decoder_len):                                                         # train.py:93-94
            decoder_hidden = decoder_hiddens[t].to(C.device)
            reconstructor_output, reconstructor_hidden = reconstructor(decoder_hidden, reconstructor_hidden, decoder_hiddens)
            outs.append(reconstructor_output); hs.append(reconstructor_hidden)
    else:
        for t in range(C.encoder_output_len):                                                # train.py:122-123
            reconstructor_output, reconstructor_hidden = reconstructor(reconstructor_hidden, decoder_hiddens)
            outs.append(reconstructor_output); hs.append(reconstructor_hidden)
    tol = 2e-5 if prec == "f32" else 1e-2
    for t, (o, hd) in enumerate(zip(outs, hs)):
        h = (hd[0] if cell == "LSTM" else hd)
        assert tuple(o.shape) == (B, Rr) and tuple(h.shape) == (1, B, Rr)
        assert np.abs(o.cpu().numpy() - g["out"][t]).max() <= tol, (name, prec, t)
        assert np.abs(h[0].cpu().numpy() - g["h"][t]).max() <= tol, (name, prec, t)
        if cell == "LSTM":
            assert np.abs(hd[1][0].cpu().numpy() - g["c"][t]).max() <= tol, (name, prec, t)


def test_eval_mode_switches_dropout_off_and_state_dict_round_trips():
    g, (B, T, F, H, Rr, RA), kind, cell, P, hid = case_inputs("recstep_local_train")
    m = R.LocalReconstructor(model_name=cell, n_layers=1, decoder_hidden_size=H, hidden_size=Rr, dropout=0.5,
                             decoder_dropout=0.5, attn_size=RA, precision="f32").to("cuda")
    m.load_state_dict(P)
    assert sorted(m.state_dict().keys()) == sorted(P.keys())
    hidden = (torch.zeros(1, B, Rr, device="cuda"), torch.zeros(1, B, Rr, device="cuda"))
    m.eval()
    o_eval, _ = m(hidden, hid.cuda())
    ge = np.load("tests/golden/recstep_local_eval.npz")
    assert np.abs(o_eval.cpu().numpy() - ge["out"][0]).max() <= 2e-5
    m.train()
    m.dropout_seed = int(g["meta_drop_seed"]); m._calls = 0
    o_train, _ = m(hidden, hid.cuda())
    assert np.abs(o_train.cpu().numpy() - g["out"][0]).max() <= 2e-5
    assert np.abs(o_train.cpu().numpy() - o_eval.cpu().numpy()).max() > 1e-4
