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


@pytest.mark.parametrize("kind", ["global", "local"])
def test_two_batches_at_the_same_address_are_not_confused(kind):
    """the loop invariants (pooled states / U_r . hiddens) were keyed on (data_ptr, _version); the caching
    allocator hands a freed batch's address to the next batch with _version 0.  Two different decoder_hiddens at ONE
    address, weights unchanged (eval mode): the second call has to see the second batch."""
    g, (B, T, F, H, Rr, RA), _, cell, P, hid = case_inputs("recstep_%s_eval" % kind)
    if kind == "global":
        m = R.GlobalReconstructor(model_name=cell, n_layers=1, decoder_hidden_size=H, hidden_size=Rr, dropout=0.5,
                                  decoder_dropout=0.5, caption_max_len=30, precision="f32")
    else:
        m = R.LocalReconstructor(model_name=cell, n_layers=1, decoder_hidden_size=H, hidden_size=Rr, dropout=0.5,
                                 decoder_dropout=0.5, attn_size=RA, precision="f32")
    m.load_state_dict(P)
    m = m.to("cuda").eval()
    zeros = lambda: (torch.zeros(1, B, Rr, device="cuda"), torch.zeros(1, B, Rr, device="cuda"))

    def first_step(hiddens_cpu):
        dh = hiddens_cpu.cuda()                      # a local: freed on return, its block goes back to the allocator
        ptr = dh.data_ptr()
        out = m(dh[0], zeros(), dh)[0] if kind == "global" else m(zeros(), dh)[0]
        return out.cpu().numpy(), ptr

    other = torch.flip(hid, dims=[0, 2]) * 0.5 + 0.1
    want_a, _ = first_step(hid)
    # reference value for `other` from a fresh module (no cached invariants)
    fresh = (R.GlobalReconstructor(model_name=cell, n_layers=1, decoder_hidden_size=H, hidden_size=Rr, dropout=0.5, decoder_dropout=0.5,
                                   caption_max_len=30, precision="f32") if kind == "global" else
             R.LocalReconstructor(model_name=cell, n_layers=1, decoder_hidden_size=H, hidden_size=Rr, dropout=0.5, decoder_dropout=0.5,
                                  attn_size=RA, precision="f32"))
    fresh.load_state_dict(P)
    fresh = fresh.to("cuda").eval()
    dho = other.cuda()
    want_b = (fresh(dho[0], zeros(), dho)[0] if kind == "global" else fresh(zeros(), dho)[0]).cpu().numpy()
    del dho
    torch.cuda.synchronize()
    got_a, pa = first_step(hid)
    got_b, pb = first_step(other)
    # Before the fix pa == pb here (the allocator reused the block) and got_b came out equal to got_a.  The module now
    # keeps the tensor it derived its invariants from alive, so the address cannot be reused while they are cached.
    assert np.abs(got_a - want_a).max() <= 1e-6
    assert np.abs(want_b - want_a).max() > 1e-4, "the two batches have to differ for the test to mean anything"
    assert np.abs(got_b - want_b).max() <= 1e-6


def test_wrong_shaped_hidden_is_a_runtime_error_not_a_device_write():
    """the torch.ops layer sizes its outputs from the handle and checks every tensor against recnet_dim."""
    g, (B, T, F, H, Rr, RA), kind, cell, P, hid = case_inputs("recstep_local_eval")
    m = R.LocalReconstructor(model_name=cell, n_layers=1, decoder_hidden_size=H, hidden_size=Rr, dropout=0.5,
                             decoder_dropout=0.5, attn_size=RA, precision="f32").to("cuda").eval()
    m.load_state_dict(P)
    dh = hid.cuda()
    good = (torch.zeros(1, B, Rr, device="cuda"), torch.zeros(1, B, Rr, device="cuda"))
    m(good, dh)
    for bad in ((torch.zeros(1, B, Rr - 8, device="cuda"),) * 2, (torch.zeros(1, B + 1, Rr, device="cuda"),) * 2):
        with pytest.raises(RuntimeError):
            m(bad, dh)
