"""Pin the CPU oracle (oracle/recnet_oracle.py) against golden vectors produced by running the
reference itself (tests/golden/make_golden.py).  CPU-only."""
import numpy as np
import pytest
import torch

from oracle import recnet_oracle as O
from tests import golden_util as GU

SMALL_CASES = ["dec_eval", "dec_train", "dec_T31", "dec_T4_samelen", "global_train", "global_eval",
               "local_train", "local_eval", "local_T31", "gru_local_train", "gru_global_eval", "gru_dec_train",
               "gru_global_train", "gru_local_train3"]
FULL_CASES = ["full_dec_B8", "full_global_B8", "full_local_B8", "full_gru_global_B8", "full_gru_local_B8"]


def _setup(name):
    g = GU.load(name)
    B, F, D, V, E, H, A, RA = [int(x) for x in g["meta_dims"]]
    cells = GU.cells_of(g)
    kind = "global" if "global" in name else ("local" if "local" in name else None)
    fs = int(g["meta_formula_seed"])
    if fs >= 0:
        decP = GU.formula_params(GU.decoder_shapes(V, E, H, A, D, cells[0]), fs)
        recP = GU.formula_params(GU.rec_shapes(kind, H, D, RA, cells[1]), fs + 1) if kind else None
        enc, targets = GU.make_batch(B, F, D, V, g["meta_lens"], int(g["meta_batch_seed"]))
        assert np.array_equal(targets.numpy(), g["targets"])
    else:
        decP = GU.group(g, "dec_init")
        recP = GU.group(g, "rec_init") if kind else None
        enc = torch.from_numpy(g["enc"])
        targets = torch.from_numpy(g["targets"])
    return g, decP, recP, kind, cells, enc, targets


def _drop(g, it=0):
    if int(g["meta_train_mode"]):
        return O.Dropper("hash", seed=int(g["meta_drop_seed"]) + it)
    return O.Dropper("eval")


@pytest.mark.parametrize("name", SMALL_CASES + FULL_CASES)
def test_step_api_and_losses(name):
    g, decP, recP, kind, cells, enc, targets = _setup(name)
    masks = targets > 0
    B = enc.shape[0]
    H = decP["rnn.weight_hh_l0"].shape[1]
    drop = _drop(g)
    # step API (Decoder.forward) — logits and states per step
    tok = torch.full((1, B), 1, dtype=torch.long)
    hid = O.zero_hidden(B, H, cells[0])
    T = int(g["T"])
    assert O.decode_len(masks) == T
    for t in range(T):
        lg, hid = O.decoder_step(decP, tok, hid, enc, cell=cells[0], drop=drop, t=t)
        tok = targets[t].view(1, -1)
        h = (hid[0] if cells[0] == "LSTM" else hid)[0]
        np.testing.assert_allclose(h.numpy(), g["step_h"][t], atol=2e-6, rtol=0)
        if "step_logits" in g:
            np.testing.assert_allclose(lg.numpy(), g["step_logits"][t], atol=1e-5, rtol=1e-5)
        if cells[0] == "LSTM":
            np.testing.assert_allclose(hid[1][0].numpy(), g["step_c"][t], atol=2e-6, rtol=0)
    # sequence API + losses
    st = O.TrainState(decP, recP, kind, cell=cells[0], rec_cell=cells[1])
    dl, rl, hiddens, ce, mse = st.losses(enc, targets, masks, _drop(g))
    np.testing.assert_allclose(hiddens.detach().numpy(), g["hiddens"], atol=2e-6, rtol=0)
    assert abs(float(dl) - float(g["dec_loss"])) <= 2e-6 * max(1.0, abs(float(g["dec_loss"])))
    assert abs(float(ce) - float(g["dec_ce"])) <= 5e-6
    if kind:
        assert abs(float(rl) - float(g["rec_loss"])) <= 2e-6 * max(1.0, abs(float(g["rec_loss"])))
        assert abs(float(mse) - float(g["rec_mse"])) <= 5e-6


@pytest.mark.parametrize("name", SMALL_CASES + FULL_CASES)
def test_gradients_and_optimizer(name):
    g, decP, recP, kind, cells, enc, targets = _setup(name)
    masks = targets > 0
    st = O.TrainState(decP, recP, kind, cell=cells[0], rec_cell=cells[1])
    n_steps = int(g["meta_n_steps"])
    for it in range(n_steps):
        dl, rl, loss, gn = st.step(enc, targets, masks, _drop(g, it))
        assert abs(loss - float(g["loss_step%d" % it])) <= 3e-6 * max(1.0, abs(loss))
        if it == 0:
            assert abs(gn - float(g["dec_grad_norm"])) <= 2e-5 * max(1.0, gn)
            # gradients of step 0 are still in .grad (clipped in place only if gn > 50)
            scale = min(1.0, 50.0 / (gn + 1e-6))
            for grp, P in (("dec", st.dec), ("rec", st.rec)):
                if P is None:
                    continue
                for k, p in P.items():
                    gg = p.grad.numpy() / (scale if grp == "dec" else 1.0)
                    ref_n = float(g["%s_gnorm/%s" % (grp, k)])
                    if ("%s_grad/%s" % (grp, k)) in g:
                        ref = g["%s_grad/%s" % (grp, k)]
                        err = np.linalg.norm((gg - ref).astype(np.float64))
                        assert err <= 2e-5 * max(ref_n, 1e-6), (grp, k, err, ref_n)
                    else:
                        assert abs(np.linalg.norm(gg.astype(np.float64)) - ref_n) <= 2e-5 * max(ref_n, 1e-6)
                        np.testing.assert_allclose(gg.reshape(-1)[:64], g["%s_gslice/%s" % (grp, k)],
                                                   atol=2e-5 * max(ref_n, 1e-6) + 1e-9, rtol=1e-3)
    if ("dec_after%d/attn_b" % n_steps) in g:
        for k, v in GU.group(g, "dec_after%d" % n_steps).items():
            np.testing.assert_allclose(st.dec[k].detach().numpy(), v.numpy(), atol=1e-6, rtol=0)
        if kind:
            for k, v in GU.group(g, "rec_after%d" % n_steps).items():
                np.testing.assert_allclose(st.rec[k].detach().numpy(), v.numpy(), atol=1e-6, rtol=0)
        names = O.decoder_param_order(st.dec)
        for k in names:
            s = st.dec_opt.state[st.dec[k]]
            np.testing.assert_allclose(s["exp_avg"].numpy(), g["dec_opt/exp_avg/" + k], atol=1e-7, rtol=1e-4)
            np.testing.assert_allclose(s["exp_avg_sq"].numpy(), g["dec_opt/exp_avg_sq/" + k], atol=1e-10, rtol=1e-4)
            np.testing.assert_allclose(s["max_exp_avg_sq"].numpy(), g["dec_opt/max_exp_avg_sq/" + k],
                                       atol=1e-10, rtol=1e-4)
    else:
        for k in st.dec:
            np.testing.assert_allclose(st.dec[k].detach().numpy().reshape(-1)[:64],
                                       g["dec_pslice_after%d/%s" % (n_steps, k)], atol=1e-6, rtol=0)


FREE_CASES = ["free_dec", "free_global", "free_local_gru"]


@pytest.mark.parametrize("name", FREE_CASES)
def test_free_running_validation_pass(name):
    """train.py:310-340: eval mode, forward_decoder with teacher_forcing_ratio 0 (arg-max fed back), then the
    reconstructor on those hidden states — tokens exact, losses / hidden states to fp32 rounding."""
    g, decP, recP, kind, cells, enc, targets = _setup(name)
    with torch.no_grad():
        dl, hid, idx = O.forward_decoder(decP, enc, targets, targets > 0, cell=cells[0], drop=O.Dropper("eval"),
                                         teacher_forcing=False)
        assert np.array_equal(idx.numpy(), g["output_indices"])
        np.testing.assert_allclose(hid.numpy(), g["hiddens"], atol=2e-6, rtol=0)
        assert abs(float(dl) - float(g["dec_loss"])) <= 2e-6 * max(1.0, abs(float(g["dec_loss"])))
        if kind:
            fwd = O.forward_global_reconstructor if kind == "global" else O.forward_local_reconstructor
            rl = fwd(recP, hid, enc, cell=cells[1], drop=O.Dropper("eval"))
            assert abs(float(rl) - float(g["rec_loss"])) <= 2e-6 * max(1.0, abs(float(g["rec_loss"])))


TF_CASES = ["tf_half_global", "tf_half_local"]


@pytest.mark.parametrize("name", TF_CASES)
def test_teacher_forcing_ratio_below_one_in_training(name):
    """config.py:71 / train.py:38,251: with decoder_teacher_forcing_ratio < 1 every training iteration draws
    `random.random() <= ratio` from Python's global generator; the iterations that draw False feed the arg-max back
    (train.py:46-51) and are differentiated and stepped like the others.  Four reference iterations from random.seed(py_seed):
    the draws, the fed-back tokens, every loss and the parameters after the fourth step."""
    import random
    g, decP, recP, kind, cells, enc, targets = _setup(name)
    masks = targets > 0
    st = O.TrainState(decP, recP, kind, cell=cells[0], rec_cell=cells[1])
    n_steps, ratio = int(g["meta_n_steps"]), float(g["meta_tf_ratio"])
    random.seed(int(g["meta_py_seed"]))
    for it in range(n_steps):
        tf = random.random() <= ratio                                # train.py:38
        assert int(tf) == int(g["meta_tf_pattern"][it])
        dl, rl, loss, gn = st.step(enc, targets, masks, _drop(g, it), teacher_forcing=tf)
        assert abs(loss - float(g["loss_step%d" % it])) <= 3e-6 * max(1.0, abs(loss)), it
        if not tf:
            assert np.array_equal(st.last_output_indices.numpy(), g["output_indices_step%d" % it])
    assert 0 < int(g["meta_tf_pattern"].sum()) < n_steps             # both kinds of iteration are in the fixture
    for k, v in GU.group(g, "dec_after%d" % n_steps).items():
        np.testing.assert_allclose(st.dec[k].detach().numpy(), v.numpy(), atol=1e-6, rtol=0)
    for k, v in GU.group(g, "rec_after%d" % n_steps).items():
        np.testing.assert_allclose(st.rec[k].detach().numpy(), v.numpy(), atol=1e-6, rtol=0)


LR_CASES = ["lr_global_chain", "lr_local_chain", "lr_gru_global_chain"]


@pytest.mark.parametrize("name", LR_CASES)
def test_large_learning_rate_runs_pin_the_update_rule(name):
    """train.py:149,186,271-273 at learning rates of 1e-2 (config.py:86-87 set on the reference's own config by the generator): four
    iterations move every parameter by ~4e-2, so the losses of iterations 2-4 depend on the updates before them.  The oracle is held
    to every loss, the parameters and BOTH optimisers' moments after the fourth step; tests/test_gpu_parity.py holds the benchmarked
    update path (replayed graph, split reconstructor update, Adam in the GEMM epilogue) to the same vectors."""
    g, decP, recP, kind, cells, enc, targets = _setup(name)
    masks = targets > 0
    lr = g["meta_lr"]
    st = O.TrainState(decP, recP, kind, cell=cells[0], rec_cell=cells[1], dec_lr=float(lr[0]), rec_lr=float(lr[1]))
    n = int(g["meta_n_steps"])
    for it in range(n):
        dl, rl, loss, gn = st.step(enc, targets, masks, _drop(g, it))
        assert abs(loss - float(g["loss_step%d" % it])) <= 2e-5 * max(1.0, abs(loss)), it
    # the updates are large: the loss moved by far more than the comparison tolerance
    assert abs(float(g["loss_step%d" % (n - 1)]) - float(g["loss_step0"])) > 1e-2
    for grp, P, opt in (("dec", st.dec, st.dec_opt), ("rec", st.rec, st.rec_opt)):
        init = GU.group(g, grp + "_init")
        for k, v in GU.group(g, "%s_after%d" % (grp, n)).items():
            moved = float((v - init[k]).abs().max())
            assert moved > 5e-3, (grp, k, moved)
            np.testing.assert_allclose(P[k].detach().numpy(), v.numpy(), atol=2e-3 * moved, rtol=0)
            s = opt.state[P[k]]
            np.testing.assert_allclose(s["exp_avg"].numpy(), g["%s_opt/exp_avg/%s" % (grp, k)], atol=2e-6, rtol=2e-3)
            np.testing.assert_allclose(s["exp_avg_sq"].numpy(), g["%s_opt/exp_avg_sq/%s" % (grp, k)], atol=1e-9, rtol=4e-3)
