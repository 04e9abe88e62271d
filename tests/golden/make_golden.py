#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by running THE REFERENCE ITSELF on CPU.

Run only in the build container (it needs /root/reference, which never travels):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

Recipe = SURVEY.md Appendix A: put /root/reference on sys.path, stub the three imports
of train.py that cannot be satisfied here (tensorboardX, dataset.MSVD, eval), set the
TrainConfig class attributes, then call the reference's own build_decoder /
build_reconstructor / forward_decoder / forward_*_reconstructor and its train-step
sequence (zero_grad, backward, clip_grad_norm_, Adam.step) unmodified.

Dropout: the reference's nn.Dropout sub-modules are replaced (attribute assignment on the
module instances, no reference file is touched) by HashDropout, which applies the
counter-based masks of oracle/dropmask.py, so train-mode results are reproducible by the
oracle and by the HIP kernels.

Only plain arrays are written (np.savez_compressed): inputs, parameters, outputs, gradients,
optimiser state.  No reference source, bytecode or pickled class goes into the repo.
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True
REF = "/root/reference"
sys.path.insert(0, REF)

from oracle import dropmask  # noqa: E402


def _stub(name, **attrs):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


_stub("tensorboardX", SummaryWriter=object)
_stub("dataset.MSVD", MSVD=object)
_stub("eval", evaluate=lambda *a, **k: None)

import train as ref_train  # noqa: E402  (the reference's train.py)
from config import TrainConfig as C  # noqa: E402


class DropState:
    def __init__(self):
        self.seed = 0
        self.counter = {}

    def reset(self, seed):
        self.seed = seed
        self.counter = {}


class HashDropout(torch.nn.Module):
    def __init__(self, p, site, state):
        super().__init__()
        self.p, self.site, self.state = p, site, state

    def forward(self, x):
        if not self.training or self.p <= 0:
            return x
        t = self.state.counter.get(self.site, 0)
        self.state.counter[self.site] = t + 1
        B, N = x.shape[-2], x.shape[-1]
        m = dropmask.keep_mask(self.state.seed, self.site, t, B, N, self.p)
        return x * torch.from_numpy(m).view(x.shape)


from tests.golden_util import make_batch, formula_params as _formula  # noqa: E402


def formula_params(sd, seed):
    return _formula({k: tuple(v.shape) for k, v in sd.items()}, seed)


def configure(*, B, F, D, V, E, H, A, dec_cell, rec_kind, rec_cell, RA):
    C.device = "cpu"
    C.decoder_model = dec_cell
    C.reconstructor_model = rec_cell
    C.batch_size = B
    C.encoder_output_len = F
    C.encoder_output_size = D
    C.embedding_size = E
    C.decoder_hidden_size = H
    C.decoder_attn_size = A
    C.use_recon = rec_kind is not None
    C.reconstructor_type = rec_kind or "local"
    C.reconstructor_hidden_size = D
    C.reconstructor_attn_size = RA
    C.caption_max_len = 30


ONLY = None   # --only a,b,c : regenerate just these cases


def run_case(name, *, B, F, D, V, E, H, A, lens, dec_cell="LSTM", rec_kind=None, rec_cell="LSTM",
             RA=None, train_mode=True, n_steps=3, seed=0, drop_seed=42, full=True, formula_seed=None,
             tf_ratio=1.0, py_seed=None, out_scale=None, dec_lr=None, rec_lr=None):
    """tf_ratio < 1 (config.py:71 decoder_teacher_forcing_ratio): forward_decoder draws `random.random() <= ratio` once per
    iteration (train.py:38) from Python's global generator, seeded here with py_seed; the iterations that draw False feed the
    arg-max back (train.py:46-51) and are differentiated like the others.  out_scale: a decisive vocabulary projection, so the
    arg-max of a free-running iteration is not a coin flip at default init."""
    if ONLY is not None and name not in ONLY:
        return
    RA = RA or A
    configure(B=B, F=F, D=D, V=V, E=E, H=H, A=A, dec_cell=dec_cell, rec_kind=rec_kind, rec_cell=rec_cell, RA=RA)
    torch.manual_seed(seed)
    # dec_lr / rec_lr: the reference's own config attributes (config.py:86-87) read by build_decoder / build_reconstructor
    # (train.py:149,186) — larger values make three updates move every parameter far beyond any comparison tolerance
    lr_keep = (C.decoder_learning_rate, C.reconstructor_learning_rate)
    if dec_lr is not None:
        C.decoder_learning_rate = dec_lr
    if rec_lr is not None:
        C.reconstructor_learning_rate = rec_lr
    dec = ref_train.build_decoder(V)
    rec = ref_train.build_reconstructor() if rec_kind else None
    C.decoder_learning_rate, C.reconstructor_learning_rate = lr_keep
    if formula_seed is not None:
        dec["model"].load_state_dict(formula_params(dec["model"].state_dict(), formula_seed))
        if rec:
            rec["model"].load_state_dict(formula_params(rec["model"].state_dict(), formula_seed + 1))
    st = DropState()
    dm = dec["model"]
    if out_scale is not None:
        with torch.no_grad():
            dm.out.weight.mul_(out_scale)
            dm.out.bias.mul_(out_scale)
    dm.embedding_dropout = HashDropout(C.embedding_dropout, dropmask.SITE_DEC_EMBED, st)
    dm.out_dropout = HashDropout(C.decoder_out_dropout, dropmask.SITE_DEC_LOGIT, st)
    if rec:
        rec["model"].decoder_dropout = HashDropout(C.reconstructor_decoder_dropout, dropmask.SITE_REC_INPUT, st)
    fwd_rec = {"global": ref_train.forward_global_reconstructor,
               "local": ref_train.forward_local_reconstructor}.get(rec_kind)

    enc, targets = make_batch(B, F, D, V, lens, seed + 100)
    masks = targets > 0
    out = {"meta_dims": np.array([B, F, D, V, E, H, A, RA], dtype=np.int64),
           "meta_lens": np.array(lens, dtype=np.int64), "meta_drop_seed": np.array(drop_seed),
           "meta_train_mode": np.array(int(train_mode)), "meta_n_steps": np.array(n_steps),
           "meta_formula_seed": np.array(-1 if formula_seed is None else formula_seed),
           "meta_cells": np.array([int(dec_cell == "GRU"), int(rec_cell == "GRU")], dtype=np.int64)}
    if dec_lr is not None or rec_lr is not None:
        out["meta_lr"] = np.array([lr_keep[0] if dec_lr is None else dec_lr, lr_keep[1] if rec_lr is None else rec_lr], dtype=np.float64)
    if full:
        out["enc"] = enc.numpy()
    out["targets"] = targets.numpy()
    out["meta_batch_seed"] = np.array(seed + 100)
    if full:
        for k, v in dm.state_dict().items():
            out["dec_init/" + k] = v.detach().numpy().copy()
        if rec:
            for k, v in rec["model"].state_dict().items():
                out["rec_init/" + k] = v.detach().numpy().copy()

    # ---- per-step logits / hidden states via the reference's step API (Decoder.forward)
    dm.train(train_mode)
    st.reset(drop_seed)
    with torch.no_grad():
        tok = torch.full((1, B), 1, dtype=torch.long)
        hid = (torch.zeros(1, B, H), torch.zeros(1, B, H)) if dec_cell == "LSTM" else torch.zeros(1, B, H)
        logits_all, h_all, c_all = [], [], []
        for t in range(31):
            lg, hid = dm(tok, hid, enc)
            tok = targets[t].view(1, -1)
            logits_all.append(lg.numpy().copy())
            h_all.append((hid[0] if dec_cell == "LSTM" else hid)[0].numpy().copy())
            if dec_cell == "LSTM":
                c_all.append(hid[1][0].numpy().copy())
            if t == 30 or not bool(masks[t + 1].any()):
                break
    if full:
        out["step_logits"] = np.stack(logits_all)
    out["step_h"] = np.stack(h_all)
    if c_all:
        out["step_c"] = np.stack(c_all)

    # ---- train steps exactly as train.py:248-273
    import random
    if py_seed is not None:
        random.seed(py_seed)
        out["meta_py_seed"] = np.array(py_seed)
        out["meta_tf_ratio"] = np.array(tf_ratio, dtype=np.float64)
        out["meta_tf_pattern"] = np.array([int(random.random() <= tf_ratio) for _ in range(n_steps)], dtype=np.int64)
        random.seed(py_seed)
    for it in range(n_steps):
        dm.train(train_mode)
        st.reset(drop_seed + it)
        dl, hiddens, fed = ref_train.forward_decoder(dec, enc, targets, masks, tf_ratio)
        if py_seed is not None and not int(out["meta_tf_pattern"][it]):
            out["output_indices_step%d" % it] = fed.numpy().astype(np.int64)
        rl = None
        if rec:
            rec["model"].train(train_mode)
            rl = fwd_rec(hiddens, enc, rec)
        loss = dl + 1.0 * rl if rec else dl
        dec["optimizer"].zero_grad()
        if rec:
            rec["optimizer"].zero_grad()
        loss.backward()
        if it == 0:
            out["T"] = np.array(hiddens.shape[0])
            out["hiddens"] = hiddens.detach().numpy().copy()
            out["dec_loss"] = np.array(dl.item(), dtype=np.float64)
            dreg = sum(torch.norm(p).item() for p in dm.parameters())
            out["dec_reg"] = np.array(dreg)
            out["dec_ce"] = np.array(dl.item() - 1e-3 * dreg)
            if rec:
                rreg = sum(torch.norm(p).item() for p in rec["model"].parameters())
                out["rec_loss"] = np.array(rl.item(), dtype=np.float64)
                out["rec_reg"] = np.array(rreg)
                out["rec_mse"] = np.array(rl.item() - 1e-2 * rreg)
            for k, p in dm.named_parameters():
                g = p.grad.detach().numpy()
                out["dec_gnorm/" + k] = np.array(np.linalg.norm(g.astype(np.float64)))
                if full:
                    out["dec_grad/" + k] = g.copy()
                else:
                    out["dec_gslice/" + k] = g.reshape(-1)[:64].copy()
            if rec:
                for k, p in rec["model"].named_parameters():
                    g = p.grad.detach().numpy()
                    out["rec_gnorm/" + k] = np.array(np.linalg.norm(g.astype(np.float64)))
                    if full:
                        out["rec_grad/" + k] = g.copy()
                    else:
                        out["rec_gslice/" + k] = g.reshape(-1)[:64].copy()
        gn = torch.nn.utils.clip_grad_norm_(dm.parameters(), C.gradient_clip)
        if it == 0:
            out["dec_grad_norm"] = np.array(float(gn))
        dec["optimizer"].step()
        if rec:
            rec["optimizer"].step()
        out["loss_step%d" % it] = np.array(loss.item(), dtype=np.float64)
        out["dec_loss_step%d" % it] = np.array(dl.item(), dtype=np.float64)
        if rec:
            out["rec_loss_step%d" % it] = np.array(rl.item(), dtype=np.float64)
        if it == 0 and full:
            for k, v in dm.state_dict().items():
                out["dec_after1/" + k] = v.detach().numpy().copy()
            if rec:
                for k, v in rec["model"].state_dict().items():
                    out["rec_after1/" + k] = v.detach().numpy().copy()
    if full:
        for k, v in dm.state_dict().items():
            out["dec_after%d/" % n_steps + k] = v.detach().numpy().copy()
        if rec:
            for k, v in rec["model"].state_dict().items():
                out["rec_after%d/" % n_steps + k] = v.detach().numpy().copy()
        names = [k for k, _ in dm.named_parameters()]
        for k, p in zip(names, dm.parameters()):
            s = dec["optimizer"].state[p]
            out["dec_opt/exp_avg/" + k] = s["exp_avg"].numpy().copy()
            out["dec_opt/exp_avg_sq/" + k] = s["exp_avg_sq"].numpy().copy()
            if "max_exp_avg_sq" in s:
                out["dec_opt/max_exp_avg_sq/" + k] = s["max_exp_avg_sq"].numpy().copy()
        if rec and (dec_lr is not None or rec_lr is not None):      # (the older cases keep their files byte for byte)
            for k, p in rec["model"].named_parameters():
                s = rec["optimizer"].state[p]
                out["rec_opt/exp_avg/" + k] = s["exp_avg"].numpy().copy()
                out["rec_opt/exp_avg_sq/" + k] = s["exp_avg_sq"].numpy().copy()
    else:
        for k, v in dm.state_dict().items():
            out["dec_pnorm_after%d/" % n_steps + k] = np.array(np.linalg.norm(v.numpy().astype(np.float64)))
            out["dec_pslice_after%d/" % n_steps + k] = v.numpy().reshape(-1)[:64].copy()
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print("%-28s T=%2d dec_loss=%.6f rec_loss=%s  %6.1f KB" % (
        name, int(out["T"]), float(out["dec_loss"]), "%.6f" % float(out["rec_loss"]) if rec else "-",
        os.path.getsize(path) / 1024))


def run_free_case(name, *, B, F, D, V, E, H, A, lens, dec_cell="LSTM", rec_kind=None, rec_cell="LSTM", RA=None, seed=0,
                  out_scale=6.0):
    """The validation pass of train.py:310-340: model.eval(), forward_decoder with the default teacher_forcing_ratio
    (0 -> free running, arg-max fed back), then the reconstructor on the free-running hidden states."""
    if ONLY is not None and name not in ONLY:
        return
    RA = RA or A
    configure(B=B, F=F, D=D, V=V, E=E, H=H, A=A, dec_cell=dec_cell, rec_kind=rec_kind, rec_cell=rec_cell, RA=RA)
    torch.manual_seed(seed)
    dec = ref_train.build_decoder(V)
    rec = ref_train.build_reconstructor() if rec_kind else None
    dm = dec["model"]
    with torch.no_grad():       # a decisive vocabulary projection, so the arg-max is not a coin flip at default init
        dm.out.weight.mul_(out_scale)
        dm.out.bias.mul_(out_scale)
    enc, targets = make_batch(B, F, D, V, lens, seed + 100)
    masks = targets > 0
    out = {"meta_dims": np.array([B, F, D, V, E, H, A, RA], dtype=np.int64), "meta_lens": np.array(lens, dtype=np.int64),
           "meta_cells": np.array([int(dec_cell == "GRU"), int(rec_cell == "GRU")], dtype=np.int64),
           "meta_formula_seed": np.array(-1), "enc": enc.numpy(), "targets": targets.numpy()}
    for k, v in dm.state_dict().items():
        out["dec_init/" + k] = v.detach().numpy().copy()
    if rec:
        for k, v in rec["model"].state_dict().items():
            out["rec_init/" + k] = v.detach().numpy().copy()
    dm.eval()
    with torch.no_grad():
        dl, hiddens, idx = ref_train.forward_decoder(dec, enc, targets, masks)       # train.py:327
        out["T"] = np.array(hiddens.shape[0])
        out["dec_loss"] = np.array(dl.item(), dtype=np.float64)
        out["hiddens"] = hiddens.numpy().copy()
        out["output_indices"] = idx.numpy().astype(np.int64)
        if rec:
            rec["model"].eval()
            fwd_rec = {"global": ref_train.forward_global_reconstructor, "local": ref_train.forward_local_reconstructor}[rec_kind]
            out["rec_loss"] = np.array(fwd_rec(hiddens, enc, rec).item(), dtype=np.float64)
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print("%-28s T=%2d dec_loss=%.6f distinct tokens %d, == targets %.2f  %6.1f KB" % (
        name, int(out["T"]), float(out["dec_loss"]), len(set(out["output_indices"].ravel().tolist())),
        float((out["output_indices"] == targets.numpy()[:int(out["T"])]).mean()), os.path.getsize(path) / 1024))


SMALL = dict(B=5, F=6, D=72, V=97, E=20, H=40, A=24)

if __name__ == "__main__":
    torch.set_num_threads(4)
    if "--only" in sys.argv:
        ONLY = set(sys.argv[sys.argv.index("--only") + 1].split(","))
    run_case("dec_eval", lens=[7, 3, 8, 1, 5], train_mode=False, **SMALL)
    run_case("dec_train", lens=[7, 3, 8, 1, 5], **SMALL)
    run_case("dec_T31", lens=[30, 4, 11, 2, 30], n_steps=1, **SMALL)
    run_case("dec_T4_samelen", lens=[3, 3, 3, 3, 3], n_steps=1, **SMALL)
    run_case("global_train", lens=[7, 3, 8, 1, 5], rec_kind="global", **SMALL)
    run_case("global_eval", lens=[6, 9, 2, 4, 4], rec_kind="global", train_mode=False, n_steps=1, **SMALL)
    run_case("local_train", lens=[7, 3, 8, 1, 5], rec_kind="local", RA=16, **SMALL)
    run_case("local_eval", lens=[6, 9, 2, 4, 4], rec_kind="local", RA=16, train_mode=False, n_steps=1, **SMALL)
    run_case("local_T31", lens=[30, 4, 11, 2, 9], rec_kind="local", RA=16, n_steps=1, **SMALL)
    run_case("gru_local_train", lens=[7, 3, 8, 1, 5], dec_cell="GRU", rec_kind="local", rec_cell="GRU", RA=16,
             n_steps=1, **SMALL)
    run_case("gru_global_eval", lens=[5, 2, 6, 3, 1], dec_cell="GRU", rec_kind="global", rec_cell="LSTM",
             train_mode=False, n_steps=1, **SMALL)
    # GRU cells (config.py:31 default decoder_model): the gru_* cases above plus 3-step runs
    run_case("gru_dec_train", lens=[7, 3, 8, 1, 5], dec_cell="GRU", **SMALL)
    run_case("gru_global_train", lens=[7, 3, 8, 1, 5], dec_cell="GRU", rec_kind="global", rec_cell="GRU", **SMALL)
    run_case("gru_local_train3", lens=[6, 9, 2, 4, 4], dec_cell="LSTM", rec_kind="local", rec_cell="GRU", RA=16, **SMALL)
    # validation pass (free-running decoder, eval mode): SURVEY.md §8f-4
    run_free_case("free_dec", lens=[7, 3, 8, 1, 5], **SMALL)
    run_free_case("free_global", lens=[6, 9, 2, 4, 4], rec_kind="global", **SMALL)
    run_free_case("free_local_gru", lens=[30, 4, 11, 2, 9], dec_cell="GRU", rec_kind="local", rec_cell="GRU", RA=16, **SMALL)
    # decoder_teacher_forcing_ratio < 1 in TRAINING (config.py:71, train.py:38,251): four iterations, the draw of each from
    # random.seed(py_seed) — teacher-forced and free-running iterations mixed, every one differentiated and stepped
    run_case("tf_half_global", lens=[6, 9, 2, 4, 4], rec_kind="global", n_steps=4, tf_ratio=0.5, py_seed=3, out_scale=6.0, **SMALL)
    run_case("tf_half_local", lens=[7, 3, 8, 1, 5], rec_kind="local", RA=16, n_steps=4, tf_ratio=0.5, py_seed=10, out_scale=6.0, **SMALL)
    # the update path the benchmark runs (GraphedStep, split reconstructor update, Adam in the GEMM epilogue): shapes the persistent
    # chains take (H % 32 == 0, R % 32 == 0), four iterations at learning rates of 1e-2 (config.py:86-87 attributes) — every update
    # moves every parameter by ~1e-2, so a missing update, a stale operand image or a wrong transpose is far outside any tolerance;
    # the reconstructor's Adam moments are stored as well
    CHAIN = dict(B=24, F=6, D=64, V=61, E=16, H=32, A=16)
    lens24 = [30, 4, 11, 2, 9, 7, 3, 8, 1, 5, 6, 9, 2, 4, 4, 12, 7, 21, 4, 9, 16, 5, 13, 10]
    run_case("lr_global_chain", lens=lens24, rec_kind="global", n_steps=4, dec_lr=1e-2, rec_lr=1e-2, **CHAIN)
    run_case("lr_local_chain", lens=lens24, rec_kind="local", RA=16, n_steps=4, dec_lr=1e-2, rec_lr=1e-2, **CHAIN)
    run_case("lr_gru_global_chain", lens=lens24, dec_cell="GRU", rec_kind="global", rec_cell="GRU", n_steps=4, dec_lr=1e-2, rec_lr=1e-2, **CHAIN)
    # full-shape cases (SURVEY.md §8a C1..C3 dims): parameters from formula_params(seed) so they can be
    # regenerated without the reference; only outputs / norms / slices are stored.
    FULL = dict(F=28, D=1536, V=4188, E=468, H=512, A=128)
    lens8 = [30, 12, 7, 21, 4, 9, 16, 5]
    run_case("full_dec_B8", B=8, lens=lens8, n_steps=1, full=False, formula_seed=7, **FULL)
    run_case("full_global_B8", B=8, lens=lens8, rec_kind="global", n_steps=1, full=False, formula_seed=7, **FULL)
    run_case("full_local_B8", B=8, lens=lens8, rec_kind="local", n_steps=1, full=False, formula_seed=7, **FULL)
    run_case("full_gru_global_B8", B=8, lens=lens8, dec_cell="GRU", rec_kind="global", rec_cell="GRU", n_steps=1,
             full=False, formula_seed=7, **FULL)
    run_case("full_gru_local_B8", B=8, lens=lens8, dec_cell="GRU", rec_kind="local", rec_cell="GRU", n_steps=1,
             full=False, formula_seed=7, **FULL)
