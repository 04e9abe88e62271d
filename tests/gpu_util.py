"""Shared helpers for the -m gpu parity tests: build the product's models from golden / formula
parameters and compare against the CPU oracle."""
import numpy as np
import torch

import recnet_amd as R
from oracle import recnet_oracle as O
from tests import golden_util as GU

TOL = {
    # The parity bars of SURVEY.md §8d.  precision: loss rel, hidden-state abs, per-tensor gradient ||d||/||g||,
    # parameter abs after the optimiser steps.  Measured worst cases over the golden set (tools/parity_report.py):
    # f32 path: hidden 2.4e-7, loss 1.9e-7, gradient 6.8e-7; bf16 path: hidden 3.0e-3, loss 1.0e-5, gradient 7.3e-3.
    "f32": dict(loss=1e-5, hid=1e-5, grad=1e-4, param=2e-6, cos=0.999999),
    # bf16: 3 Adam steps of lr 1e-5: a sign flip of a tiny gradient moves a parameter by up to 2*lr per step
    "bf16": dict(loss=1e-3, hid=1e-2, grad=2e-2, param=6e-5, cos=0.9995),
}


def make_models(dims, kind, precision, decP, recP, device="cuda", batch=None, cells=("LSTM", "LSTM"), attn_normalize="none",
                **config):
    """config: further TrainConfig attributes (e.g. decoder_learning_rate, reconstructor_learning_rate)."""
    B, F, D, V, E, H, A, RA = dims
    C = R.make_config(batch_size=B, encoder_output_len=F, encoder_output_size=D, embedding_size=E,
                      decoder_hidden_size=H, decoder_attn_size=A, use_recon=kind is not None,
                      reconstructor_type=kind or "local", reconstructor_hidden_size=D,
                      reconstructor_attn_size=RA, precision=precision, device=device, decoder_model=cells[0],
                      reconstructor_model=cells[1], decoder_attn_normalize=attn_normalize, **config)
    dec = R.build_decoder(V, C)
    dec["model"].load_state_dict({k: v.clone() for k, v in decP.items()})
    rec = None
    if kind:
        rec = R.build_reconstructor(C)
        rec["model"].load_state_dict({k: v.clone() for k, v in recP.items()})
    return C, dec, rec


def load_case(name):
    g = GU.load(name)
    dims = [int(x) for x in g["meta_dims"]]
    B, F, D, V, E, H, A, RA = dims
    kind = "global" if "global" in name else ("local" if "local" in name else None)
    fs = int(g["meta_formula_seed"])
    cells = g["_cells"] = GU.cells_of(g)
    if fs >= 0:
        decP = GU.formula_params(GU.decoder_shapes(V, E, H, A, D, cells[0]), fs)
        recP = GU.formula_params(GU.rec_shapes(kind, H, D, RA, cells[1]), fs + 1) if kind else None
        enc, targets = GU.make_batch(B, F, D, V, g["meta_lens"], int(g["meta_batch_seed"]))
    else:
        decP, recP = GU.group(g, "dec_init"), (GU.group(g, "rec_init") if kind else None)
        enc, targets = torch.from_numpy(g["enc"]), torch.from_numpy(g["targets"])
    return g, dims, kind, decP, recP, enc, targets


def cosine(a, b):
    a = np.asarray(a, dtype=np.float64).ravel()
    b = np.asarray(b, dtype=np.float64).ravel()
    return float(a @ b / max(np.linalg.norm(a) * np.linalg.norm(b), 1e-300))


def rel_err(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-12))


def oracle_grads(decP, recP, kind, enc, targets, train, seed, B_global=None, b_offset=0, cells=("LSTM", "LSTM")):
    """Reference gradients (incl. the regulariser term) from the CPU oracle's autograd."""
    st = O.TrainState(decP, recP, kind, cell=cells[0], rec_cell=cells[1])
    drop = O.Dropper("hash", seed=seed) if train else O.Dropper("eval")
    masks = targets > 0
    dl, rl, hid, ce, mse = st.losses(enc, targets, masks, drop)
    loss = dl if rl is None else dl + rl
    loss.backward()
    out = dict(dec_loss=float(dl), dec_ce=float(ce), hiddens=hid.detach().numpy(),
               dec_grad={k: v.grad.numpy().copy() for k, v in st.dec.items()})
    if rl is not None:
        out.update(rec_loss=float(rl), rec_mse=float(mse),
                   rec_grad={k: v.grad.numpy().copy() for k, v in st.rec.items()})
    return out
