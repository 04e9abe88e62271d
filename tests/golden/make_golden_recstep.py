#!/usr/bin/env python3
"""Golden vectors for the reconstructors' PER-STEP API, produced by the reference's own modules and its own time loops
(train.py:82-94 global, train.py:112-123 local) on CPU.  Build container only (needs /root/reference):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_recstep.py

Parameters come from tests.golden_util.formula_params (regenerable anywhere), decoder_hiddens from a seeded generator;
stored: the per-step outputs and hidden states.  Plain arrays only.
"""
import os
import sys
sys.dont_write_bytecode = True      # the reference tree under /root/reference stays untouched (no __pycache__ beside its modules)

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as MG  # noqa: E402  (stubs + the reference's train / config modules, HashDropout)

from models.global_reconstructor import GlobalReconstructor  # noqa: E402  (the reference's)
from models.local_reconstructor import LocalReconstructor  # noqa: E402
from oracle import dropmask  # noqa: E402
from tests.golden_util import formula_params, rec_shapes  # noqa: E402

C = MG.C


def make_hiddens(T, B, H, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.tanh(torch.randn(T, 1, B, H, generator=g)) * 0.6


def run(name, kind, cell, *, B=5, T=8, F=6, H=40, R=72, RA=16, train_mode=False, seed=3, drop_seed=11):
    C.device = "cpu"; C.batch_size = B; C.reconstructor_model = cell; C.reconstructor_n_layers = 1
    C.reconstructor_hidden_size = R; C.encoder_output_len = F; C.caption_max_len = 30
    if kind == "global":
        m = GlobalReconstructor(model_name=cell, n_layers=1, decoder_hidden_size=H, hidden_size=R, dropout=0.5,
                                decoder_dropout=0.5, caption_max_len=30)
    else:
        m = LocalReconstructor(model_name=cell, n_layers=1, decoder_hidden_size=H, hidden_size=R, dropout=0.5,
                               decoder_dropout=0.5, attn_size=RA)
    m.load_state_dict(formula_params(rec_shapes(kind, H, R, RA, cell), seed))
    st = MG.DropState(); st.reset(drop_seed)
    m.decoder_dropout = MG.HashDropout(0.5, dropmask.SITE_REC_INPUT, st)
    m.train(train_mode)
    decoder_hiddens = make_hiddens(T, B, H, seed + 50)
    outs, hs, cs = [], [], []
    with torch.no_grad():
        # ---- train.py:82-89 / 112-119
        if cell == "LSTM":
            reconstructor_hidden = (torch.zeros(1, B, R), torch.zeros(1, B, R))
        else:
            reconstructor_hidden = torch.zeros(1, B, R)
        if kind == "global":
            for t in range(decoder_hiddens.size(0)):                       # train.py:92-94
                decoder_hidden = decoder_hiddens[t].to(C.device)
                reconstructor_output, reconstructor_hidden = m(decoder_hidden, reconstructor_hidden, decoder_hiddens)
                outs.append(reconstructor_output.numpy().copy())
                hs.append((reconstructor_hidden[0] if cell == "LSTM" else reconstructor_hidden)[0].numpy().copy())
                if cell == "LSTM":
                    cs.append(reconstructor_hidden[1][0].numpy().copy())
        else:
            for t in range(C.encoder_output_len):                          # train.py:122-123
                reconstructor_output, reconstructor_hidden = m(reconstructor_hidden, decoder_hiddens)
                outs.append(reconstructor_output.numpy().copy())
                hs.append((reconstructor_hidden[0] if cell == "LSTM" else reconstructor_hidden)[0].numpy().copy())
                if cell == "LSTM":
                    cs.append(reconstructor_hidden[1][0].numpy().copy())
    out = {"meta_dims": np.array([B, T, F, H, R, RA], dtype=np.int64), "meta_seed": np.array(seed),
           "meta_drop_seed": np.array(drop_seed), "meta_train_mode": np.array(int(train_mode)),
           "meta_gru": np.array(int(cell == "GRU")), "out": np.stack(outs), "h": np.stack(hs)}
    if cs:
        out["c"] = np.stack(cs)
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print("%-28s steps=%d |out|=%.5f  %.1f KB" % (name, len(outs), float(np.abs(out["out"]).mean()), os.path.getsize(path) / 1024))


if __name__ == "__main__":
    torch.set_num_threads(2)
    run("recstep_global_eval", "global", "LSTM")
    run("recstep_global_train", "global", "LSTM", train_mode=True)
    run("recstep_local_eval", "local", "LSTM")
    run("recstep_local_train", "local", "LSTM", train_mode=True)
    run("recstep_global_gru", "global", "GRU", train_mode=True)
    run("recstep_local_gru", "local", "GRU")
