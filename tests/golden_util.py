"""Helpers shared by tests/golden/make_golden.py (the generator) and the tests that consume the
vectors: deterministic synthetic batches and formula-defined parameters that can be regenerated
on any box without the reference."""
import os

import numpy as np
import torch

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def make_batch(B, F, D, V, lens, seed):
    g = torch.Generator().manual_seed(int(seed))
    enc = torch.randn(B, F, D, generator=g)
    targets = torch.zeros(31, B, dtype=torch.long)
    for b, L in enumerate(lens):
        L = int(L)
        targets[:L, b] = torch.randint(3, V, (L,), generator=g)
        targets[L, b] = 2
    return enc, targets


def formula_params(shapes, seed):
    """shapes: {state_dict key: shape}.  Keys visited in sorted order; U(-k,k), k = 1/sqrt(last dim)
    for matrices, 0.05 for vectors; embedding N(0,1)*0.5; attn_b ones."""
    g = torch.Generator().manual_seed(int(seed))
    out = {}
    for k in sorted(shapes.keys()):
        shp = tuple(shapes[k])
        if k == "attn_b":
            out[k] = torch.ones(shp)
        elif k == "embedding.weight":
            out[k] = torch.randn(shp, generator=g) * 0.5
        else:
            kk = 1.0 / np.sqrt(shp[-1]) if len(shp) > 1 else 0.05
            out[k] = (torch.rand(shp, generator=g) * 2 - 1) * kk
    return out


def decoder_shapes(V, E, H, A, D, cell="LSTM"):
    G = 4 if cell == "LSTM" else 3
    return {"attn_b": (A,), "embedding.weight": (V, E), "attn_W.weight": (A, H), "attn_U.weight": (A, D),
            "attn_w.weight": (1, A), "rnn.weight_ih_l0": (G * H, E + D), "rnn.weight_hh_l0": (G * H, H),
            "rnn.bias_ih_l0": (G * H,), "rnn.bias_hh_l0": (G * H,), "out.weight": (V, H), "out.bias": (V,)}


def rec_shapes(kind, H, R, A, cell="LSTM"):
    G = 4 if cell == "LSTM" else 3
    s = {}
    if kind == "local":
        s.update({"attn_b": (A,), "attn_W.weight": (A, R), "attn_U.weight": (A, H), "attn_w.weight": (1, A)})
    s.update({"rnn.weight_ih_l0": (G * R, H if kind == "local" else 2 * H), "rnn.weight_hh_l0": (G * R, R),
              "rnn.bias_ih_l0": (G * R,), "rnn.bias_hh_l0": (G * R,), "out.weight": (R, R), "out.bias": (R,)})
    return s


def cells_of(g):
    """(decoder cell, reconstructor cell) of a golden case; cases written before GRU support are LSTM/LSTM."""
    mc = g.get("meta_cells")
    if mc is None:
        return ("LSTM", "LSTM")
    return tuple("GRU" if int(x) else "LSTM" for x in mc)


def load(name):
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    return {k: z[k] for k in z.files}


def group(g, prefix):
    """{'key': tensor} for every entry 'prefix/key'."""
    n = len(prefix) + 1
    return {k[n:]: torch.from_numpy(np.array(v)) for k, v in g.items() if k.startswith(prefix + "/")}
