#!/usr/bin/env python3
"""Golden vectors for the inference search (eval.py:19-120), produced by running THE REFERENCE's own
greedy_search / beam_search on CPU (build container only; /root/reference never travels).

Recipe = SURVEY.md Appendix A for eval.py: stub dataset.MSVD and the coco_caption modules, set
torch.cuda.FloatTensor = torch.FloatTensor (eval.py:39,57 hard-code it) and eval.C.device = "cpu".
The decoder is the reference's models/decoder.py:Decoder in eval mode with formula-defined parameters
(tests/golden_util.formula_params), so the consumer can rebuild it without the reference.
Only plain arrays are written."""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference")


def _stub(name, **attrs):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


_stub("dataset.MSVD", MSVD=object)
for n in ("coco_caption", "coco_caption.pycocotools", "coco_caption.pycocoevalcap"):
    _stub(n)
_stub("coco_caption.pycocotools.msvd", MSVD=object)
_stub("coco_caption.pycocoevalcap.eval", COCOEvalCap=object)
_stub("coco_caption.pycocotools.utils", load_res=lambda *a: None)
torch.cuda.FloatTensor = torch.FloatTensor

import eval as ref_eval  # noqa: E402  (the reference's eval.py)
from models.decoder import Decoder as RefDecoder  # noqa: E402
from tests.golden_util import decoder_shapes, formula_params  # noqa: E402

ref_eval.C.device = "cpu"


class Cfg:
    pass


class Vocab:
    pass


def run(name, B, F, D, V, E, H, A, seed, beam_widths=(1, 3, 5), scale=1.0, eos_bias=0.0, pad_bias=0.0, cell="LSTM"):
    cfg = Cfg()
    cfg.caption_max_len, cfg.batch_size, cfg.decoder_model = 30, B, cell
    ref_eval.C.decoder_model = cell
    vocab = Vocab()
    vocab.word2idx = {"<PAD>": 0, "<SOS>": 1, "<EOS>": 2}
    vocab.n_vocabs = V
    dec = RefDecoder(cell, 1, D, E, 1, H, A, V, 0.5, 0.5, 0.5)
    P = formula_params(decoder_shapes(V, E, H, A, D, cell), seed)
    # make the vocabulary projection decisive enough that <EOS>/<PAD> actually occur and hypotheses differ
    P["out.weight"] = P["out.weight"] * scale
    P["out.bias"] = P["out.bias"] * scale
    P["out.bias"][2] += eos_bias      # make <EOS> (length normalisation branch, eval.py:52-55) ...
    P["out.bias"][0] += pad_bias      # ... and <PAD> (early stop, eval.py:30,116) actually occur
    dec.load_state_dict(P)
    dec.eval()
    g = torch.Generator().manual_seed(seed + 50)
    enc = torch.randn(B, F, D, generator=g)
    out = {"meta_dims": np.array([B, F, D, V, E, H, A], dtype=np.int64), "meta_seed": np.array(seed),
           "meta_scale": np.array(scale), "meta_eos_bias": np.array(eos_bias), "meta_pad_bias": np.array(pad_bias),
           "enc": enc.numpy(), "meta_cells": np.array([int(cell == "GRU"), 0], dtype=np.int64)}
    zero = (lambda: (torch.zeros(1, B, H), torch.zeros(1, B, H))) if cell == "LSTM" else (lambda: torch.zeros(1, B, H))
    with torch.no_grad():
        inp = torch.full((1, B), 1, dtype=torch.long)
        hid = zero()
        gi = ref_eval.greedy_search(cfg, dec, inp, hid, enc)
        out["greedy"] = np.array([[int(x) for x in row] for row in gi], dtype=np.int64)      # [n_steps][B]
        for bw in beam_widths:
            inp = torch.full((1, B), 1, dtype=torch.long)
            hid = zero()
            bo = ref_eval.beam_search(cfg, bw, vocab, dec, inp, hid, enc)                      # list over b of token lists
            out["beam%d" % bw] = np.array(bo, dtype=np.int64)                                  # [B][n_steps]
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print(name, "greedy steps", out["greedy"].shape[0], {k: v.shape for k, v in out.items() if k.startswith("beam")},
          "distinct greedy tokens", len(set(out["greedy"].ravel().tolist())), "EOS in greedy", int((out["greedy"] == 2).sum()),
          "EOS in beam5", int((out["beam5"] == 2).sum()))


if __name__ == "__main__":
    torch.set_num_threads(4)
    if "--gru-only" in sys.argv:
        run("search_gru", B=6, F=5, D=72, V=97, E=20, H=40, A=24, seed=6, scale=8.0, eos_bias=1.2, cell="GRU")
        run("search_gru_b", B=6, F=5, D=72, V=97, E=20, H=40, A=24, seed=11, scale=20.0, eos_bias=1.2, cell="GRU")
        run("search_gru_stop", B=4, F=6, D=64, V=61, E=16, H=32, A=16, seed=9, scale=4.0, pad_bias=9.0, cell="GRU")
        sys.exit(0)
    run("search_small", B=6, F=5, D=72, V=97, E=20, H=40, A=24, seed=5, scale=8.0)
    run("search_small_b", B=4, F=6, D=64, V=61, E=16, H=32, A=16, seed=9, scale=20.0)
    run("search_eos", B=6, F=5, D=72, V=97, E=20, H=40, A=24, seed=5, scale=8.0, eos_bias=1.2)
    run("search_stop", B=4, F=6, D=64, V=61, E=16, H=32, A=16, seed=9, scale=4.0, pad_bias=9.0)
    run("search_gru", B=6, F=5, D=72, V=97, E=20, H=40, A=24, seed=6, scale=8.0, eos_bias=1.2, cell="GRU")
    run("search_gru_b", B=6, F=5, D=72, V=97, E=20, H=40, A=24, seed=11, scale=20.0, eos_bias=1.2, cell="GRU")
    run("search_gru_stop", B=4, F=6, D=64, V=61, E=16, H=32, A=16, seed=9, scale=4.0, pad_bias=9.0, cell="GRU")
