"""TEST INFRASTRUCTURE ONLY — CPU restatement of the reference's inference search (eval.py:19-120), on top of
oracle/recnet_oracle.py:decoder_step.  Pinned against golden vectors produced by the reference's own
greedy_search / beam_search (tests/golden/make_golden_search.py -> tests/golden/search_*.npz)."""
import numpy as np
import torch

from . import recnet_oracle as O


def greedy_search(P, enc, caption_max_len=30, cell="LSTM"):
    """eval.py:19-33 with the start state of eval.py:131-141.  Returns [n_steps][B] int64."""
    B = enc.shape[0]
    H = P["rnn.weight_hh_l0"].shape[1]
    tok = torch.full((1, B), O.SOS, dtype=torch.long)
    hid = O.zero_hidden(B, H, cell)
    out = []
    for t in range(caption_max_len + 1):
        logits, hid = O.decoder_step(P, tok, hid, enc, cell=cell, t=t)
        top = logits.argmax(dim=1)                                   # :24 topk(1)
        tok = top.view(1, -1)
        out.append(top.numpy().copy())
        if t == caption_max_len or bool((tok == 0).all()):           # :30
            break
    return np.stack(out)


def beam_search(P, enc, beam_width, caption_max_len=30, cell="LSTM"):
    """eval.py:36-120.  Returns the top-1 hypothesis per caption, [B][n_steps] int64."""
    B = enc.shape[0]
    H = P["rnn.weight_hh_l0"].shape[1]
    V = P["out.weight"].shape[0]
    toks = [torch.full((1, B), O.SOS, dtype=torch.long)]
    hids = [O.zero_hidden(B, H, cell)]
    cums = [torch.zeros(B)]                                          # :39-40 log(1)
    hist = np.zeros((1, B, 0), dtype=np.int64)                       # [beam][B][t]
    for t in range(caption_max_len + 1):
        scores, nxt = [], []
        for i in range(len(toks)):
            logits, nh = O.decoder_step(P, toks[i], hids[i], enc, cell=cell, t=t)
            nxt.append(nh)
            seq_len = np.full(B, t + 1, dtype=np.float64)            # :53-54
            for b in range(B):
                pos = np.where(hist[i, b] == O.EOS)[0]               # :52  (last <EOS> wins, :55)
                if len(pos):
                    seq_len[b] = pos[-1] + 1
            norm = torch.from_numpy((seq_len ** 0.7).astype(np.float32))   # :56-57
            cp = cums[i] / norm                                      # :59
            scores.append(torch.log(torch.sigmoid(logits)) + cp.unsqueeze(1))   # :61-62
        allsc = torch.cat(scores, dim=1)                             # :63
        vals, flat = allsc.topk(beam_width)                          # :64
        src, tok = (flat // V).numpy(), (flat % V).numpy()           # :68-69
        new_hist = np.zeros((beam_width, B, t + 1), dtype=np.int64)
        toks2, hids2, cums2 = [], [], []
        for k in range(beam_width):
            if cell == "LSTM":
                h = torch.stack([nxt[src[b, k]][0][0, b] for b in range(B)]).unsqueeze(0)   # :78-93
                c = torch.stack([nxt[src[b, k]][1][0, b] for b in range(B)]).unsqueeze(0)
                hids2.append((h, c))
            else:                                                    # :94-102, single hidden tensor
                hids2.append(torch.stack([nxt[src[b, k]][0, b] for b in range(B)]).unsqueeze(0))
            toks2.append(torch.from_numpy(tok[:, k].copy()).view(1, -1))
            cums2.append(vals[:, k].clone())                         # :75 (the length-normalised value is carried on)
            for b in range(B):
                new_hist[k, b, :t] = hist[src[b, k], b]
                new_hist[k, b, t] = tok[b, k]                        # :104-108
        toks, hids, cums, hist = toks2, hids2, cums2, new_hist
        if t == caption_max_len or all(bool((x == 0).all()) for x in toks):   # :116
            break
    return hist[0]                                                   # :119
