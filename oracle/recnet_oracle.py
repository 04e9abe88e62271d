"""TEST INFRASTRUCTURE ONLY — CPU oracle for the RecNet train-step hot path.

An as-written restatement, in plain fp32 PyTorch-CPU tensor ops, of the algorithm
the reference executes on its hot path.  It keeps the reference's
operation order, its per-time-step structure and its loop-invariant recomputation,
so that it doubles as the CPU baseline ("port") that bench.py times.  Gradients
come from torch autograd over these ops, exactly as the reference's
`loss.backward()` (train.py:268) does.

Parity status: PINNED against the reference itself — tests/golden/make_golden.py
imports the reference modules from /root/reference (in the build container only),
runs them on seeded inputs and commits the results under tests/golden/;
tests/test_oracle_golden.py checks this file against those vectors.  The reference
has no tests of its own.

Nothing in the product package imports this file.  Each function cites the
reference lines it restates.  Parameters are passed as plain dicts keyed by the
reference's state_dict names.
"""
import math

import numpy as np
import torch

from . import dropmask

PAD, SOS, EOS = 0, 1, 2          # config.py:56


class Dropper:
    """Dropout source.  mode 'eval': identity.  mode 'hash': the product's counter-based
    masks (oracle/dropmask.py).  mode 'rng': torch's global RNG (what the reference does;
    used only for timing the CPU baseline)."""

    def __init__(self, mode="eval", seed=0, B_global=None, b_offset=0):
        assert mode in ("eval", "hash", "rng")
        self.mode, self.seed, self.B_global, self.b_offset = mode, seed, B_global, b_offset

    def __call__(self, x, p, site, t):
        if self.mode == "eval" or p <= 0.0:
            return x
        if self.mode == "rng":
            return torch.nn.functional.dropout(x, p, True)
        B, N = x.shape[-2], x.shape[-1]
        Bg = self.B_global if self.B_global is not None else B
        m = dropmask.keep_mask(self.seed, site, t, Bg, N, p, self.b_offset, B)
        return x * torch.from_numpy(m).view(x.shape)


# ----------------------------------------------------------------------------- cells
def lstm_cell(x, h, c, w_ih, w_hh, b_ih, b_hh):
    """torch.nn.LSTM single layer, seq_len 1 (decoder.py:36-40,66): gate order i,f,g,o,
    both bias vectors added."""
    gates = x @ w_ih.t() + b_ih + h @ w_hh.t() + b_hh
    i, f, g, o = gates.chunk(4, dim=1)
    i, f, g, o = torch.sigmoid(i), torch.sigmoid(f), torch.tanh(g), torch.sigmoid(o)
    c2 = f * c + i * g
    h2 = o * torch.tanh(c2)
    return h2, c2


def gru_cell(x, h, w_ih, w_hh, b_ih, b_hh):
    """torch.nn.GRU single layer, seq_len 1 (decoder.py:34-40): gate order r,z,n."""
    gi = x @ w_ih.t() + b_ih
    gh = h @ w_hh.t() + b_hh
    ir, iz, in_ = gi.chunk(3, dim=1)
    hr, hz, hn = gh.chunk(3, dim=1)
    r = torch.sigmoid(ir + hr)
    z = torch.sigmoid(iz + hz)
    n = torch.tanh(in_ + r * hn)
    return (1.0 - z) * n + z * h


RNN_IMPL = "explicit"     # "aten": the fused single-step op nn.LSTM / nn.GRU dispatch to (what the reference executes on
#                           the CPU: mkldnn_rnn_layer) — same arithmetic, used when the oracle is TIMED as the CPU baseline


def _rnn(P, prefix, cell, x, hidden):
    w = [P[prefix + k] for k in ("weight_ih_l0", "weight_hh_l0", "bias_ih_l0", "bias_hh_l0")]
    if RNN_IMPL == "aten":
        if cell == "LSTM":
            out, h2, c2 = torch._VF.lstm(x.unsqueeze(0), hidden, w, True, 1, 0.0, False, False, False)
            return out[0], (h2, c2)
        out, h2 = torch._VF.gru(x.unsqueeze(0), hidden, w, True, 1, 0.0, False, False, False)
        return out[0], h2
    if cell == "LSTM":
        h, c = hidden
        h2, c2 = lstm_cell(x, h[0], c[0], *w)
        return h2, (h2.unsqueeze(0), c2.unsqueeze(0))
    h2 = gru_cell(x, hidden[0], *w)
    return h2, h2.unsqueeze(0)


def _last_h(hidden, cell):
    return hidden[0][-1] if cell == "LSTM" else hidden[-1]


# ----------------------------------------------------------------------------- decoder
ATTN_NORMALIZE = "none"   # "softmax": the product's opt-in normalisation of the decoder's attention energies over the frames
#                           (the reference constructs nn.Softmax(dim=1) at decoder.py:30 and never calls it)


def decoder_step(P, tok, hidden, enc, *, cell="LSTM", emb_scale=1.0, p_emb=0.5, p_out=0.5,
                 drop=None, t=0):
    """Decoder.forward, models/decoder.py:45-70.  tok [1,B] int64; hidden (h,c) each [1,B,H]
    (or a single tensor for GRU); enc [B,F,D].  Returns logits [B,V], hidden."""
    drop = drop or Dropper("eval")
    emb = P["embedding.weight"][tok[0]] * emb_scale                       # :46-47
    emb = drop(emb, p_emb, dropmask.SITE_DEC_EMBED, t)                   # :48
    Wh = _last_h(hidden, cell) @ P["attn_W.weight"].t()                  # :50-53
    Uv = enc @ P["attn_U.weight"].t()                                    # :54 (recomputed each step)
    al = torch.tanh(Wh.unsqueeze(1) + Uv + P["attn_b"])                  # :55-57
    al = al @ P["attn_w.weight"].t()                                     # :58  [B,F,1]  (no softmax)
    if ATTN_NORMALIZE == "softmax":
        al = torch.softmax(al, dim=1)
    ctx = (al * enc).mean(dim=1)                                         # :59-61
    x = torch.cat((emb, ctx), dim=1)                                     # :64
    out, hidden = _rnn(P, "rnn.", cell, x, hidden)                       # :66
    logits = out @ P["out.weight"].t() + P["out.bias"]                   # :68
    logits = drop(logits, p_out, dropmask.SITE_DEC_LOGIT, t)             # :69
    return logits, hidden


def zero_hidden(B, H, cell, dtype=torch.float32):
    if cell == "LSTM":
        return (torch.zeros(1, B, H, dtype=dtype), torch.zeros(1, B, H, dtype=dtype))
    return torch.zeros(1, B, H, dtype=dtype)


def decode_len(target_masks, caption_max_len=30):
    """Number of decoder steps the reference runs: the loop exit at train.py:66."""
    for t in range(caption_max_len + 1):
        if t == caption_max_len or not bool(target_masks[t + 1].any()):
            return t + 1
    return caption_max_len + 1


def forward_decoder(P, enc, targets, target_masks, *, cell="LSTM", lambda_reg=1e-3,
                    caption_max_len=30, emb_scale=1.0, p_emb=0.5, p_out=0.5, drop=None,
                    teacher_forcing=True, global_counts=None, return_parts=False):
    """train.py:17-75.  targets [31,B] int64, target_masks [31,B] bool.
    Returns (loss, hiddens[T,1,B,H], output_indices).  `global_counts` (n_t list, N) lets a
    data-parallel shard use the global normalisers; None = local batch."""
    B = enc.shape[0]
    H = P["rnn.weight_hh_l0"].shape[1]
    tok = torch.full((1, B), SOS, dtype=torch.long)                      # :25
    hidden = zero_hidden(B, H, cell, enc.dtype)                          # :28-35
    ce_sum, n_totals, hiddens, out_idx = 0.0, 0, [], []
    for t in range(caption_max_len + 1):                                 # :41
        logits, hidden = decoder_step(P, tok, hidden, enc, cell=cell, emb_scale=emb_scale,
                                      p_emb=p_emb, p_out=p_out, drop=drop, t=t)
        if teacher_forcing:
            tok = targets[t].view(1, -1)                                 # :45
        else:
            top = logits.argmax(dim=1)                                   # :47-51
            tok = top.view(1, -1)
            out_idx.append(top)
        m = target_masks[t]
        if global_counts is None:
            if bool(m.any()):
                ce_sum = ce_sum + torch.nn.functional.cross_entropy(logits[m], targets[t][m])  # :54-56
            n_totals += int(m.sum())                                     # :57,60
        else:
            n_t = global_counts[0][t]
            if bool(m.any()):
                ce_sum = ce_sum + torch.nn.functional.cross_entropy(
                    logits[m], targets[t][m], reduction="sum") / n_t
        hiddens.append(hidden[0] if cell == "LSTM" else hidden)          # :61-64
        if global_counts is None:
            stop = t == caption_max_len or not bool(target_masks[t + 1].any())   # :66
        else:
            stop = t + 1 == len(global_counts[0])
        if stop:
            break
    N = n_totals if global_counts is None else global_counts[1]
    ce = ce_sum / N                                                      # :68
    reg = sum(torch.norm(P[k]) for k in decoder_param_order(P))          # :69
    loss = ce + lambda_reg * reg                                         # :70
    hiddens = torch.stack(hiddens)                                       # :73
    idx = torch.stack(out_idx) if out_idx else torch.zeros(0, dtype=torch.long)
    if return_parts:
        return loss, hiddens, idx, ce, reg
    return loss, hiddens, idx


def decoder_param_order(P):
    """nn.Module.parameters() order of the reference Decoder (models/decoder.py:22-42)."""
    return [k for k in ("attn_b", "embedding.weight", "attn_W.weight", "attn_U.weight", "attn_w.weight",
                        "rnn.weight_ih_l0", "rnn.weight_hh_l0", "rnn.bias_ih_l0", "rnn.bias_hh_l0",
                        "out.weight", "out.bias") if k in P]


# ----------------------------------------------------------------------------- global reconstructor
def global_rec_step(P, inp, hidden, dec_hiddens, *, cell="LSTM", caption_max_len=30, p_drop=0.5,
                    drop=None, t=0):
    """GlobalReconstructor.forward, models/global_reconstructor.py:30-46.
    inp [L,B,H] (= dec_hiddens[t]); dec_hiddens [T,L,B,H]."""
    drop = drop or Dropper("eval")
    T = dec_hiddens.shape[0]                                             # :31
    mp = dec_hiddens.transpose(0, 2).transpose(1, 3)                     # :33-34  [B,H,T,L]
    mp = mp.mean(2).mean(2)                                              # :35-36  [B,H]
    mp = mp / T * caption_max_len                                        # :37
    mp = drop(mp, p_drop, dropmask.SITE_REC_INPUT, t)                    # :38
    x = torch.cat((inp[0], mp), dim=1)                                   # :40
    out, hidden = _rnn(P, "rnn.", cell, x, hidden)                       # :43
    out = out @ P["out.weight"].t() + P["out.bias"]                      # :45
    return out, hidden


def rec_param_order(P):
    """parameters() order of Global/LocalReconstructor."""
    return [k for k in ("attn_b", "attn_W.weight", "attn_U.weight", "attn_w.weight",
                        "rnn.weight_ih_l0", "rnn.weight_hh_l0", "rnn.bias_ih_l0", "rnn.bias_hh_l0",
                        "out.weight", "out.bias") if k in P]


def forward_global_reconstructor(P, dec_hiddens, enc, *, cell="LSTM", lambda_reg=1e-2,
                                 caption_max_len=30, p_drop=0.5, drop=None, mse_count=None,
                                 return_parts=False):
    """train.py:78-105.  `mse_count` = global B*R for data-parallel shards (None = local mean)."""
    B = enc.shape[0]
    R = P["rnn.weight_hh_l0"].shape[1]
    hidden = zero_hidden(B, R, cell, enc.dtype)                          # :82-89
    outs = []
    T = dec_hiddens.shape[0]                                             # :92
    for t in range(T):                                                   # :93
        o, hidden = global_rec_step(P, dec_hiddens[t], hidden, dec_hiddens, cell=cell,
                                    caption_max_len=caption_max_len, p_drop=p_drop, drop=drop, t=t)
        outs.append(o)
    outs = torch.stack(outs).mean(0)                                     # :96-98
    encm = enc.mean(1)                                                   # :99
    if mse_count is None:
        mse = torch.nn.functional.mse_loss(outs, encm)                   # :101
    else:
        mse = ((outs - encm) ** 2).sum() / mse_count
    mse = mse / T                                                        # :102
    reg = sum(torch.norm(P[k]) for k in rec_param_order(P))              # :103
    loss = mse + lambda_reg * reg                                        # :104
    if return_parts:
        return loss, mse, reg
    return loss


# ----------------------------------------------------------------------------- local reconstructor
def local_rec_step(P, hidden, dec_hiddens, *, cell="LSTM", p_drop=0.5, drop=None, t=0):
    """LocalReconstructor.forward, models/local_reconstructor.py:37-55.  dec_hiddens [T,L,B,H]."""
    drop = drop or Dropper("eval")
    Wh = _last_h(hidden, cell) @ P["attn_W.weight"].t()                  # :38-41
    Uv = dec_hiddens @ P["attn_U.weight"].t()                            # :42 (recomputed each step)
    be = torch.tanh(Wh.unsqueeze(0).unsqueeze(0) + Uv + P["attn_b"])     # :43-45
    be = be @ P["attn_w.weight"].t()                                     # :46  [T,L,B,1] (no softmax)
    x = (be * dec_hiddens).mean(dim=0)                                   # :47-49  [L,B,H]
    x = drop(x, p_drop, dropmask.SITE_REC_INPUT, t)                      # :50
    out, hidden = _rnn(P, "rnn.", cell, x[0], hidden)                    # :52 (L == 1)
    out = out @ P["out.weight"].t() + P["out.bias"]                      # :54
    return out, hidden


def forward_local_reconstructor(P, dec_hiddens, enc, *, cell="LSTM", lambda_reg=1e-2, p_drop=0.5,
                                drop=None, mse_count=None, return_parts=False):
    """train.py:108-131.  `mse_count` = global B*F*D for data-parallel shards."""
    B, F, _ = enc.shape
    R = P["rnn.weight_hh_l0"].shape[1]
    hidden = zero_hidden(B, R, cell, enc.dtype)                          # :112-119
    outs = []
    for t in range(F):                                                   # :122
        o, hidden = local_rec_step(P, hidden, dec_hiddens, cell=cell, p_drop=p_drop, drop=drop, t=t)
        outs.append(o)
    outs = torch.stack(outs).transpose(0, 1)                             # :125-127  [B,F,R]
    if mse_count is None:
        mse = torch.nn.functional.mse_loss(outs, enc)                    # :128
    else:
        mse = ((outs - enc) ** 2).sum() / mse_count
    reg = sum(torch.norm(P[k]) for k in rec_param_order(P))              # :129
    loss = mse + lambda_reg * reg                                        # :130
    if return_parts:
        return loss, mse, reg
    return loss


# ----------------------------------------------------------------------------- parameters
def init_decoder_params(V, E=468, H=512, A=128, D=1536, cell="LSTM", gen=None):
    """Same distributions as the reference's default initialisers:
    Embedding N(0,1); Linear U(+-1/sqrt(fan_in)); LSTM/GRU U(+-1/sqrt(H)); attn_b = 1
    (decoder.py:27).  Values are NOT the reference's RNG stream; goldens carry their own params."""
    G = 4 if cell == "LSTM" else 3
    u = lambda *s, k: (torch.rand(*s, generator=gen) * 2 - 1) * k
    P = {
        "attn_b": torch.ones(A),
        "embedding.weight": torch.randn(V, E, generator=gen),
        "attn_W.weight": u(A, H, k=1 / math.sqrt(H)),
        "attn_U.weight": u(A, D, k=1 / math.sqrt(D)),
        "attn_w.weight": u(1, A, k=1 / math.sqrt(A)),
        "rnn.weight_ih_l0": u(G * H, E + D, k=1 / math.sqrt(H)),
        "rnn.weight_hh_l0": u(G * H, H, k=1 / math.sqrt(H)),
        "rnn.bias_ih_l0": u(G * H, k=1 / math.sqrt(H)),
        "rnn.bias_hh_l0": u(G * H, k=1 / math.sqrt(H)),
        "out.weight": u(V, H, k=1 / math.sqrt(H)),
        "out.bias": u(V, k=1 / math.sqrt(H)),
    }
    return P


def init_rec_params(kind, H=512, R=1536, A=128, cell="LSTM", gen=None):
    G = 4 if cell == "LSTM" else 3
    u = lambda *s, k: (torch.rand(*s, generator=gen) * 2 - 1) * k
    P = {}
    if kind == "local":
        P["attn_b"] = torch.ones(A)
        P["attn_W.weight"] = u(A, R, k=1 / math.sqrt(R))
        P["attn_U.weight"] = u(A, H, k=1 / math.sqrt(H))
        P["attn_w.weight"] = u(1, A, k=1 / math.sqrt(A))
    inp = H if kind == "local" else 2 * H
    P["rnn.weight_ih_l0"] = u(G * R, inp, k=1 / math.sqrt(R))
    P["rnn.weight_hh_l0"] = u(G * R, R, k=1 / math.sqrt(R))
    P["rnn.bias_ih_l0"] = u(G * R, k=1 / math.sqrt(R))
    P["rnn.bias_hh_l0"] = u(G * R, k=1 / math.sqrt(R))
    P["out.weight"] = u(R, R, k=1 / math.sqrt(R))
    P["out.bias"] = u(R, k=1 / math.sqrt(R))
    return P


def synthetic_batch(B, F, D, V, seed=1234, caption_max_len=30, full_length=True, min_len=4):
    """SURVEY.md §8d synthetic inputs: randn features; word counts U{min_len..30} with len_0 = 30
    (so T = 31) when full_length; tokens U{3..V-1}; <EOS>=2 after the last word; zeros after."""
    g = torch.Generator().manual_seed(seed)
    enc = torch.randn(B, F, D, generator=g)
    lens = torch.randint(min_len, caption_max_len + 1, (B,), generator=g)
    if full_length:
        lens[0] = caption_max_len
    targets = torch.zeros(caption_max_len + 1, B, dtype=torch.long)
    for b in range(B):
        L = int(lens[b])
        targets[:L, b] = torch.randint(3, V, (L,), generator=g)
        targets[L, b] = EOS
    return enc, targets, targets > PAD


# ----------------------------------------------------------------------------- train step
class TrainState:
    """Parameters as autograd leaves + the two torch.optim.Adam instances of
    build_decoder / build_reconstructor (train.py:148-150,185-187)."""

    def __init__(self, dec_P, rec_P=None, rec_kind=None, *, cell="LSTM", rec_cell="LSTM",
                 dec_lr=1e-5, rec_lr=1e-6, dec_wd=1e-5, rec_wd=1e-5, dec_amsgrad=True,
                 rec_amsgrad=False, clip=50.0, lambda_recon=1.0, dec_lambda_reg=1e-3,
                 rec_lambda_reg=1e-2, caption_max_len=30, p_emb=0.5, p_out=0.5, p_rec=0.5):
        self.dec = {k: v.clone().requires_grad_(True) for k, v in dec_P.items()}
        self.rec = None if rec_P is None else {k: v.clone().requires_grad_(True) for k, v in rec_P.items()}
        self.rec_kind, self.cell, self.rec_cell = rec_kind, cell, rec_cell
        self.clip, self.lambda_recon = clip, lambda_recon
        self.dec_lambda_reg, self.rec_lambda_reg = dec_lambda_reg, rec_lambda_reg
        self.caption_max_len, self.p_emb, self.p_out, self.p_rec = caption_max_len, p_emb, p_out, p_rec
        self.dec_opt = torch.optim.Adam([self.dec[k] for k in decoder_param_order(self.dec)], lr=dec_lr,
                                        weight_decay=dec_wd, amsgrad=dec_amsgrad)
        self.rec_opt = None
        if self.rec is not None:
            self.rec_opt = torch.optim.Adam([self.rec[k] for k in rec_param_order(self.rec)], lr=rec_lr,
                                            weight_decay=rec_wd, amsgrad=rec_amsgrad)

    def losses(self, enc, targets, masks, drop=None, teacher_forcing=True):
        dl, hid, idx, ce, _ = forward_decoder(self.dec, enc, targets, masks, cell=self.cell,
                                              lambda_reg=self.dec_lambda_reg,
                                              caption_max_len=self.caption_max_len, p_emb=self.p_emb,
                                              p_out=self.p_out, drop=drop, return_parts=True,
                                              teacher_forcing=teacher_forcing)
        self.last_output_indices = idx            # (train.py:50: the tokens a free-running pass fed back)
        rl = mse = None
        if self.rec is not None:
            if self.rec_kind == "global":
                rl, mse, _ = forward_global_reconstructor(self.rec, hid, enc, cell=self.rec_cell,
                                                          lambda_reg=self.rec_lambda_reg,
                                                          caption_max_len=self.caption_max_len,
                                                          p_drop=self.p_rec, drop=drop, return_parts=True)
            else:
                rl, mse, _ = forward_local_reconstructor(self.rec, hid, enc, cell=self.rec_cell,
                                                         lambda_reg=self.rec_lambda_reg, p_drop=self.p_rec,
                                                         drop=drop, return_parts=True)
        return dl, rl, hid, ce, mse

    def step(self, enc, targets, masks, drop=None, teacher_forcing=True):
        """The train-step body, train.py:248-273.  Returns python floats (dec_loss, rec_loss, total,
        decoder grad-norm before clipping).  teacher_forcing: the iteration's draw of train.py:38
        (`random.random() <= C.decoder_teacher_forcing_ratio`, made by the caller)."""
        dl, rl, _, _, _ = self.losses(enc, targets, masks, drop, teacher_forcing)
        loss = dl if rl is None else dl + self.lambda_recon * rl         # :259-262
        self.dec_opt.zero_grad()                                         # :265-267
        if self.rec_opt is not None:
            self.rec_opt.zero_grad()
        loss.backward()                                                  # :268
        gn = torch.nn.utils.clip_grad_norm_([self.dec[k] for k in decoder_param_order(self.dec)],
                                            self.clip)                   # :269-270
        self.dec_opt.step()                                              # :271
        if self.rec_opt is not None:
            self.rec_opt.step()                                          # :272-273
        return float(dl), (None if rl is None else float(rl)), float(loss), float(gn)
