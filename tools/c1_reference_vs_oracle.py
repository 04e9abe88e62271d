#!/usr/bin/env python3
"""BASELINE.md section 3 check, build container only (needs /root/reference): at BASELINE.json configs[0] (decoder only,
B=8, 28x1536, T=31, CPU) the as-written oracle (oracle/recnet_oracle.py — the CPU baseline bench.py times on the GPU box,
where the reference cannot travel) computes the same loss as the imported reference and takes the same time within noise.

    PYTHONDONTWRITEBYTECODE=1 python tools/c1_reference_vs_oracle.py > profiles/r02_c1_reference_vs_oracle.json
"""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
sys.path.insert(0, ROOT)
import make_golden as MG  # noqa: E402  (stubs + the reference's train / config modules)

from oracle import recnet_oracle as O  # noqa: E402

B, F, D, V = 8, 28, 1536, 4188
threads = int(os.environ.get("THREADS", "8"))
torch.set_num_threads(threads)
MG.configure(B=B, F=F, D=D, V=V, E=468, H=512, A=128, dec_cell="LSTM", rec_kind=None, rec_cell="LSTM", RA=128)
C, ref = MG.C, MG.ref_train
torch.manual_seed(0)
dec = ref.build_decoder(V)
enc, targets, masks = O.synthetic_batch(B, F, D, V)
P0 = {k: v.detach().clone() for k, v in dec["model"].state_dict().items()}
st = O.TrainState(P0, None, None)


def ref_step():
    dec["model"].train()
    loss, _, _ = ref.forward_decoder(dec, enc, targets, masks, 1.0)
    dec["optimizer"].zero_grad()
    loss.backward()
    torch.nn.utils.clip_grad_norm_(dec["model"].parameters(), C.gradient_clip)
    dec["optimizer"].step()
    return float(loss)


def ora_step():
    return st.step(enc, targets, masks, O.Dropper("rng"))[0]


def timed(fn, warm=2, n=7):
    ts = []
    for i in range(warm + n):
        t0 = time.perf_counter()
        fn()
        if i >= warm:
            ts.append(time.perf_counter() - t0)
    ts.sort()
    return ts[len(ts) // 2], ts[0], ts[-1]


# same loss in eval mode from the same parameters (dropout off): the restatement is the same arithmetic
dec["model"].eval()
with torch.no_grad():
    l_ref = float(ref.forward_decoder(dec, enc, targets, masks, 1.0)[0])
l_ora = float(O.forward_decoder(P0, enc, targets, masks, drop=O.Dropper("eval"))[0])
r_med, r_min, r_max = timed(ref_step)
e_med, e_min, e_max = timed(ora_step)               # explicit matmul cells (the form the parity tests use)
O.RNN_IMPL = "aten"                                   # the fused one-step RNN op the reference's nn.LSTM executes
st = O.TrainState(P0, None, None)
l_aten = float(O.forward_decoder(P0, enc, targets, masks, drop=O.Dropper("eval"))[0])
o_med, o_min, o_max = timed(ora_step)
print(json.dumps({"config": "BASELINE.json configs[0]: decoder only, B=8, 28x1536, V=4188, T=31, torch-CPU %d threads" % threads,
                  "eval_loss_reference": l_ref, "eval_loss_oracle": l_ora, "eval_loss_oracle_aten_rnn": l_aten,
                  "oracle_explicit_cells_s_per_step": {"median": round(e_med, 4), "min": round(e_min, 4), "max": round(e_max, 4)},
                  "reference_s_per_step": {"median": round(r_med, 4), "min": round(r_min, 4), "max": round(r_max, 4)},
                  "oracle_s_per_step": {"median": round(o_med, 4), "min": round(o_min, 4), "max": round(o_max, 4)},
                  "oracle_over_reference": round(o_med / r_med, 3),
                  "captions_per_s_reference": round(B / r_med, 2), "captions_per_s_oracle": round(B / o_med, 2)}, indent=1))
