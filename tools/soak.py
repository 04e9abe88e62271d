#!/usr/bin/env python3
"""Soak run of the hipGraph-replayed train step: many thousand steps per configuration in windows; after each window the
chain status word (recnet_chain_status: a bounded wait gave up / poisoned loss) and the losses are checked and the window's
mean step time recorded.  Evidence that the persistent chain kernels neither hang nor degrade over a long run.

   python tools/soak.py [steps per configuration = 30000] > soak.json"""
import json
import math
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import recnet_amd as R  # noqa: E402
from recnet_amd.synthetic import synthetic_features, synthetic_targets  # noqa: E402
from bench import build_models  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 30000
WIN = 2000
CONFIGS = [("c2", "global", 100, 28, 1536), ("c3", "local", 100, 28, 1536), ("c4", "local", 32, 40, 2048), ("c5", "local", 64, 28, 3584)]
V = 4188
out = {}
for name, kind, B, F, D in CONFIGS:
    C, dec, rec = build_models(R, dict(batch_size=B, use_recon=True, reconstructor_type=kind, encoder_output_len=F,
                                       encoder_output_size=D, reconstructor_hidden_size=D, precision="bf16", device="cuda:0"), V)
    tg = synthetic_targets(B, V, seed=1234)
    enc = synthetic_features(B, F, D, seed=1234).cuda()
    targets = tg.cuda()
    step = R.DataParallelTrainStep(dec, rec, B, 0, 1, n_frames=F)
    T, w = step.prepare(tg.numpy())
    runner = R.GraphedStep(step, enc, targets, T, w, defer_reconstructor_update="recurrent")      # the bench default (round 4): split update
    for _ in range(10):
        runner()
    torch.cuda.synchronize()
    eng = step.step_impl.engine
    wins, status_or, bad_loss, first, last = [], 0, 0, None, None
    t_all = time.perf_counter()
    for w0 in range(0, N, WIN):
        t0 = time.perf_counter()
        for _ in range(WIN):
            runner()
        runner.flush()
        torch.cuda.synchronize()
        wins.append((time.perf_counter() - t0) / WIN * 1e3)
        st = eng.chain_status()
        status_or |= st
        sc = eng.scalar_dict()
        if not math.isfinite(sc["total_loss"]):
            bad_loss += 1
        first = sc["total_loss"] if first is None else first
        last = sc["total_loss"]
    out[name] = {"steps": len(wins) * WIN, "seconds": round(time.perf_counter() - t_all, 1), "chain_status_or": status_or,
                 "windows_with_non_finite_loss": bad_loss, "ms_per_step_first_window": round(wins[0], 4),
                 "ms_per_step_min": round(min(wins), 4), "ms_per_step_max": round(max(wins), 4),
                 "ms_per_step_last_window": round(wins[-1], 4), "total_loss_after_first_window": round(first, 5),
                 "total_loss_at_end": round(last, 5)}
    del runner, step, dec, rec
    print(name, out[name], file=sys.stderr, flush=True)
print(json.dumps(out, indent=1))
