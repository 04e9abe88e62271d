#!/usr/bin/env python3
"""Liveness stress of the persistent chain kernels: many fresh engines over small shapes, a few steps each; after every
step the chain status word is read (recnet_chain_status: which chain gave up a bounded wait).  Prints every event."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import recnet_amd as R
from tests import golden_util as GU
from tests.gpu_util import make_models

SHAPES = {
    "H32_T2_local": ([70, 3, 32, 29, 8, 32, 16, 16], [i % 2 for i in range(70)], "local"),
    "R128_B65_F1_global": ([65, 1, 128, 29, 8, 32, 16, 12], [(5 * i) % 7 for i in range(65)], "global"),
    "H32_A16_global": ([100, 5, 48, 29, 8, 32, 16, 16], [(7 * i) % 9 for i in range(100)], "global"),
    "R64_B100_local": ([100, 4, 64, 29, 8, 32, 16, 24], [(7 * i) % 9 for i in range(100)], "local"),
    "R96_B40_local": ([40, 3, 96, 29, 8, 64, 16, 128], [(5 * i) % 8 for i in range(40)], "local"),
}
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 30
t0 = time.time()
events = 0
for r in range(rounds):
    for name, (dims, lens, kind) in SHAPES.items():
        B, F, D, V, E, H, A, RA = dims
        decP = GU.formula_params(GU.decoder_shapes(V, E, H, A, D), 31)
        recP = GU.formula_params(GU.rec_shapes(kind, H, D, RA), 32)
        enc, targets = GU.make_batch(B, F, D, V, lens, 78)
        C, dec, rec = make_models(dims, kind, "bf16", decP, recP)
        step = R.TrainStep(dec, rec)
        T, w = step.prepare(targets.numpy())
        e, t = enc.cuda(), targets.cuda()
        for it in range(3):
            t1 = time.time()
            step.fwd_bwd(e, t, T, w, seed=6 + it)
            st = step.engine.chain_status()
            dt = time.time() - t1
            if st or dt > 0.5:
                events += 1
                print("round %d %s iter %d: status 0x%x, %.2f s" % (r, name, it, st, dt), flush=True)
                step.engine.chain_reset(False)
        del step, dec, rec
print("done: %d rounds, %d events, %.0f s" % (rounds, events, time.time() - t0))
