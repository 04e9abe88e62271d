"""Loss trajectories of the bf16-MFMA path and the exact-fp32 path on the same synthetic stream (same initial
parameters, same dropout seeds): sanity that the bf16 operand rounding does not change how the model trains.
Uses a larger learning rate than the reference's 1e-5 so that a few hundred steps move the loss visibly."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import recnet_amd as R
from recnet_amd.synthetic import synthetic_features, synthetic_targets

B, F, D, V = 100, 28, 1536, 4188
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
curves = {}
for prec in ("f32", "bf16"):
    C = R.make_config(batch_size=B, use_recon=True, reconstructor_type="global", precision=prec, decoder_learning_rate=3e-4,
                      reconstructor_learning_rate=3e-5)
    torch.manual_seed(0)
    dec, rec = R.build_decoder(V, C), R.build_reconstructor(C)
    step = R.TrainStep(dec, rec)
    data = [(synthetic_features(B, F, D, seed=50 + i).cuda(), synthetic_targets(B, V, seed=50 + i)) for i in range(4)]
    out = []
    for it in range(steps):
        enc, tg = data[it % 4]
        T, w = step.prepare(tg.numpy())
        sc = step(enc, tg.cuda(), T, w)
        if it % 25 == 0 or it == steps - 1:
            v = sc.cpu().tolist()
            out.append((it, round(v[0], 4), round(v[3], 5)))      # decoder CE, reconstruction MSE
    curves[prec] = out
for a, b in zip(curves["f32"], curves["bf16"]):
    print("step %4d   CE f32 %.4f bf16 %.4f   MSE f32 %.5f bf16 %.5f" % (a[0], a[1], b[1], a[2], b[2]))
