import sys, numpy as np, torch
sys.path.insert(0, '.')
import recnet_amd as R
from tests.gpu_util import load_case, make_models
for name in ["full_global_B8", "full_local_B8", "global_train"]:
    for prec in ["f32"]:
        g, dims, kind, decP, recP, enc, targets = load_case(name)
        C, dec, rec = make_models(dims, kind, prec, decP, recP)
        encd, tg = enc.cuda(), targets.cuda()
        dl, hid, _ = R.forward_decoder(dec, encd, tg, tg > 0, 1.0, seed=int(g["meta_drop_seed"]))
        fwd = R.forward_global_reconstructor if kind == "global" else R.forward_local_reconstructor
        rl = fwd(hid, encd, rec, seed=int(g["meta_drop_seed"]))
        eng = rec["_state"].engines[("rec", dims[0], dims[1])]
        sc = eng.scalar_dict()
        print(name, prec, {k: sc[k] for k in ("rec_mse", "rec_reg", "rec_loss")},
              {k: float(g[k]) for k in ("rec_mse", "rec_reg", "rec_loss")})
        # per tensor norms on GPU via torch
        for k, p in rec["model"].named_parameters():
            print("   ", k, float(p.detach().double().norm()), float(p.detach().norm()))
