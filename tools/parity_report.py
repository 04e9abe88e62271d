"""Worst-case parity numbers of the HIP path against the golden vectors, per precision (run on a GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import recnet_amd as R
from tests.gpu_util import load_case, make_models, rel_err
cases = ["dec_eval", "dec_train", "dec_T31", "global_train", "global_eval", "local_train", "local_eval", "local_T31",
         "full_dec_B8", "full_global_B8", "full_local_B8"]
for prec in ("f32", "bf16"):
    worst = dict(hid=0, loss=0, grad=0, cos=1.0); who = {}
    for name in cases:
        g, dims, kind, decP, recP, enc, targets = load_case(name)
        C, dec, rec = make_models(dims, kind, prec, decP, recP)
        train = bool(int(g["meta_train_mode"])); seed = int(g["meta_drop_seed"])
        dec["model"].train(train)
        encd, tg = enc.cuda(), targets.cuda()
        dl, hid, _ = R.forward_decoder(dec, encd, tg, tg > 0, 1.0, seed=seed)
        loss = dl
        if kind:
            rec["model"].train(train)
            fwd = R.forward_global_reconstructor if kind == "global" else R.forward_local_reconstructor
            rl = fwd(hid, encd, rec, seed=seed); loss = dl + rl
        loss.backward(); torch.cuda.synchronize()
        e = np.abs(hid.detach().cpu().numpy() - g["hiddens"]).max()
        if e > worst["hid"]: worst["hid"] = e; who["hid"] = name
        sc = dec["_state"].engines[("dec", dims[0], dims[1])].scalar_dict()
        e = abs(sc["dec_ce"] - float(g["dec_ce"])) / abs(float(g["dec_ce"]))
        if e > worst["loss"]: worst["loss"] = e; who["loss"] = name
        for grp, md in (("dec", dec), ("rec", rec)):
            if md is None: continue
            for k, p in md["model"].named_parameters():
                key = "%s_grad/%s" % (grp, k)
                if key not in g: continue
                a = p.grad.detach().cpu().numpy().astype(np.float64).ravel(); b = g[key].astype(np.float64).ravel()
                e = rel_err(a, b); c = float(a @ b / (np.linalg.norm(a) * np.linalg.norm(b) + 1e-300))
                if e > worst["grad"]: worst["grad"] = e; who["grad"] = name + ":" + grp + "." + k
                if c < worst["cos"]: worst["cos"] = c; who["cos"] = name + ":" + grp + "." + k
    print(prec, {k: float("%.3g" % v) for k, v in worst.items()}, who)
