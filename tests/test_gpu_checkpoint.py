"""Checkpoint round trip on the device: the reference's dict layout, resume == uninterrupted run, and the optimiser
state interchanges with torch.optim.Adam (what the reference's checkpoints contain)."""
import os

import numpy as np
import pytest
import torch

import recnet_amd as R
from tests import golden_util as GU
from tests.gpu_util import TOL, load_case, make_models

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", ["global_train", "gru_local_train3"])
def test_resume_equals_uninterrupted_run(name, tmp_path):
    g, dims, kind, decP, recP, enc, targets = load_case(name)
    encd, tg = enc.cuda(), targets.cuda()
    seed0 = int(g["meta_drop_seed"])
    C, dec, rec = make_models(dims, kind, "f32", decP, recP, cells=g["_cells"])
    step = R.TrainStep(dec, rec)
    T, w = step.prepare(targets.numpy())
    sc = step(encd, tg, T, w, seed=seed0)
    path = os.path.join(tmp_path, "1_checkpoint.tar")
    ck = R.save_checkpoint(path, 1, dec, rec, loss=sc[6], config=C)
    assert sorted(ck.keys()) == ["config", "dec", "dec_opt", "iteration", "loss", "rec", "rec_opt"]      # train.py:402-410
    assert list(ck["dec"].keys()) == list(decP.keys())
    # a fresh pair of models, parameters and optimiser state from the file, two more steps
    C2, dec2, rec2 = make_models(dims, kind, "f32", {k: torch.zeros_like(v) for k, v in decP.items()},
                                 {k: torch.zeros_like(v) for k, v in recP.items()}, cells=g["_cells"])
    ck2 = R.load_checkpoint(path, dec2, rec2)
    assert ck2["iteration"] == 1 and dec2["_state"].step == 1 and rec2["_state"].step == 1
    step2 = R.TrainStep(dec2, rec2)
    for it in (1, 2):
        step2(encd, tg, T, w, seed=seed0 + it)
    for grp, md in (("dec", dec2), ("rec", rec2)):
        for k, v in GU.group(g, "%s_after3" % grp).items():
            assert np.abs(md["model"].state_dict()[k].cpu().numpy() - v.numpy()).max() <= TOL["f32"]["param"], (grp, k)


def test_optimizer_state_interchanges_with_torch_adam():
    g, dims, kind, decP, recP, enc, targets = load_case("dec_train")
    C, dec, _ = make_models(dims, None, "f32", decP, None)
    step = R.TrainStep(dec, None)
    T, w = step.prepare(targets.numpy())
    step(enc.cuda(), targets.cuda(), T, w, seed=int(g["meta_drop_seed"]))
    sd = dec["optimizer"].state_dict()
    # ours -> torch.optim.Adam (train.py:149 construction)
    ps = [torch.nn.Parameter(p.detach().clone()) for p in dec["model"].parameters()]
    ref = torch.optim.Adam(ps, lr=1e-5, weight_decay=1e-5, amsgrad=True)
    ref.load_state_dict(sd)
    st = ref.state[ps[1]]
    assert float(st["step"]) == 1.0 and torch.equal(st["exp_avg"], sd["state"][1]["exp_avg"])
    assert set(sd["state"][0].keys()) == {"step", "exp_avg", "exp_avg_sq", "max_exp_avg_sq"}
    # torch.optim.Adam -> ours: the moments land in the buffers the HIP optimiser updates
    sd2 = ref.state_dict()
    for v in sd2["state"].values():
        v["exp_avg"] = v["exp_avg"] * 0 + 0.25
        v["step"] = torch.tensor(5.0)
    dec["optimizer"].load_state_dict(sd2)
    fl = dec["_state"].flat()
    assert dec["_state"].step == 5 and all(bool((v == 0.25).all()) for v in fl["exp_avg"].views.values())
