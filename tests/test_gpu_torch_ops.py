"""torch.ops.recnet.* on the device: the ops called directly (not through api.py) reproduce the golden losses /
gradients of the reference, the fused train_step op equals fwd_bwd + optimizer_step, and malformed tensors raise."""
import numpy as np
import pytest
import torch

import recnet_amd as R
from recnet_amd import _lib, _ops
from tests import golden_util as GU
from tests.gpu_util import TOL, load_case, make_models, rel_err

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", ["global_train", "local_train"])
def test_paired_forward_backward_ops_match_reference_goldens(name):
    g, dims, kind, decP, recP, enc, targets = load_case(name)
    ops = _ops.load()
    C, dec, rec = make_models(dims, kind, "f32", decP, recP)
    step = R.TrainStep(dec, rec)                       # engine with both models and their gradient buffers bound
    h = int(step.engine.handle.value)
    T, w = step.prepare(targets.numpy())
    seed = int(g["meta_drop_seed"])
    encd, tg = enc.cuda(), targets.cuda()
    loss_d, hid, sc_d = ops.forward_decoder(h, encd, tg, T, w, True, seed)
    loss_r, sc_r = ops.forward_reconstructor(h, encd, hid, T, True, seed)
    assert tuple(hid.shape) == (T, 1, dims[0], dims[5])
    assert abs(float(sc_d[0]) - float(g["dec_ce"])) <= TOL["f32"]["loss"] * abs(float(g["dec_ce"]))
    assert abs(float(sc_r[3]) - float(g["rec_mse"])) <= TOL["f32"]["loss"] * abs(float(g["rec_mse"]))
    assert np.abs(hid.cpu().numpy() - g["hiddens"]).max() <= TOL["f32"]["hid"]
    dh = ops.backward_reconstructor(h, encd, T, 1.0)
    ops.add_reg_grad(h, 1, 1.0)
    ops.backward_decoder(h, encd, tg, dh, 1.0)
    ops.add_reg_grad(h, 0, 1.0)
    torch.cuda.synchronize()
    for grp, md in (("dec", dec), ("rec", rec)):
        for k, v in GU.group(g, grp + "_grad").items():
            got = md["_state"].flat()["grad"].views[k].cpu().numpy()
            assert rel_err(got, v.numpy()) <= 3e-4, (grp, k)     # golden regulariser norm is float32


def test_train_step_op_equals_fwd_bwd_plus_optimizer_ops():
    g, dims, kind, decP, recP, enc, targets = load_case("global_train")
    ops = _ops.load()
    outs = []
    for fused in (True, False):
        C, dec, rec = make_models(dims, kind, "f32", decP, recP)
        step = R.TrainStep(dec, rec)
        h = int(step.engine.handle.value)
        T, w = step.prepare(targets.numpy())
        if fused:
            sc = ops.train_step(h, enc.cuda(), targets.cuda(), T, w, 42, 1)
        else:
            sc = ops.train_step_fwd_bwd(h, enc.cuda(), targets.cuda(), T, w, 42)
            ops.optimizer_step(h, 1, _lib.OPT_REG | _lib.OPT_CLIP)
        torch.cuda.synchronize()
        outs.append((sc.clone(), {k: v.clone() for k, v in dec["model"].state_dict().items()}))
    assert torch.allclose(outs[0][0][:7], outs[1][0][:7], rtol=1e-6, atol=0)
    for k in outs[0][1]:
        assert torch.equal(outs[0][1][k], outs[1][1][k]), k


def test_ops_validate_their_tensors():
    g, dims, kind, decP, recP, enc, targets = load_case("dec_train")
    ops = _ops.load()
    C, dec, _ = make_models(dims, None, "f32", decP, None)
    step = R.TrainStep(dec, None)
    h = int(step.engine.handle.value)
    T, w = step.prepare(targets.numpy())
    with pytest.raises(RuntimeError, match="dtype"):
        ops.forward_decoder(h, enc.cuda().double(), targets.cuda(), T, w, True, 1)
    with pytest.raises(RuntimeError, match="contiguous"):
        ops.forward_decoder(h, enc.cuda().transpose(0, 1), targets.cuda(), T, w, True, 1)
    with pytest.raises(RuntimeError, match="T entries"):
        ops.forward_decoder(h, enc.cuda(), targets.cuda(), T, w[:-1].contiguous(), True, 1)
    with pytest.raises(RuntimeError, match="null engine handle"):
        ops.forward_decoder(0, enc.cuda(), targets.cuda(), T, w, True, 1)
    with pytest.raises((RuntimeError, NotImplementedError)):
        ops.forward_decoder(h, enc, targets, T, w.cpu(), True, 1)          # CPU tensors: no kernel registered
