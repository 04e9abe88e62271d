"""`attn_normalize="softmax"` (opt-in; BASELINE north_star's "wavefront softmax reduction"): the decoder's attention
energies are normalised over the frames before the weighted mean.  The reference constructs `nn.Softmax(dim=1)`
(models/decoder.py:30) and never calls it, so there is no reference behaviour to match and "none" stays the default; the
option is held against the oracle with the same switch (oracle.recnet_oracle.ATTN_NORMALIZE) — losses and every gradient,
on the per-step kernels (fp32 path), the persistent chain kernels (bf16 path, incl. frames served from LDS) and the
per-step module API."""
import numpy as np
import pytest
import torch

import recnet_amd as R
from oracle import recnet_oracle as O
from tests import golden_util as GU
from tests.gpu_util import TOL, make_models, oracle_grads, rel_err

pytestmark = pytest.mark.gpu

CASES = {
    "small_per_step": ([6, 7, 40, 37, 10, 24, 16, 16], [5, 2, 7, 3, 1, 4]),              # H % 16 != 0: per-step kernels
    "chain_H32": ([37, 5, 64, 29, 8, 32, 16, 16], [(5 * i) % 8 for i in range(37)]),        # persistent decoder chains
    "chain_F40_A128": ([6, 40, 64, 53, 12, 64, 128, 16], [9, 2, 5, 12, 1, 7]),             # + frames 32..39 from LDS
    "F70_two_softmax_rounds": ([4, 70, 32, 29, 8, 24, 16, 16], [3, 5, 1, 2]),               # F > 64: the wave loops twice
}


@pytest.fixture
def softmax_oracle():
    O.ATTN_NORMALIZE = "softmax"
    yield
    O.ATTN_NORMALIZE = "none"


@pytest.mark.parametrize("prec", ["f32", "bf16"])
@pytest.mark.parametrize("kind", [None, "global", "local"])
@pytest.mark.parametrize("case", sorted(CASES))
def test_softmax_attention_matches_oracle(case, kind, prec, softmax_oracle):
    dims, lens = CASES[case]
    B, F, D, V, E, H, A, RA = dims
    decP = GU.formula_params(GU.decoder_shapes(V, E, H, A, D), 41)
    recP = GU.formula_params(GU.rec_shapes(kind, H, D, RA), 42) if kind else None
    enc, targets = GU.make_batch(B, F, D, V, lens, 79)
    C, dec, rec = make_models(dims, kind, prec, decP, recP, attn_normalize="softmax")
    step = R.TrainStep(dec, rec)
    T, w = step.prepare(targets.numpy())
    step.engine.poison_lds()
    step.fwd_bwd(enc.cuda(), targets.cuda(), T, w, seed=6)
    step.engine.add_reg_grad(0, 1.0)
    if rec:
        step.engine.add_reg_grad(1, 1.0)
    torch.cuda.synchronize()
    ref = oracle_grads(decP, recP, kind, enc, targets, True, 6)
    sc = step.engine.scalar_dict()
    tol = TOL[prec]
    assert abs(sc["dec_ce"] - ref["dec_ce"]) <= tol["loss"] * abs(ref["dec_ce"])
    for grp, md in (("dec", dec), ("rec", rec)):
        if md is None:
            continue
        gv = md["_state"].flat()["grad"].views
        for k in gv:
            assert rel_err(gv[k].cpu().numpy(), ref[grp + "_grad"][k]) <= tol["grad"] * 1.5, (case, kind, prec, grp, k)
    # and the option does change the result: the un-normalised default gives a different loss
    O.ATTN_NORMALIZE = "none"
    assert abs(oracle_grads(decP, recP, kind, enc, targets, True, 6)["dec_ce"] - ref["dec_ce"]) > 1e-4 * abs(ref["dec_ce"])


def test_step_api_and_unknown_value(softmax_oracle):
    dims, lens = CASES["small_per_step"]
    B, F, D, V, E, H, A, RA = dims
    decP = GU.formula_params(GU.decoder_shapes(V, E, H, A, D), 41)
    enc, targets = GU.make_batch(B, F, D, V, lens, 79)
    m = R.Decoder(model_name="LSTM", n_layers=1, encoder_size=D, embedding_size=E, embedding_scale=1, hidden_size=H, attn_size=A,
                  output_size=V, embedding_dropout=0.5, dropout=0.5, out_dropout=0.5, precision="f32", attn_normalize="softmax").cuda()
    m.load_state_dict(decP)
    m.eval()
    tok = torch.full((1, B), 1, dtype=torch.long, device="cuda")
    hid = (torch.zeros(1, B, H, device="cuda"), torch.zeros(1, B, H, device="cuda"))
    ohid = O.zero_hidden(B, H, "LSTM")
    for t in range(3):
        lg, hid = m(tok, hid, enc.cuda())
        olg, ohid = O.decoder_step(decP, tok.cpu(), ohid, enc, t=t)
        assert np.abs(lg.cpu().numpy() - olg.numpy()).max() <= 2e-5
        tok = targets[t].view(1, -1).cuda()
    with pytest.raises(NotImplementedError):
        R.Decoder(model_name="LSTM", n_layers=1, encoder_size=D, embedding_size=E, embedding_scale=1, hidden_size=H, attn_size=A,
                  output_size=V, embedding_dropout=0.5, dropout=0.5, out_dropout=0.5, attn_normalize="sparsemax")
