"""The train-loop shell on the device: feeder -> eager/graph steps == plain TrainStep steps, validation pass, scoring,
checkpoint files."""
import os

import numpy as np
import pytest
import torch

import recnet_amd as R
from recnet_amd import feed
from recnet_amd.checkpoint import read_checkpoint

pytestmark = pytest.mark.gpu

DIMS = dict(batch_size=6, encoder_output_len=5, encoder_output_size=32, embedding_size=12, decoder_hidden_size=24,
            decoder_attn_size=8, reconstructor_hidden_size=32, reconstructor_attn_size=8, precision="f32")
V = 41


def _batches(n, seed=0):
    rng = np.random.RandomState(seed)
    out = []
    for _ in range(n):
        lens = rng.randint(1, 9, size=6)
        caps = [feed.pad_caption(rng.randint(3, V, size=L), 30) for L in lens]
        vids = [rng.randn(5, 32).astype(np.float32) for _ in lens]
        out.append(feed.collate_batch(vids, caps, 6))
    return out


# bf16 with H % 16 == 0: the persistent chain kernels (csrc/dec_chain.hpp, csrc/rec_chain.hpp) inside replayed hipGraphs,
# held bit for bit against the same kernels launched eagerly (every eager dispatch starts with a full acquire)
DIMS_CHAIN = dict(DIMS, decoder_hidden_size=32, decoder_attn_size=16, precision="bf16")


@pytest.mark.parametrize("dims", [DIMS, DIMS_CHAIN], ids=["f32", "bf16_chain_kernels"])
@pytest.mark.parametrize("kind", ["global", "local"])
def test_fit_equals_plain_steps_and_writes_checkpoints(kind, dims, tmp_path):
    C = R.make_config(use_recon=True, reconstructor_type=kind, **dims)
    data = _batches(7)
    torch.manual_seed(0)
    tr = R.Trainer(C, V)                                  # 2 eager steps, then one hipGraph per T
    logs = []
    hist = tr.fit(iter(data), 7, log_every=2, val_batches=lambda: _batches(2, 5), validate_every=4, save_every=7,
                  save_dpath=str(tmp_path), log=logs.append)
    assert tr.iteration == 7 and len(tr._graphs) >= 2            # several loop lengths T -> several graphs
    # the same 7 steps with the plain (un-graphed, un-fed) TrainStep from the same initial parameters
    torch.manual_seed(0)
    dec, rec = R.build_decoder(V, C), R.build_reconstructor(C)
    step = R.TrainStep(dec, rec)
    for enc, tg in data:
        T, w = step.prepare(tg)
        step(torch.from_numpy(enc).cuda(), torch.from_numpy(tg).cuda(), T, w)
    for (k, a), (_, b) in zip(tr.decoder["model"].state_dict().items(), dec["model"].state_dict().items()):
        assert torch.equal(a, b), k
    for (k, a), (_, b) in zip(tr.reconstructor["model"].state_dict().items(), rec["model"].state_dict().items()):
        assert torch.equal(a, b), k
    assert any("[Validation]" in m for m in logs) and sum(m.startswith("Iter") for m in logs) == 3
    assert [h["iteration"] for h in hist if "loss" in h] == [2, 4, 6]
    ck = read_checkpoint(os.path.join(str(tmp_path), "7_checkpoint.tar"))
    assert ck["iteration"] == 7 and float(ck["dec_opt"]["state"][0]["step"]) == 7.0 and "rec_opt" in ck


def test_validate_and_evaluate_shapes():
    C = R.make_config(use_recon=True, reconstructor_type="local", **DIMS)
    torch.manual_seed(1)
    tr = R.Trainer(C, V)
    idx2word = {i: "w%d" % i for i in range(V)}
    v = tr.validate(_batches(2, 3), idx2word)
    assert v["loss"] > 0 and abs(v["loss"] - (v["dec"] + v["rec"])) < 1e-4 and len(v["captions"]) == 12
    enc = _batches(1, 9)[0][0]
    vids = ["a", "b", "c", "d", "PAD", "PAD"]
    refs = {k: ["w3 w4 w5", "w7 w8"] for k in "abcd"}
    for method in ("greedy", ("beam", 3)):
        s = R.evaluate(C, [(vids, enc)], tr.decoder["model"], method, idx2word, refs)
        assert set(s) == {"Bleu_1", "Bleu_2", "Bleu_3", "Bleu_4", "CIDEr", "ROUGE_L"}
    with pytest.raises(NotImplementedError):
        R.evaluate(C, [(vids, enc)], tr.decoder["model"], "sampling", idx2word, refs)
