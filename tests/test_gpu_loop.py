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


def test_search_after_training_uses_the_current_weights():
    """The HIP optimiser updates parameters through raw pointers (no torch `_version` bump): a search engine created
    before further training must not keep decoding with the weights of its first call (periodic test scoring,
    train.py:376-395)."""
    C = R.make_config(use_recon=True, reconstructor_type="global", decoder_learning_rate=3e-2, **DIMS)
    torch.manual_seed(3)
    tr = R.Trainer(C, V, use_graphs=False)
    dm = tr.decoder["model"]
    enc = torch.from_numpy(_batches(1, 9)[0][0]).cuda()
    B, H = 6, DIMS["decoder_hidden_size"]

    def search(model):
        model.eval()
        inp = torch.full((1, B), 1, dtype=torch.long, device="cuda")
        hid = (torch.zeros(1, B, H, device="cuda"), torch.zeros(1, B, H, device="cuda"))
        out = (R.greedy_search(C, model, inp, hid, enc), R.beam_search(C, 3, None, model, inp, hid, enc))
        lg, _ = model(inp, hid, enc)                      # the per-step API keeps a packed-weights cache of its own
        model.train()
        return out, lg.clone()

    def fresh_copy():
        m = R.Decoder(model_name=C.decoder_model, n_layers=1, encoder_size=C.encoder_output_size, embedding_size=C.embedding_size,
                      embedding_scale=C.embedding_scale, hidden_size=C.decoder_hidden_size, attn_size=C.decoder_attn_size,
                      output_size=V, embedding_dropout=C.embedding_dropout, dropout=C.decoder_dropout,
                      out_dropout=C.decoder_out_dropout, precision="f32").cuda()
        m.load_state_dict(dm.state_dict())
        return m

    (s0, lg0) = search(dm)
    tr.fit(iter(_batches(6, 1)), 6)
    (s1, lg1) = search(dm)
    (s1_fresh, lg1_fresh) = search(fresh_copy())
    assert s1 == s1_fresh and torch.equal(lg1, lg1_fresh)
    assert not torch.equal(lg0, lg1)                       # the six steps at lr 3e-2 did move the logits
    tr.fit(iter(_batches(4, 2)), 10)
    (s2, lg2) = search(dm)
    (s2_fresh, lg2_fresh) = search(fresh_copy())
    assert s2 == s2_fresh and torch.equal(lg2, lg2_fresh)


def test_trainer_resume_equals_uninterrupted_run(tmp_path):
    """Trainer built FIRST, checkpoint loaded afterwards (its engines already hold packed images of the random
    initial weights): the resumed run must equal the uninterrupted one."""
    C = R.make_config(use_recon=True, reconstructor_type="local", **DIMS_CHAIN)
    data = _batches(6)
    torch.manual_seed(0)
    a = R.Trainer(C, V)
    a.fit(iter(data), 6, save_every=3, save_dpath=str(tmp_path))
    torch.manual_seed(123)                                   # different initial weights
    b = R.Trainer(C, V)
    ck = b.resume(os.path.join(str(tmp_path), "3_checkpoint.tar"))
    assert ck["iteration"] == 3 and b.iteration == 3 and b.decoder["_state"].step == 3
    b.fit(iter(data[3:]), 6)
    assert b.iteration == 6
    for md_a, md_b in ((a.decoder, b.decoder), (a.reconstructor, b.reconstructor)):
        for (k, x), (_, y) in zip(md_a["model"].state_dict().items(), md_b["model"].state_dict().items()):
            assert torch.equal(x, y), k


def test_chain_give_up_skips_the_update_and_is_reported():
    """A persistent chain kernel that gives up a bounded wait (here: its sticky word is set by hand) must not corrupt
    the run: the optimiser kernels skip the update, the loss is NaN, Trainer.check_health raises, and after the reset
    the handle keeps training on the per-step kernels."""
    C = R.make_config(use_recon=True, reconstructor_type="global", **DIMS_CHAIN)
    torch.manual_seed(0)
    tr = R.Trainer(C, V, use_graphs=False)
    data = _batches(4)
    tr.fit(iter(data[:1]), 1, log_every=1, log=lambda m: None)
    eng = tr.dp.step_impl.engine
    assert eng.chain_status() == 0
    before = {k: v.clone() for k, v in tr.decoder["model"].state_dict().items()}
    # sticky word of the reconstructor's forward chain: workspace word 64 (ctrl block) + 257
    off = (eng._ws_ptr - eng.workspace.data_ptr()) + (64 + 257) * 4
    eng.workspace[off:off + 4] = torch.tensor([1, 0, 0, 0], dtype=torch.uint8, device="cuda")
    with pytest.raises(RuntimeError, match="gave up"):
        tr.fit(iter(data[1:2]), 2, log_every=1, log=lambda m: None)
    for k, v in tr.decoder["model"].state_dict().items():
        assert torch.equal(v, before[k]), k                   # the poisoned step did not touch the parameters
    assert eng.chain_status() == 0                          # check_health reset the words and disabled the chains
    hist = tr.fit(iter(data[2:]), 4, log_every=1, log=lambda m: None)
    assert len(hist) == 2 and all(np.isfinite(h["loss"]) for h in hist)
    assert any(not torch.equal(v, before[k]) for k, v in tr.decoder["model"].state_dict().items())


def test_teacher_forcing_ratio_below_one_runs_eager_steps_with_the_host_draw():
    """config.py:71 decoder_teacher_forcing_ratio < 1: the draw of train.py:38 is a host decision per iteration, so the Trainer
    keeps to eager steps (a replayed graph makes no draw; GraphedStep refuses such a step) — and the same seven iterations
    through a plain TrainStep with the same Python seed give the same parameters, free-running iterations included."""
    import random
    C = R.make_config(use_recon=True, reconstructor_type="global", decoder_teacher_forcing_ratio=0.5, **DIMS)
    data = _batches(7)
    torch.manual_seed(0)
    tr = R.Trainer(C, V)
    random.seed(11)
    pattern = [random.random() <= 0.5 for _ in range(7)]
    assert any(pattern) and not all(pattern)
    random.seed(11)
    tr.fit(iter(data), 7, log_every=100)
    assert tr.iteration == 7 and len(tr._graphs) == 0
    torch.manual_seed(0)
    dec, rec = R.build_decoder(V, C), R.build_reconstructor(C)
    step = R.TrainStep(dec, rec)
    assert step.teacher_forcing_ratio == 0.5
    with pytest.raises(ValueError):
        enc0, tg0 = torch.from_numpy(data[0][0]).cuda(), torch.from_numpy(data[0][1]).cuda()
        T0, w0 = step.prepare(data[0][1])
        R.GraphedStep(R.DataParallelTrainStep(dec, rec, 6, 0, 1), enc0, tg0, T0, w0)
    random.seed(11)
    for (enc, tg), tf in zip(data, pattern):
        T, w = step.prepare(tg)
        step(torch.from_numpy(enc).cuda(), torch.from_numpy(tg).cuda(), T, w)
        assert (step.output_indices is None) == tf
    for k, v in dec["model"].state_dict().items():
        assert torch.equal(v, tr.decoder["model"].state_dict()[k]), k
    for k, v in rec["model"].state_dict().items():
        assert torch.equal(v, tr.reconstructor["model"].state_dict()[k]), k
