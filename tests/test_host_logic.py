"""CPU-only tests of the host-side mirror of the reference interface: loop-length / loss-normaliser
arithmetic, module parameter sets (state_dict keys and shapes identical to the reference's), config
attribute names, shard bounds, synthetic batches."""
import numpy as np
import pytest
import torch

import recnet_amd as R
from oracle import recnet_oracle as O
from tests import golden_util as GU


def test_decode_len_matches_reference_loop_exit():
    for name in ["dec_eval", "dec_T31", "dec_T4_samelen", "local_T31", "global_eval"]:
        g = GU.load(name)
        masks = g["targets"] > 0
        assert R.decode_len(masks) == int(g["T"]) == O.decode_len(torch.from_numpy(masks))


def test_step_weights_reproduce_reference_loss_normalisation():
    g = GU.load("dec_eval")
    masks = g["targets"] > 0
    T = int(g["T"])
    w = R.step_weights(masks, T)
    n_t = masks[:T].sum(1)
    np.testing.assert_allclose(w, 1.0 / (n_t * n_t.sum()), rtol=1e-6)
    with pytest.raises(ValueError):
        R.step_weights(masks, T + 2)          # a step with no unmasked caption


def test_shard_bounds():
    assert [R.shard_bounds(100, 8, r) for r in range(8)] == [(0, 13), (13, 26), (26, 39), (39, 52), (52, 64), (64, 76),
                                                             (76, 88), (88, 100)]
    assert R.shard_bounds(7, 2, 0) == (0, 4) and R.shard_bounds(7, 2, 1) == (4, 7)
    for Bg, G in ((256, 8), (512, 8), (5, 3)):
        b = [R.shard_bounds(Bg, G, r) for r in range(G)]
        assert b[0][0] == 0 and b[-1][1] == Bg and all(b[i][1] == b[i + 1][0] for i in range(G - 1))


@pytest.mark.parametrize("name,kind", [("dec_eval", None), ("global_train", "global"), ("local_train", "local")])
def test_modules_have_the_reference_parameter_sets(name, kind):
    g = GU.load(name)
    B, F, D, V, E, H, A, RA = [int(x) for x in g["meta_dims"]]
    dec = R.Decoder("LSTM", 1, D, E, 1, H, A, V, 0.5, 0.5, 0.5)
    ref = GU.group(g, "dec_init")
    sd = dec.state_dict()
    assert list(sd.keys()) == list(ref.keys())                      # same keys, same order
    assert [tuple(v.shape) for v in sd.values()] == [tuple(v.shape) for v in ref.values()]
    dec.load_state_dict(ref)                                       # reference checkpoints load (eval.py:204)
    assert [n for n, _ in dec.named_parameters()] == O.decoder_param_order(ref)
    assert torch.equal(R.Decoder("LSTM", 1, D, E, 1, H, A, V, 0.5, 0.5, 0.5).attn_b, torch.ones(A))   # decoder.py:27
    if kind:
        cls = R.GlobalReconstructor if kind == "global" else R.LocalReconstructor
        last = 30 if kind == "global" else RA
        rec = cls("LSTM", 1, H, D, 0.5, 0.5, last)
        rref = GU.group(g, "rec_init")
        assert list(rec.state_dict().keys()) == list(rref.keys())
        rec.load_state_dict(rref)


def test_unsupported_variants_raise_like_the_reference_enum_checks():
    with pytest.raises(NotImplementedError):
        R.Decoder("RNN", 1, 8, 4, 1, 8, 4, 16, 0.5, 0.5, 0.5)
    assert R.Decoder("GRU", 1, 8, 4, 1, 8, 4, 16, 0.5, 0.5, 0.5).rnn.weight_ih_l0.shape == (3 * 8, 4 + 8)   # nn.GRU rows
    with pytest.raises(NotImplementedError):
        R.LocalReconstructor("LSTM", 2, 8, 8, 0.5, 0.5, 4)
    C = R.make_config(reconstructor_type="middle", device="cpu")
    with pytest.raises(NotImplementedError, match="Unknown reconstructor"):
        R.build_reconstructor(C)
    with pytest.raises(AttributeError):
        R.make_config(no_such_option=1)


def test_config_keeps_the_reference_attribute_names():
    C = R.TrainConfig
    for k, v in dict(caption_max_len=30, batch_size=100, embedding_size=468, embedding_dropout=0.5, embedding_scale=1,
                     encoder_output_size=1536, encoder_output_len=28, decoder_n_layers=1, decoder_hidden_size=512,
                     decoder_attn_size=128, decoder_dropout=0.5, decoder_out_dropout=0.5,
                     decoder_teacher_forcing_ratio=1.0, reconstructor_type="local", reconstructor_hidden_size=1536,
                     reconstructor_attn_size=128, decoder_learning_rate=1e-5, reconstructor_learning_rate=1e-6,
                     decoder_weight_decay=1e-5, reconstructor_weight_decay=1e-5, decoder_use_amsgrad=True,
                     reconstructor_use_amsgrad=False, use_gradient_clip=True, gradient_clip=50.0).items():
        assert getattr(C, k) == v, k
    assert C.init_word2idx == {"<PAD>": 0, "<SOS>": 1, "<EOS>": 2}


def test_synthetic_batch_shape_and_loop_length():
    from recnet_amd.synthetic import synthetic_features, synthetic_targets
    t = synthetic_targets(16, 4188)
    assert t.shape == (31, 16) and t.dtype == torch.int64
    assert R.decode_len((t > 0).numpy()) == 31                       # caption 0 has 30 words + <EOS>
    assert int((t == 2).sum()) == 16 and int(t.max()) < 4188
    assert synthetic_features(3, 28, 1536).shape == (3, 28, 1536)


def test_fused_adam_state_dict_has_torch_adam_layout():
    C = R.make_config(device="cpu", encoder_output_size=16, reconstructor_hidden_size=16, embedding_size=8,
                      decoder_hidden_size=8, decoder_attn_size=4)
    dec = R.build_decoder(20, C)
    opt = dec["optimizer"]
    opt._ensure_state()
    sd = opt.state_dict()
    assert set(sd["state"][0].keys()) == {"step", "exp_avg", "exp_avg_sq", "max_exp_avg_sq"}   # amsgrad (train.py:149)
    assert sd["param_groups"][0]["lr"] == 1e-5 and sd["param_groups"][0]["weight_decay"] == 1e-5
    assert len(sd["state"]) == 11
