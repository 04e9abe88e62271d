"""Checkpoint layout (train.py:398-420): CPU part — reading a file whose 'config' entry is a class of a code base that
is not importable here (the reference pickles `config.TrainConfig`)."""
import os
import sys
import types

import pytest
import torch

from recnet_amd import checkpoint as CK


def test_reads_a_reference_style_checkpoint_without_its_config_module(tmp_path):
    mod = types.ModuleType("config_of_another_codebase")

    class TrainConfig:
        batch_size = 100
    TrainConfig.__module__ = mod.__name__
    TrainConfig.__qualname__ = "TrainConfig"
    mod.TrainConfig = TrainConfig
    sys.modules[mod.__name__] = mod
    path = os.path.join(tmp_path, "10_checkpoint.tar")
    try:
        torch.save({"iteration": 10, "dec": {"attn_b": torch.ones(3)}, "dec_opt": {"state": {}, "param_groups": []},
                    "loss": torch.tensor(1.5), "config": TrainConfig}, path)
    finally:
        del sys.modules[mod.__name__]
    with pytest.raises(Exception):
        torch.load(path, weights_only=False)
    ck = CK.read_checkpoint(path)
    assert ck["iteration"] == 10 and torch.equal(ck["dec"]["attn_b"], torch.ones(3)) and float(ck["loss"]) == 1.5
    assert ck["config"].__name__ == "TrainConfig"


def test_config_to_dict_keeps_plain_attributes():
    import recnet_amd as R
    d = CK.config_to_dict(R.make_config(batch_size=7))
    assert d["batch_size"] == 7 and d["decoder_model"] == "LSTM" and "init_word2idx" in d
