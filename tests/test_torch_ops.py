"""The custom-op layer (csrc/torch_ops.cpp) on a box without a GPU: the extension loads, every recnet:: schema that
api.py / modules.py / search.py call is registered with the documented signature, and a CPU tensor is refused (there is
no CPU kernel: the HIP path is the only path)."""
import pytest
import torch

from recnet_amd import _ops


def test_every_op_is_registered_with_tensor_schemas():
    ops = _ops.load()
    for n in _ops.OP_NAMES:
        sch = getattr(ops, n).default._schema
        assert str(sch).startswith("recnet::" + n + "(int handle"), sch
    s = str(ops.forward_decoder.default._schema)
    assert "Tensor encoder_outputs, Tensor targets, int T, Tensor step_weight, bool train, int seed" in s
    assert "-> (Tensor loss, Tensor hiddens, Tensor scalars)" in s
    assert "Tensor? dhiddens" in str(ops.backward_decoder.default._schema)
    assert "-> Tensor dhiddens" in str(ops.backward_reconstructor.default._schema)


def test_cpu_tensors_are_refused():
    ops = _ops.load()
    enc = torch.zeros(2, 3, 8)
    with pytest.raises((RuntimeError, NotImplementedError)):
        ops.greedy_search(1, enc)
    with pytest.raises((RuntimeError, NotImplementedError)):
        ops.train_step(1, enc, torch.zeros(31, 2, dtype=torch.long), 3, torch.ones(3), 0, 1)
