"""Checkpoints with the reference's dict layout (train.py:398-420 writes, eval.py:173,204 reads):

    {'iteration', 'dec', 'rec', 'dec_opt', 'rec_opt', 'loss', 'config'}

'dec' / 'rec' are the modules' state_dicts (same keys and shapes as the reference's nn.Modules), 'dec_opt' / 'rec_opt'
are torch.optim.Adam-layout state_dicts, so a file written here loads in the reference and vice versa.  One difference:
the reference pickles its config CLASS (`'config': C`), which only unpickles where its `config` module is importable;
this module stores a plain dict of the config attributes and, when reading a reference checkpoint, replaces classes it
cannot import by placeholders instead of failing.
"""
import io
import pickle

import torch


def config_to_dict(C):
    return {k: getattr(C, k) for k in dir(C) if not k.startswith("_") and not callable(getattr(C, k))}


def save_checkpoint(path, iteration, decoder, reconstructor=None, loss=None, config=None):
    """decoder / reconstructor: the dicts of build_decoder / build_reconstructor."""
    ckpt = {"iteration": int(iteration), "dec": decoder["model"].state_dict(),
            "dec_opt": decoder["optimizer"].state_dict(),
            "loss": loss.detach().cpu() if isinstance(loss, torch.Tensor) else loss,
            "config": config_to_dict(config if config is not None else decoder["_C"])}
    if reconstructor is not None:
        ckpt["rec"] = reconstructor["model"].state_dict()
        ckpt["rec_opt"] = reconstructor["optimizer"].state_dict()
    torch.save(ckpt, path)
    return ckpt


class _Placeholder:
    """Stands in for a class of the writer's code base that is not importable here (e.g. config.TrainConfig)."""

    def __init__(self, *a, **k):
        pass

    def __setstate__(self, state):
        self.__dict__.update(state if isinstance(state, dict) else {"state": state})


class _TolerantUnpickler(pickle.Unpickler):
    def find_class(self, module, name):
        try:
            return super().find_class(module, name)
        except (ImportError, AttributeError):
            return type(name, (_Placeholder,), {"__module__": module})


class _tolerant_pickle:
    """pickle_module for torch.load: the stock module with the tolerant Unpickler."""
    Unpickler = _TolerantUnpickler
    load = staticmethod(lambda f, **kw: _TolerantUnpickler(f, **kw).load())
    loads = staticmethod(lambda b, **kw: _TolerantUnpickler(io.BytesIO(b), **kw).load())
    __name__ = "pickle"


def read_checkpoint(path, map_location="cpu"):
    """torch.load of a checkpoint written here or by the reference's train.py."""
    return torch.load(path, map_location=map_location, weights_only=False, pickle_module=_tolerant_pickle)


def load_checkpoint(path, decoder, reconstructor=None, load_optimizer=True, map_location="cpu"):
    """Restores model parameters (eval.py:204: `decoder.load_state_dict(checkpoint['dec'])`) and, for resuming a run,
    the optimiser state and the iteration.  `decoder` may be the build_decoder dict or a bare Decoder module.
    Returns the checkpoint dict."""
    ckpt = read_checkpoint(path, map_location)
    dm = decoder["model"] if isinstance(decoder, dict) else decoder
    dm.load_state_dict(ckpt["dec"])
    if reconstructor is not None:
        if "rec" not in ckpt:
            raise KeyError("checkpoint has no reconstructor ('rec')")
        rm = reconstructor["model"] if isinstance(reconstructor, dict) else reconstructor
        rm.load_state_dict(ckpt["rec"])
    # Engines that are already bound (a Trainer built before the load) hold packed operand images of the OLD weights:
    # refresh them, and tell the per-step / search engines of the modules that the weights moved.
    for md in (decoder, reconstructor):
        if md is None:
            continue
        model = md["model"] if isinstance(md, dict) else md
        model.mark_weights_changed()
        if isinstance(md, dict):
            for eng in {id(e): e for e in md["_state"].engines.values()}.values():
                eng.pack_weights()
    if load_optimizer and isinstance(decoder, dict):
        decoder["optimizer"].load_state_dict(ckpt["dec_opt"])
        if reconstructor is not None and isinstance(reconstructor, dict):
            reconstructor["optimizer"].load_state_dict(ckpt["rec_opt"])
            reconstructor["_state"].step = decoder["_state"].step
    return ckpt
