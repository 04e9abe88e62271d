"""Sequence-level entry points with the reference's names and signatures (train.py:17-197) and the
fused train-step body (train.py:248-273).

  forward_decoder(decoder, encoder_outputs, targets, target_masks, teacher_forcing_ratio)
      -> (loss, hiddens[T,1,B,H], output_indices)                       train.py:17-75
  forward_global_reconstructor(decoder_hiddens, encoder_outputs, reconstructor) -> loss   train.py:78-105
  forward_local_reconstructor(decoder_hiddens, encoder_outputs, reconstructor)  -> loss   train.py:108-131
  build_decoder(n_vocabs) / build_reconstructor()  -> {'model','loss','optimizer','lambda_reg'}
                                                                        train.py:134-197
The returned losses are autograd-connected: `(dec_loss + lambda * rec_loss).backward()` runs the HIP
backward kernels and leaves the gradients in `param.grad` (views of one flat buffer per model).
`TrainStep` is the same computation as one stream-ordered launch sequence with the regulariser
gradient, clipping and both Adam updates fused in — the path bench.py times.
"""
import random

import numpy as np
import torch

from . import _lib, _ops
from .config import TrainConfig
from .engine import Engine, FlatState
from .modules import Decoder, GlobalReconstructor, LocalReconstructor

PAD, SOS, EOS = 0, 1, 2


# ----------------------------------------------------------------------------- host-side helpers
def decode_len(target_masks, caption_max_len=30):
    """How many decoder steps the reference's loop runs (exit test at train.py:66).
    target_masks: [caption_max_len+1, B] array-like on the HOST."""
    m = np.asarray(target_masks).astype(bool)
    for t in range(caption_max_len + 1):
        if t == caption_max_len or not m[t + 1].any():
            return t + 1
    return caption_max_len + 1


def step_weights(target_masks, T):
    """w[t] = 1 / (n_t * sum_t n_t): the reference's loss is (sum_t mean_{b in mask_t} CE) / (sum_t n_t)
    (train.py:56-68).  target_masks must cover the GLOBAL batch under data parallelism."""
    m = np.asarray(target_masks).astype(bool)
    n_t = m[:T].sum(axis=1).astype(np.float64)
    N = n_t.sum()
    if N <= 0 or (n_t <= 0).any():
        raise ValueError("every executed decoder step needs at least one unmasked caption (n_t >= 1)")
    return (1.0 / (n_t * N)).astype(np.float32)


def _host_masks(targets, target_masks):
    src = target_masks if target_masks is not None else (targets > PAD)
    return src.detach().cpu().numpy() if isinstance(src, torch.Tensor) else np.asarray(src)


# ----------------------------------------------------------------------------- per-model state
class ModelState:
    """fp32 master parameters of one model + flat gradient / Adam-state buffers bound to engines."""

    def __init__(self, model, amsgrad):
        self.model = model
        self.amsgrad = bool(amsgrad)
        self.step = 0
        self._flat = None
        self.engines = {}

    def params(self):
        return {k: v for k, v in self.model.named_parameters()}

    def flat(self):
        if self._flat is None:
            P = self.params()
            dev = next(iter(P.values())).device
            shapes = {k: tuple(v.shape) for k, v in P.items()}
            self._flat = {"grad": FlatState(shapes, dev), "exp_avg": FlatState(shapes, dev),
                          "exp_avg_sq": FlatState(shapes, dev)}
            if self.amsgrad:
                self._flat["max_exp_avg_sq"] = FlatState(shapes, dev)
        return self._flat

    def bind(self, eng, which):
        fl = self.flat()
        P = {k: v.data for k, v in self.params().items()}
        args = (P, fl["grad"].views, fl["exp_avg"].views, fl["exp_avg_sq"].views,
                fl["max_exp_avg_sq"].views if self.amsgrad else None)
        (eng.bind_decoder if which == 0 else eng.bind_reconstructor)(*args)

    def publish_grads(self):
        g = self.flat()["grad"].views
        for k, p in self.params().items():
            p.grad = g[k]


def _hyper_from(C, model=None):
    hy = dict(embedding_scale=float(C.embedding_scale), embedding_dropout=C.embedding_dropout,
              decoder_out_dropout=C.decoder_out_dropout,
              reconstructor_decoder_dropout=getattr(C, "reconstructor_decoder_dropout", 0.5),
              decoder_learning_rate=C.decoder_learning_rate,
              reconstructor_learning_rate=C.reconstructor_learning_rate,
              decoder_weight_decay=C.decoder_weight_decay, reconstructor_weight_decay=C.reconstructor_weight_decay,
              decoder_use_amsgrad=C.decoder_use_amsgrad, reconstructor_use_amsgrad=C.reconstructor_use_amsgrad,
              gradient_clip=C.gradient_clip if C.use_gradient_clip else 0.0,
              decoder_lambda_reg=getattr(C, "decoder_lambda_reg", 1e-3),
              reconstructor_lambda_reg=getattr(C, "reconstructor_lambda_reg", 1e-2),
              lambda_recon=getattr(C, "lambda_recon", 1.0), caption_max_len=C.caption_max_len)
    return hy


def _dims(dec_model, B, F, rec_model=None):
    d = dec_model.dims(B, F)
    if rec_model is not None:
        d["R"] = rec_model.hidden_size
        d["rec_cell"] = rec_model.model_name
        if rec_model.kind == "local":
            d["RA"] = rec_model.attn_size
    return d


# ----------------------------------------------------------------------------- optimiser
class FusedAdam(torch.optim.Optimizer):
    """torch.optim.Adam semantics (coupled weight decay, optional AMSGrad — train.py:149,186) executed by
    the multi-tensor HIP kernel.  `state_dict()` has torch.optim.Adam's layout (step / exp_avg /
    exp_avg_sq / max_exp_avg_sq per parameter) so checkpoints interchange (train.py:398-420)."""

    def __init__(self, mstate, which, lr, weight_decay=0.0, amsgrad=False, betas=(0.9, 0.999), eps=1e-8,
                 hyper=None):
        self._ms, self._which, self._hyper = mstate, which, dict(hyper or {})
        params = [p for _, p in mstate.model.named_parameters()]
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, amsgrad=amsgrad))

    def _ensure_state(self):
        fl = self._ms.flat()
        for k, p in self._ms.model.named_parameters():
            st = self.state[p]
            if "exp_avg" not in st:
                st["step"] = torch.tensor(float(self._ms.step))
                st["exp_avg"] = fl["exp_avg"].views[k]
                st["exp_avg_sq"] = fl["exp_avg_sq"].views[k]
                if self._ms.amsgrad:
                    st["max_exp_avg_sq"] = fl["max_exp_avg_sq"].views[k]

    def _engine(self):
        ms = self._ms
        if ms.engines:
            return next(iter(ms.engines.values()))
        # no forward has run yet: a minimal engine is enough for the optimiser tables
        m = ms.model
        if self._which == 0:
            eng = Engine(m.dims(1, 1), None, m.precision, self._hyper, device=next(m.parameters()).device)
        else:
            d = dict(B=1, F=1, D=m.hidden_size, E=4, H=m.decoder_hidden_size, A=4, V=8, R=m.hidden_size,
                     RA=getattr(m, "attn_size", 0), rec_cell=m.model_name)
            eng = Engine(d, m.kind, m.precision, self._hyper, device=next(m.parameters()).device)
        ms.bind(eng, self._which)
        ms.engines[("opt",)] = eng
        return eng

    @torch.no_grad()
    def step(self, closure=None):
        if closure is not None:
            raise NotImplementedError("closure is not supported")
        self._ensure_state()
        g = self.param_groups[0]
        eng = self._engine()
        # hyper-parameters may have been edited through param_groups (lr schedules)
        c = eng.cfg
        if self._which == 0:
            c.decoder_learning_rate, c.decoder_weight_decay = float(g["lr"]), float(g["weight_decay"])
        else:
            c.reconstructor_learning_rate, c.reconstructor_weight_decay = float(g["lr"]), float(g["weight_decay"])
        self._ms.step += 1
        skip = _lib.OPT_SKIP_RECONSTRUCTOR if self._which == 0 else _lib.OPT_SKIP_DECODER
        eng.optimizer_step(self._ms.step, skip)
        self._ms.model.mark_weights_changed()
        for st in self.state.values():
            st["step"] = torch.tensor(float(self._ms.step))

    def state_dict(self):
        """torch.optim.Adam's layout ({'state': {i: {step, exp_avg, exp_avg_sq[, max_exp_avg_sq]}}, 'param_groups'}),
        which is what the reference stores under 'dec_opt' / 'rec_opt' (train.py:404-406)."""
        self._ensure_state()
        for st in self.state.values():
            st["step"] = torch.tensor(float(self._ms.step))
        return super().state_dict()

    @torch.no_grad()
    def load_state_dict(self, state_dict):
        """Accepts a torch.optim.Adam state_dict (the reference's checkpoints) or one of ours: the moments are copied
        INTO the flat device buffers the HIP optimiser is bound to (torch's own load would re-point the state at fresh
        tensors), the step count into the model state, the hyper-parameters into the param group."""
        self._ensure_state()
        params = self.param_groups[0]["params"]
        saved = state_dict["state"]
        ids = state_dict["param_groups"][0]["params"]
        if len(ids) != len(params):
            raise ValueError("optimizer state_dict has %d parameters, the model has %d" % (len(ids), len(params)))
        step = None
        for pid, p in zip(ids, params):
            st = saved.get(pid, saved.get(str(pid)))
            if st is None:                                # a parameter that never received a step
                continue
            mine = self.state[p]
            for k in ("exp_avg", "exp_avg_sq") + (("max_exp_avg_sq",) if self._ms.amsgrad else ()):
                if k not in st:
                    raise KeyError("optimizer state lacks %r (amsgrad mismatch?)" % k)
                mine[k].copy_(st[k].to(mine[k].device, torch.float32).reshape(mine[k].shape))
            step = int(float(st["step"])) if step is None else step
        self._ms.step = step or 0
        g, sg = self.param_groups[0], state_dict["param_groups"][0]
        for k in ("lr", "betas", "eps", "weight_decay", "amsgrad"):
            if k in sg:
                g[k] = sg[k]
        for st in self.state.values():
            st["step"] = torch.tensor(float(self._ms.step))

    def zero_grad(self, set_to_none=True):
        # every backward overwrites the flat gradient buffer (the reference zero_grad()s before each
        # backward, train.py:265-267), so dropping the references is all that is needed
        for p in self.param_groups[0]["params"]:
            p.grad = None


def clip_grad_norm_(model_dict, max_norm):
    """torch.nn.utils.clip_grad_norm_(model.parameters(), max_norm) (train.py:270) on the HIP path.
    Returns the total norm as a device tensor."""
    ms = model_dict["_state"]
    eng = next(iter(ms.engines.values()))
    return eng.clip_grad_norm(0 if isinstance(ms.model, Decoder) else 1, max_norm)


# ----------------------------------------------------------------------------- builders
def build_decoder(n_vocabs, C=TrainConfig):
    """train.py:134-160."""
    model = Decoder(model_name=C.decoder_model, n_layers=C.decoder_n_layers, encoder_size=C.encoder_output_size,
                    embedding_size=C.embedding_size, embedding_scale=C.embedding_scale,
                    hidden_size=C.decoder_hidden_size, attn_size=C.decoder_attn_size, output_size=n_vocabs,
                    embedding_dropout=C.embedding_dropout, dropout=C.decoder_dropout,
                    out_dropout=C.decoder_out_dropout, precision=getattr(C, "precision", "bf16"),
                    attn_normalize=getattr(C, "decoder_attn_normalize", "none"))
    model = model.to(C.device)
    ms = ModelState(model, C.decoder_use_amsgrad)
    hy = _hyper_from(C)
    opt = FusedAdam(ms, 0, lr=C.decoder_learning_rate, weight_decay=C.decoder_weight_decay,
                    amsgrad=C.decoder_use_amsgrad, hyper=hy)
    return {"model": model, "loss": "masked-cross-entropy (fused, train.py:54-68)", "optimizer": opt,
            "lambda_reg": hy["decoder_lambda_reg"], "_state": ms, "_hyper": hy, "_C": C}


def build_reconstructor(C=TrainConfig):
    """train.py:163-197."""
    prec = getattr(C, "precision", "bf16")
    if C.reconstructor_type == "local":
        model = LocalReconstructor(model_name=C.reconstructor_model, n_layers=C.reconstructor_n_layers,
                                   decoder_hidden_size=C.decoder_hidden_size,
                                   hidden_size=C.reconstructor_hidden_size, dropout=C.reconstructor_dropout,
                                   decoder_dropout=C.reconstructor_decoder_dropout,
                                   attn_size=C.reconstructor_attn_size, precision=prec)
    elif C.reconstructor_type == "global":
        model = GlobalReconstructor(model_name=C.reconstructor_model, n_layers=C.reconstructor_n_layers,
                                    decoder_hidden_size=C.decoder_hidden_size,
                                    hidden_size=C.reconstructor_hidden_size, dropout=C.reconstructor_dropout,
                                    decoder_dropout=C.reconstructor_decoder_dropout,
                                    caption_max_len=C.caption_max_len, precision=prec)
    else:
        raise NotImplementedError("Unknown reconstructor: {}".format(C.reconstructor_type))
    model = model.to(C.device)
    ms = ModelState(model, C.reconstructor_use_amsgrad)
    hy = _hyper_from(C)
    opt = FusedAdam(ms, 1, lr=C.reconstructor_learning_rate, weight_decay=C.reconstructor_weight_decay,
                    amsgrad=C.reconstructor_use_amsgrad, hyper=hy)
    return {"model": model, "loss": "mse (fused, train.py:101,128)", "optimizer": opt,
            "lambda_reg": hy["reconstructor_lambda_reg"], "_state": ms, "_hyper": hy, "_C": C}


# ----------------------------------------------------------------------------- autograd bridges
def _h(eng):
    return int(eng.handle.value)


class _DecoderSeq(torch.autograd.Function):
    """forward_decoder as a paired forward / backward custom op: torch.ops.recnet.forward_decoder / backward_decoder."""

    @staticmethod
    def forward(ctx, eng, ms, enc, targets, T, stepw, train, seed, *params):
        ops = _ops.load()
        eng._chk_step(enc, targets, T, stepw)
        loss, hid, sc = ops.forward_decoder(_h(eng), enc, targets, T, stepw, bool(train), seed & 0xFFFFFFFF)
        eng.scalars.copy_(sc)
        eng.T = T
        ctx.eng, ctx.ms, ctx.enc, ctx.targets = eng, ms, enc, targets
        return loss, hid

    @staticmethod
    def backward(ctx, dloss, dhid):
        ops, eng = _ops.load(), ctx.eng
        gs = float(dloss) if dloss is not None else 0.0
        ops.backward_decoder(_h(eng), ctx.enc, ctx.targets, None if dhid is None else dhid.contiguous(), gs)
        ops.add_reg_grad(_h(eng), 0, gs)
        ctx.ms.publish_grads()
        return (None,) * 8 + (None,) * len(ctx.ms.params())


class _DecoderSeqFree(torch.autograd.Function):
    """forward_decoder without teacher forcing (train.py:46-51): arg-max fed back on the device.  Differentiable like
    the reference's (the arg-max passes no gradient; the embedding gradient goes to the rows of the tokens fed)."""

    @staticmethod
    def forward(ctx, eng, ms, enc, targets, T, stepw, train, seed, *params):
        ops = _ops.load()
        eng._chk_step(enc, targets, T, stepw)
        loss, hid, out, sc = ops.forward_decoder_free(_h(eng), enc, targets, T, stepw, bool(train), seed & 0xFFFFFFFF)
        eng.scalars.copy_(sc)
        eng.T = T
        ctx.eng, ctx.ms, ctx.enc, ctx.targets = eng, ms, enc, targets
        ctx.mark_non_differentiable(out)
        return loss, hid, out

    @staticmethod
    def backward(ctx, dloss, dhid, _dout):
        ops, eng = _ops.load(), ctx.eng
        gs = float(dloss) if dloss is not None else 0.0
        ops.backward_decoder(_h(eng), ctx.enc, ctx.targets, None if dhid is None else dhid.contiguous(), gs)
        ops.add_reg_grad(_h(eng), 0, gs)
        ctx.ms.publish_grads()
        return (None,) * 8 + (None,) * len(ctx.ms.params())


class _ReconstructorSeq(torch.autograd.Function):
    """forward_{global,local}_reconstructor: torch.ops.recnet.forward_reconstructor / backward_reconstructor."""

    @staticmethod
    def forward(ctx, eng, ms, hiddens, enc, T, train, seed, *params):
        ops = _ops.load()
        eng._chk_rec(enc, hiddens, T)
        loss, sc = ops.forward_reconstructor(_h(eng), enc, hiddens, T, bool(train), seed & 0xFFFFFFFF)
        eng.scalars.copy_(sc)
        eng.T = T
        ctx.eng, ctx.ms, ctx.enc, ctx.T = eng, ms, enc, T
        return loss

    @staticmethod
    def backward(ctx, dloss):
        ops, eng = _ops.load(), ctx.eng
        gs = float(dloss)
        dh = ops.backward_reconstructor(_h(eng), ctx.enc, ctx.T, gs)
        ops.add_reg_grad(_h(eng), 1, gs)
        ctx.ms.publish_grads()
        return (None, None, dh, None, None, None, None) + (None,) * len(ctx.ms.params())


def _engine_for(model_dict, key, factory, which):
    ms = model_dict["_state"]
    eng = ms.engines.get(key)
    if eng is None:
        eng = factory()
        ms.bind(eng, which)
        ms.engines[key] = eng
    return eng


def forward_decoder(decoder, encoder_outputs, targets, target_masks, teacher_forcing_ratio=0., seed=None):
    """train.py:17-75.  Returns (loss, hiddens [T,1,B,H], output_indices).

    Teacher forcing is drawn like the reference does (`random.random() <= teacher_forcing_ratio`, train.py:38).  The
    teacher-forced pass (training: config.py:71 ratio 1.0) is differentiable; output_indices is empty.  The free-running
    pass (validation, train.py:327 — the default ratio 0; training when the draw says so) feeds the arg-max back on
    the device and returns output_indices [T,B]; it is differentiable too (the embedding gradient goes to the tokens
    that were fed)."""
    use_teacher_forcing = random.random() <= teacher_forcing_ratio
    model = decoder["model"]
    B, F = encoder_outputs.shape[0], encoder_outputs.shape[1]
    masks = _host_masks(targets, target_masks)
    cml = decoder["_hyper"]["caption_max_len"]
    T = decode_len(masks, cml)
    stepw = torch.from_numpy(step_weights(masks, T)).to(encoder_outputs.device)
    eng = _engine_for(decoder, ("dec", B, F), lambda: Engine(_dims(model, B, F), None, model.precision,
                                                             decoder["_hyper"], device=encoder_outputs.device), 0)
    eng.pack_weights()
    seed = decoder["_C"].dropout_seed + decoder["_state"].step if seed is None else seed
    decoder["_last_seed"] = seed
    if not use_teacher_forcing:
        return _DecoderSeqFree.apply(eng, decoder["_state"], encoder_outputs.contiguous(), targets.contiguous(), T, stepw,
                                     model.training, seed, *decoder["_state"].params().values())
    loss, hid = _DecoderSeq.apply(eng, decoder["_state"], encoder_outputs.contiguous(), targets.contiguous(), T,
                                  stepw, model.training, seed, *decoder["_state"].params().values())
    return loss, hid, torch.zeros(0, dtype=torch.long)


def _forward_reconstructor(kind, decoder_hiddens, encoder_outputs, reconstructor, seed=None):
    model = reconstructor["model"]
    if model.kind != kind:
        raise ValueError("reconstructor is %s, not %s" % (model.kind, kind))
    T, _, B, H = decoder_hiddens.shape
    F = encoder_outputs.shape[1]
    d = dict(B=B, F=F, D=encoder_outputs.shape[2], E=4, H=H, A=4, V=8, R=model.hidden_size,
             RA=getattr(model, "attn_size", 0), rec_cell=model.model_name)
    eng = _engine_for(reconstructor, ("rec", B, F), lambda: Engine(d, kind, model.precision, reconstructor["_hyper"],
                                                                   device=encoder_outputs.device), 1)
    eng.pack_weights()
    seed = reconstructor["_C"].dropout_seed + reconstructor["_state"].step if seed is None else seed
    return _ReconstructorSeq.apply(eng, reconstructor["_state"], decoder_hiddens.contiguous(),
                                   encoder_outputs.contiguous(), T, model.training, seed,
                                   *reconstructor["_state"].params().values())


def forward_global_reconstructor(decoder_hiddens, encoder_outputs, reconstructor, seed=None):
    """train.py:78-105."""
    return _forward_reconstructor("global", decoder_hiddens, encoder_outputs, reconstructor, seed)


def forward_local_reconstructor(decoder_hiddens, encoder_outputs, reconstructor, seed=None):
    """train.py:108-131."""
    return _forward_reconstructor("local", decoder_hiddens, encoder_outputs, reconstructor, seed)


# ----------------------------------------------------------------------------- fused train step
class TrainStep:
    """The train-step body train.py:248-273 as one launch sequence on the current stream:
    forward decoder, forward reconstructor, backward (reconstructor then decoder BPTT), regulariser
    gradient, decoder clip, AMSGrad / Adam updates, weight re-pack.  `scalars` (device, 8 floats) holds the
    losses (names: _lib.SCALAR_NAMES); reading them is the only host synchronisation and is up to the
    caller (the reference syncs three times per step, train.py:275-277)."""

    def __init__(self, decoder, reconstructor=None, batch_size=None, n_frames=None, global_batch=None,
                 batch_offset=0, teacher_forcing_ratio=None):
        C = decoder["_C"]
        self.decoder, self.reconstructor = decoder, reconstructor
        # config.py:71 decoder_teacher_forcing_ratio, handed to forward_decoder at train.py:251.  Every call of the step draws
        # `random.random() <= ratio` from Python's global generator like train.py:38 does (also at ratio 1, where the draw is
        # always True — the generator advances as the reference's does).  A False draw runs the free-running iteration.
        # (Data parallel with a ratio below 1: rank 0's draw is broadcast, see _draw.  GraphedStep replays the teacher-forced step
        # and makes NO draw — it refuses a ratio below 1 — so Python's generator advances per eager step only.)
        self.teacher_forcing_ratio = float(getattr(C, "decoder_teacher_forcing_ratio", 1.0) if teacher_forcing_ratio is None
                                           else teacher_forcing_ratio)
        self.output_indices = None        # [T, B] tokens the last free-running iteration fed back (train.py:50), else None
        self.sync_draw = None             # data parallel: callable(local draw) -> the draw every rank uses (dp.py)
        dm = decoder["model"]
        rm = reconstructor["model"] if reconstructor else None
        B = batch_size or C.batch_size
        F = n_frames or C.encoder_output_len
        self.B, self.F = B, F
        dev = next(dm.parameters()).device
        hy = dict(decoder["_hyper"])
        self.engine = Engine(_dims(dm, B, F, rm), rm.kind if rm else None, dm.precision, hy, device=dev,
                             global_batch=global_batch or B, batch_offset=batch_offset)
        decoder["_state"].bind(self.engine, 0)
        decoder["_state"].engines[("step", B, F)] = self.engine
        if reconstructor:
            reconstructor["_state"].bind(self.engine, 1)
            reconstructor["_state"].engines[("step", B, F)] = self.engine
        self.engine.pack_weights()
        self.seed_base = C.dropout_seed
        self.caption_max_len = C.caption_max_len

    @property
    def scalars(self):
        return self.engine.scalars

    def _draw(self):
        """The iteration's teacher-forcing draw, train.py:38: one `random.random()` from Python's global generator per call of the
        step (also at ratio 1, where it is always True).  Under data parallelism every rank has to run the same kind of iteration:
        dp.DataParallelTrainStep installs `sync_draw`, which replaces the local draw by rank 0's (each rank still consumes one
        value of its own generator per step)."""
        tf = random.random() <= self.teacher_forcing_ratio
        if self.sync_draw is not None and self.teacher_forcing_ratio < 1.0:
            tf = bool(self.sync_draw(tf))
        return tf

    def prepare(self, targets_host):
        """Host-side, from the batch's (global) targets: (T, step weights as a device tensor)."""
        masks = np.asarray(targets_host) > PAD
        T = decode_len(masks, self.caption_max_len)
        w = torch.from_numpy(step_weights(masks, T)).to(self.engine.device)
        return T, w

    def fwd_bwd(self, enc, targets, T, step_weight, seed=None):
        ms = self.decoder["_state"]
        # dropout seed of optimiser step n (1-based) = seed_base + n, the same rule the device-side counter of
        # the graph-replay path applies (recnet_train_step_fwd_bwd_dev)
        seed = self.seed_base + ms.step + 1 if seed is None else seed
        if self._draw():
            self.output_indices = None
            self.engine.train_step_fwd_bwd(enc, targets, T, step_weight, seed)
        else:
            self._free_fwd_bwd(enc, targets, T, step_weight, seed)

    def optimizer_step(self):
        ms = self.decoder["_state"]
        ms.step += 1
        if self.reconstructor:
            self.reconstructor["_state"].step = ms.step
        self.engine.optimizer_step(ms.step, _lib.OPT_REG | _lib.OPT_CLIP)
        self._mark()

    def _mark(self):
        self.decoder["model"].mark_weights_changed()
        if self.reconstructor:
            self.reconstructor["model"].mark_weights_changed()

    def __call__(self, enc, targets, T, step_weight, seed=None):
        ms = self.decoder["_state"]
        seed = self.seed_base + ms.step + 1 if seed is None else seed
        ms.step += 1
        if self.reconstructor:
            self.reconstructor["_state"].step = ms.step
        if self._draw():
            self.output_indices = None
            self.engine.train_step(enc, targets, T, step_weight, seed, ms.step)
        else:
            self._free_fwd_bwd(enc, targets, T, step_weight, seed)
            self.engine.optimizer_step(ms.step, _lib.OPT_REG | _lib.OPT_CLIP)
        self._mark()
        return self.engine.scalars

    def _free_fwd_bwd(self, enc, targets, T, step_weight, seed):
        """Forward + backward of the iteration of train.py:248-268 whose draw said no teacher forcing: the decoder feeds its own
        arg-max back (train.py:46-51, per-step kernels: the arg-max of step t is the input of step t + 1, so nothing of the
        forward can be batched over time), the reconstructor reads those hidden states, and the backward differentiates that
        unroll — the arg-max passes no gradient, the embedding gradient goes to the rows of the tokens that were fed.  The
        optimiser step that follows is the teacher-forced iteration's (regulariser gradient, decoder clip, AMSGrad / Adam, re-pack)."""
        eng = self.engine
        eng.flush()                                   # (a deferred update left by a replayed graph completes first)
        _, self.output_indices = eng.forward_decoder_free(enc, targets, T, step_weight, train=True, seed=seed)
        dh = None
        if self.reconstructor:
            eng.forward_reconstructor(enc, None, T, train=True, seed=seed)
            dh = eng.backward_reconstructor(enc, grad_scale=float(eng.hyper["lambda_recon"]))      # train.py:260
        eng.backward_decoder(enc, targets, dh, 1.0)


class GraphedStep:
    """Replays the train step from captured hipGraphs (one per fixed T): the step is ~100 (per-step kernels: ~340) dependent
    launches, so eager launching is host-bound; a graph removes the launch overhead.  The optimiser
    step count and the dropout seed advance on the device (recnet_train_step_*_dev), so every replay is
    a new training step.

    One rank: a single graph (fwd + bwd + optimiser).  Data parallel: three graphs with the two gradient
    all-reduces in between,
        graph A (fwd, reconstructor bwd)  ->  all-reduce(reconstructor bucket + decoder out.*, async, RCCL stream)
        graph B (decoder bwd)  [runs while those are on the wire]
        all-reduce(rest of the decoder bucket)  ->  wait all  ->  graph C (regulariser, clip, Adam, re-pack)
    so only the all-reduce of the decoder's recurrent / attention / embedding gradients is exposed."""

    def __init__(self, dp_step, enc, targets, T, step_weight, warmup=2, defer_reconstructor_update=False):
        """defer_reconstructor_update (one rank, global reconstructor; opt-in — measured slower at the benchmark shape, see
        DESIGN.md section 5): every replay leaves the reconstructor's weight-gradient products and
        Adam step pending and the NEXT replay runs them under its decoder forward chain (recnet_hip.h:
        recnet_set_deferred_reconstructor_update).  Call flush() before reading the reconstructor's parameters,
        gradients or optimiser state from Python (state_dict, checkpoints, evaluation); parameters after flush() are
        bit-identical to the non-deferred step's."""
        self.dp = dp_step
        st = dp_step.step_impl
        if st.teacher_forcing_ratio < 1.0:
            # train.py:38 draws per iteration on the host; a replayed graph makes no draw.  TrainStep (eager) runs such a schedule.
            raise ValueError("GraphedStep replays the teacher-forced step; decoder_teacher_forcing_ratio = %g < 1 needs "
                             "TrainStep, which draws per call" % st.teacher_forcing_ratio)
        self.eng = st.engine
        self.enc, self.targets, self.T, self.w = enc, targets, T, step_weight
        self.ms = st.decoder["_state"]
        self.rs = st.reconstructor["_state"] if st.reconstructor else None
        self.seed_base = st.seed_base
        self.flags = _lib.OPT_REG | _lib.OPT_CLIP
        self.split = bool(dp_step.reduce)
        self._comm = None
        eng = self.eng
        recurrent_only = defer_reconstructor_update in (2, "recurrent")
        self.deferred = (bool(defer_reconstructor_update) and not self.split and self.rs is not None and
                         (st.reconstructor["model"].kind == "global" or recurrent_only))
        # "recurrent": only d W_hh and its Adam step are left to the next replay (mode 2 of recnet_set_deferred_reconstructor_update)
        self.defer_mode = ("recurrent" if defer_reconstructor_update in (2, "recurrent") else True) if self.deferred else False
        eng.set_deferred_reconstructor_update(self.defer_mode)
        eng.set_step(self.ms.step)
        # Data parallel: ONE graph with the collectives captured inside it (RCCL kernels are capturable; round 4) — one replay per
        # step instead of three with two eager collectives between them.  It has run on RCCL with ONE rank only (no multi-GPU box
        # was available to any round), so it is the default at world size 1 and OPT-IN (RN_DP_ONE_GRAPH=1) above that, where the
        # three-graph form with eager collectives is the default.  A backend whose collectives cannot be captured (gloo in the CPU
        # tests) and the direct reduce-scatter transport (it stages through torch ops on a stream of its own) always take the
        # three-graph form.
        self.one_graph = False
        want_one = False
        if self.split:
            import os as _os
            import torch.distributed as _dist
            nccl = _dist.is_available() and _dist.is_initialized() and _dist.get_backend(dp_step.group) == "nccl"
            world = _dist.get_world_size(dp_step.group) if nccl else 1
            want_one = bool(int(_os.environ.get("RN_DP_ONE_GRAPH", "1" if world == 1 else "0")) and self.enc.is_cuda and
                            dp_step.transport.algo == "ring" and nccl)
            if nccl and world > 1:
                # rank 0's choice is everyone's (the environment is per rank; ranks that disagree about the form would
                # hang in the agreement all-reduce below, which only the ranks that want one graph entered)
                flag = torch.tensor([1 if want_one else 0], device=self.enc.device, dtype=torch.int32)
                _dist.broadcast(flag, src=_dist.get_global_rank(dp_step.group, 0) if hasattr(_dist, "get_global_rank") else 0, group=dp_step.group)
                want_one = bool(int(flag.item()))
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):                      # warm-up outside capture (lazy module loading, RCCL init)
            for _ in range(warmup):
                if want_one:                               # the very sequence the capture records: its first execution is eager
                    self._dp_body()
                    self._bump()
                else:
                    self._eager()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        # thread_local: the RCCL watchdog thread of torch.distributed issues event queries of its own; in the
        # default (global) mode those would invalidate an ongoing capture
        mode = dict(capture_error_mode="thread_local")
        self.graphs = []
        if not self.split:
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, **mode):
                eng.train_step_dev(self.enc, self.targets, self.T, self.w, self.seed_base, self.flags)
            self.graphs = [g]
        else:
            import torch.distributed as _dist
            if want_one:
                ok = 1
                try:
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g, **mode):
                        self._dp_body()
                except Exception as e:          # noqa: BLE001  (the capture is abandoned; nothing ran)
                    import warnings
                    warnings.warn("one-graph data-parallel step unavailable on this rank (%s)" % (e,))
                    ok = 0
                    torch.cuda.synchronize()
                    eng.abort_step()             # the handle's enqueue-time flags back to "between two steps"
                # every rank takes the SAME form: one rank replaying captured collectives against peers that issue eager ones
                # would deadlock or, worse, pair up the wrong buffers
                flag = torch.tensor([ok], device=self.enc.device, dtype=torch.int32)
                _dist.all_reduce(flag, op=_dist.ReduceOp.MIN, group=dp_step.group)
                torch.cuda.synchronize()
                if int(flag.item()) == 1:
                    self.graphs = [g]
                    self.one_graph = True
                else:
                    self.graphs = []
                    if ok:
                        import warnings
                        warnings.warn("one-graph data-parallel step unavailable on a peer rank: three graphs + eager collectives")
            if not self.one_graph:
                for part in (1, 2):
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g, **mode):
                        eng.train_step_part_dev(part, self.enc, self.targets, self.T, self.w, self.seed_base)
                    self.graphs.append(g)
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, **mode):
                    eng.optimizer_step_dev(self.flags)
                self.graphs.append(g)

    def _dp_body(self):
        """The data-parallel step in stream order: part 1 (forward, reconstructor backward) -> early buckets on the wire ->
        part 2 (decoder BPTT + its weight gradients) -> late bucket -> wait -> optimiser.  Eager, or captured as one graph."""
        # the early buckets go out from a stream of their own that waits for the library's side stream (the reconstructor's weight
        # gradients are still being formed there): this stream goes straight on to the decoder's BPTT
        cur = torch.cuda.current_stream()
        if self._comm is None:
            self._comm = torch.cuda.Stream()
        self.eng.set_dp_overlap(True)
        try:
            self.eng.train_step_part_dev(1, self.enc, self.targets, self.T, self.w, self.seed_base)
        finally:
            self.eng.set_dp_overlap(False)
        self._comm.wait_stream(cur)
        with torch.cuda.stream(self._comm):
            self.eng.join_side()
            works = self._reduce_async(self.dp.early_buffers())
        # (Tried in round 4 and left out: the reconstructor's Adam step on that stream as soon as its bucket is reduced, i.e. beside
        # the BPTT like in the single-rank step — hipStreamEndCapture of the resulting graph segfaults in the HIP runtime of
        # ROCm 7.2, so both optimiser steps stay at the end.)
        self.eng.train_step_part_dev(2, self.enc, self.targets, self.T, self.w, self.seed_base)
        works += self._reduce_async(self.dp.late_buffers())
        for w in works:
            self.dp.transport.finish(w)
        cur.wait_stream(self._comm)
        self.eng.optimizer_step_dev(self.flags)

    def flush(self):
        """Completes a pending deferred reconstructor update (stream-ordered, no host sync); a no-op otherwise."""
        self.eng.flush()

    def _reduce_async(self, bufs):
        return [self.dp.transport.start(b) for b in bufs]

    def _eager(self):
        if not self.split:
            self.eng.train_step_dev(self.enc, self.targets, self.T, self.w, self.seed_base, self.flags)
            self._bump()
            return
        else:
            self.eng.train_step_part_dev(1, self.enc, self.targets, self.T, self.w, self.seed_base)
            works = self._reduce_async(self.dp.early_buffers())
            self.eng.train_step_part_dev(2, self.enc, self.targets, self.T, self.w, self.seed_base)
            works += self._reduce_async(self.dp.late_buffers())
            for w in works:
                self.dp.transport.finish(w)
        self.eng.optimizer_step_dev(self.flags)
        self._bump()

    def _bump(self):
        self.ms.step += 1
        self.ms.model.mark_weights_changed()
        if self.rs:
            self.rs.step = self.ms.step
            self.rs.model.mark_weights_changed()

    def __call__(self):
        if not self.split:
            self.graphs[0].replay()
            self.eng.mark_pending()          # (a replay runs no host code: the handle's lazily refreshed weight images / a deferred update)
        elif self.one_graph:
            self.graphs[0].replay()
        else:
            ga, gb, gc = self.graphs
            ga.replay()
            works = self._reduce_async(self.dp.early_buffers())      # reconstructor bucket + the decoder's output layer
            gb.replay()                                              # decoder BPTT while they are on the wire
            works += self._reduce_async(self.dp.late_buffers())
            for w in works:
                self.dp.transport.finish(w)
            gc.replay()
        self._bump()
        return self.eng.scalars
