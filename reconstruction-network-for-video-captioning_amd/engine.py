"""Engine: one recnet_handle (C ABI) plus the device memory it works in.

PyTorch is used here for what the task allows it for — device allocations, the current stream
and (in dp.py) torch.distributed.  Every computation is a call into librecnet_hip.so.
"""
import ctypes as C

import torch

from . import _lib
from ._lib import DECODER_KEYS, REC_KEYS

_KIND = {None: _lib.REC_NONE, "none": _lib.REC_NONE, "global": _lib.REC_GLOBAL, "local": _lib.REC_LOCAL}
_CELL = {"LSTM": 0, "GRU": 1}
_PREC = {"f32": _lib.PREC_F32, "fp32": _lib.PREC_F32, "bf16": _lib.PREC_BF16}


def _ptr(t):
    return C.c_void_p(0 if t is None else t.data_ptr())


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _chk_tensor(t, shape, dtype, name):
    if not t.is_cuda:
        raise RuntimeError("%s must be a CUDA (HIP) tensor — the RecNet HIP path has no CPU fallback" % name)
    if t.dtype != dtype or tuple(t.shape) != tuple(shape) or not t.is_contiguous():
        raise RuntimeError("%s: expected contiguous %s %s, got %s %s" % (name, dtype, tuple(shape), t.dtype,
                                                                          tuple(t.shape)))


class FlatState:
    """A set of named fp32 tensors carved out of ONE flat device buffer (so a data-parallel
    all-reduce is a single collective over `flat`)."""

    def __init__(self, shapes, device):
        self.shapes = dict(shapes)
        n = 0
        self.offsets = {}
        for k, s in self.shapes.items():
            self.offsets[k] = n
            cnt = 1
            for d in s:
                cnt *= d
            n += (cnt + 3) // 4 * 4          # keep every tensor 16-byte aligned
        self.flat = torch.zeros(n, dtype=torch.float32, device=device)
        self.views = {}
        for k, s in self.shapes.items():
            cnt = 1
            for d in s:
                cnt *= d
            self.views[k] = self.flat[self.offsets[k]:self.offsets[k] + cnt].view(s)

    def struct(self, cls, keys):
        st = cls()
        for k in keys:
            setattr(st, k.replace(".", "_"), self.views[k].data_ptr() if k in self.views else None)
        return st


def _struct_from(cls, keys, tensors):
    st = cls()
    for k in keys:
        t = tensors.get(k)
        setattr(st, k.replace(".", "_"), None if t is None else t.data_ptr())
    return st


class Engine:
    """dims: dict with B,F,D,E,H,A,V and (optional) R, RA.  kind: None | 'global' | 'local'."""

    def __init__(self, dims, kind=None, precision="bf16", hyper=None, device=None, global_batch=None,
                 batch_offset=0):
        self.lib = _lib.load()
        if not torch.cuda.is_available():
            raise RuntimeError("RecNet HIP engine needs a GPU (torch.cuda.is_available() is False)")
        self.device = torch.device(device if device is not None else "cuda")
        hy = dict(embedding_scale=1.0, embedding_dropout=0.5, decoder_out_dropout=0.5,
                  reconstructor_decoder_dropout=0.5, decoder_learning_rate=1e-5, reconstructor_learning_rate=1e-6,
                  decoder_weight_decay=1e-5, reconstructor_weight_decay=1e-5, decoder_use_amsgrad=True,
                  reconstructor_use_amsgrad=False, gradient_clip=50.0, decoder_lambda_reg=1e-3,
                  reconstructor_lambda_reg=1e-2, lambda_recon=1.0, adam_beta1=0.9, adam_beta2=0.999, adam_eps=1e-8,
                  caption_max_len=30)
        hy.update(hyper or {})
        self.hyper = hy
        self.dims = dict(dims)
        self.kind = kind if kind not in ("none",) else None
        self.precision = precision
        c = _lib.Config()
        c.batch_size = dims["B"]; c.encoder_output_len = dims["F"]; c.encoder_output_size = dims["D"]
        c.embedding_size = dims["E"]; c.decoder_hidden_size = dims["H"]; c.decoder_attn_size = dims["A"]
        c.n_vocabs = dims["V"]
        c.reconstructor_hidden_size = dims.get("R", dims["D"]) if self.kind else 0
        c.reconstructor_attn_size = dims.get("RA", 0) if self.kind == "local" else 0
        c.caption_max_len = hy["caption_max_len"]
        c.reconstructor_type = _KIND[self.kind]
        c.precision = _PREC[precision]
        c.global_batch_size = global_batch if global_batch else dims["B"]
        c.batch_offset = batch_offset
        c.decoder_cell = _CELL[dims.get("dec_cell", "LSTM")]
        c.reconstructor_cell = _CELL[dims.get("rec_cell", "LSTM")]
        c.decoder_attn_normalize = {"none": 0, "softmax": 1}[dims.get("attn_normalize", "none")]
        c.decoder_use_amsgrad = int(bool(hy["decoder_use_amsgrad"]))
        c.reconstructor_use_amsgrad = int(bool(hy["reconstructor_use_amsgrad"]))
        for k in ("embedding_scale", "embedding_dropout", "decoder_out_dropout", "reconstructor_decoder_dropout",
                  "decoder_lambda_reg", "reconstructor_lambda_reg", "lambda_recon", "decoder_learning_rate",
                  "reconstructor_learning_rate", "decoder_weight_decay", "reconstructor_weight_decay", "adam_beta1",
                  "adam_beta2", "adam_eps"):
            setattr(c, k, float(hy[k]))
        c.gradient_clip = float(hy["gradient_clip"] or 0.0)
        self.cfg = c
        self.handle = C.c_void_p()
        _lib.check(self.lib.recnet_create(C.byref(c), C.byref(self.handle)), "recnet_create")
        nbytes = self.lib.recnet_workspace_bytes(self.handle)
        with torch.cuda.device(self.device):
            self.workspace = torch.empty(nbytes + 256, dtype=torch.uint8, device=self.device)
            off = (-self.workspace.data_ptr()) % 256
            self._ws_ptr = self.workspace.data_ptr() + off
            _lib.check(self.lib.recnet_bind_workspace(self.handle, C.c_void_p(self._ws_ptr), C.c_size_t(nbytes)),
                       "recnet_bind_workspace")
            self.scalars = torch.zeros(8, dtype=torch.float32, device=self.device)
        self._keep = []
        self.T = None

    def __del__(self):
        try:
            if getattr(self, "handle", None) is not None and self.handle.value:
                self.lib.recnet_destroy(self.handle)
                self.handle = C.c_void_p()
        except Exception:
            pass

    # ------------------------------------------------------------------ binding
    def decoder_shapes(self):
        d = self.dims
        G = 3 if d.get("dec_cell", "LSTM") == "GRU" else 4
        return {"attn_b": (d["A"],), "embedding.weight": (d["V"], d["E"]), "attn_W.weight": (d["A"], d["H"]),
                "attn_U.weight": (d["A"], d["D"]), "attn_w.weight": (1, d["A"]),
                "rnn.weight_ih_l0": (G * d["H"], d["E"] + d["D"]), "rnn.weight_hh_l0": (G * d["H"], d["H"]),
                "rnn.bias_ih_l0": (G * d["H"],), "rnn.bias_hh_l0": (G * d["H"],),
                "out.weight": (d["V"], d["H"]), "out.bias": (d["V"],)}

    def rec_shapes(self):
        d = self.dims
        R, H = d.get("R", d["D"]), d["H"]
        G = 3 if d.get("rec_cell", "LSTM") == "GRU" else 4
        s = {}
        if self.kind == "local":
            RA = d["RA"]
            s.update({"attn_b": (RA,), "attn_W.weight": (RA, R), "attn_U.weight": (RA, H), "attn_w.weight": (1, RA)})
        s.update({"rnn.weight_ih_l0": (G * R, H if self.kind == "local" else 2 * H), "rnn.weight_hh_l0": (G * R, R),
                  "rnn.bias_ih_l0": (G * R,), "rnn.bias_hh_l0": (G * R,), "out.weight": (R, R), "out.bias": (R,)})
        return s

    def bind_decoder(self, params, grads=None, exp_avg=None, exp_avg_sq=None, max_exp_avg_sq=None):
        """params etc.: dict key -> CUDA fp32 tensor (state_dict names)."""
        shp = self.decoder_shapes()
        for k in DECODER_KEYS:
            _chk_tensor(params[k], shp[k], torch.float32, "decoder." + k)
        sts = [_struct_from(_lib.DecoderTensors, DECODER_KEYS, params)]
        for grp in (grads, exp_avg, exp_avg_sq, max_exp_avg_sq):
            if grp is not None:
                for k in DECODER_KEYS:
                    _chk_tensor(grp[k], shp[k], torch.float32, "decoder state " + k)
            sts.append(None if grp is None else _struct_from(_lib.DecoderTensors, DECODER_KEYS, grp))
        self._keep.append((params, grads, exp_avg, exp_avg_sq, max_exp_avg_sq))
        args = [C.byref(s) if s is not None else None for s in sts]
        with torch.cuda.device(self.device):
            _lib.check(self.lib.recnet_bind_decoder(self.handle, *args), "recnet_bind_decoder")

    def bind_reconstructor(self, params, grads=None, exp_avg=None, exp_avg_sq=None, max_exp_avg_sq=None):
        shp = self.rec_shapes()
        for k in shp:
            _chk_tensor(params[k], shp[k], torch.float32, "reconstructor." + k)
        sts = [_struct_from(_lib.ReconstructorTensors, REC_KEYS, params)]
        for grp in (grads, exp_avg, exp_avg_sq, max_exp_avg_sq):
            if grp is not None:
                for k in shp:
                    _chk_tensor(grp[k], shp[k], torch.float32, "reconstructor state " + k)
            sts.append(None if grp is None else _struct_from(_lib.ReconstructorTensors, REC_KEYS, grp))
        self._keep.append((params, grads, exp_avg, exp_avg_sq, max_exp_avg_sq))
        args = [C.byref(s) if s is not None else None for s in sts]
        with torch.cuda.device(self.device):
            _lib.check(self.lib.recnet_bind_reconstructor(self.handle, *args), "recnet_bind_reconstructor")

    def set_shard(self, global_batch, batch_offset):
        _lib.check(self.lib.recnet_set_shard(self.handle, global_batch, batch_offset), "recnet_set_shard")

    # ------------------------------------------------------------------ hot path
    def pack_weights(self):
        _lib.check(self.lib.recnet_pack_weights(self.handle, _stream()), "recnet_pack_weights")

    def decoder_prepare(self, enc):
        d = self.dims
        _chk_tensor(enc, (d["B"], d["F"], d["D"]), torch.float32, "encoder_outputs")
        _lib.check(self.lib.recnet_decoder_prepare(self.handle, _ptr(enc), _stream()), "recnet_decoder_prepare")

    def greedy_search(self, enc):
        """eval.py:19-33 on the device.  Returns (tokens [Tm, B] int64, n_steps int32[1]) device tensors."""
        d = self.dims
        _chk_tensor(enc, (d["B"], d["F"], d["D"]), torch.float32, "encoder_outputs")
        Tm = self.hyper["caption_max_len"] + 1
        toks = torch.zeros(Tm, d["B"], dtype=torch.int64, device=self.device)
        n = torch.zeros(1, dtype=torch.int32, device=self.device)
        _lib.check(self.lib.recnet_greedy_search(self.handle, _ptr(enc), _ptr(toks), _ptr(n), _stream()),
                   "recnet_greedy_search")
        return toks, n

    def beam_search(self, enc, beam_width):
        """eval.py:36-120 on the device.  Returns (best [Tm, B] int64, n_steps int32[1])."""
        d = self.dims
        _chk_tensor(enc, (d["B"], d["F"], d["D"]), torch.float32, "encoder_outputs")
        Tm = self.hyper["caption_max_len"] + 1
        best = torch.zeros(Tm, d["B"], dtype=torch.int64, device=self.device)
        n = torch.zeros(1, dtype=torch.int32, device=self.device)
        _lib.check(self.lib.recnet_beam_search(self.handle, _ptr(enc), int(beam_width), _ptr(best), _ptr(n), _stream()),
                   "recnet_beam_search")
        return best, n

    def decoder_step(self, tokens, h_in, c_in, enc, train=False, seed=0, t=0):
        d = self.dims
        B = d["B"]
        _chk_tensor(tokens, (B,), torch.int64, "tokens")
        for nm, st in (("h", h_in), ("c", c_in)):
            if st is not None:
                _chk_tensor(st, (B, d["H"]), torch.float32, "hidden state " + nm)
        if enc is not None:
            _chk_tensor(enc, (B, d["F"], d["D"]), torch.float32, "encoder_outputs")
        logits = torch.empty(B, d["V"], dtype=torch.float32, device=self.device)
        h_out = torch.empty(B, d["H"], dtype=torch.float32, device=self.device)
        c_out = torch.empty_like(h_out)
        _lib.check(self.lib.recnet_decoder_step(self.handle, _ptr(tokens), _ptr(h_in), _ptr(c_in), _ptr(enc),
                                                _ptr(logits), _ptr(h_out), _ptr(c_out), int(train), seed & 0xFFFFFFFF,
                                                int(t), _stream()), "recnet_decoder_step")
        return logits, h_out, c_out

    def reconstructor_step(self, inp, hr_in, cr_in, decoder_hiddens, T, train=False, seed=0, t=0):
        """One step of GlobalReconstructor.forward / LocalReconstructor.forward (global_reconstructor.py:30-46,
        local_reconstructor.py:37-55).  inp [B,H] (global only), hr_in / cr_in [B,R] or None (zero state),
        decoder_hiddens [T,1,B,H] or None (reuse the previous call's loop invariants) -> (out [B,R], hr', cr')."""
        d = self.dims
        B, R = d["B"], d.get("R", d["D"])
        for nm, x in (("hr", hr_in), ("cr", cr_in)):
            if x is not None:
                _chk_tensor(x, (B, R), torch.float32, "reconstructor state " + nm)
        if inp is not None:
            _chk_tensor(inp, (B, d["H"]), torch.float32, "input")
        if decoder_hiddens is not None:
            _chk_tensor(decoder_hiddens, (T, 1, B, d["H"]), torch.float32, "decoder_hiddens")
        out = torch.empty(B, R, dtype=torch.float32, device=self.device)
        hr = torch.empty_like(out)
        cr = torch.empty_like(out)
        _lib.check(self.lib.recnet_reconstructor_step(self.handle, _ptr(inp), _ptr(hr_in), _ptr(cr_in),
                                                      _ptr(decoder_hiddens), int(T), _ptr(out), _ptr(hr), _ptr(cr),
                                                      int(train), seed & 0xFFFFFFFF, int(t), _stream()),
                   "recnet_reconstructor_step")
        return out, hr, cr

    def poison_lds(self):
        """Test hook: NaN patterns into the LDS of every CU (uninitialised-LDS reads then show up in the parity tests)."""
        _lib.check(self.lib.recnet_debug_poison_lds(self.handle, _stream()), "recnet_debug_poison_lds")

    def chain_status(self):
        """Bit mask of persistent chain kernels that gave up a bounded wait (0 = healthy).  Synchronises the stream."""
        s = C.c_int32(0)
        _lib.check(self.lib.recnet_chain_status(self.handle, C.byref(s), _stream()), "recnet_chain_status")
        return s.value

    def debug_raise_give_up(self, chain_bit=1):
        """Test hook: raise chain `chain_bit`'s sticky word and the poison word the way a chain kernel that gives up does."""
        _lib.check(self.lib.recnet_debug_raise_give_up(self.handle, int(chain_bit), _stream()), "recnet_debug_raise_give_up")

    def set_deferred_reconstructor_update(self, on):
        """Opt in to / out of the deferred reconstructor update of the fused step (include/recnet_hip.h); completes a
        pending update first."""
        mode = 2 if on in (2, "recurrent") else int(bool(on))      # 2: only the recurrent weights' update is deferred
        _lib.check(self.lib.recnet_set_deferred_reconstructor_update(self.handle, mode, _stream()), "recnet_set_deferred_reconstructor_update")

    def set_dp_overlap(self, on):
        _lib.check(self.lib.recnet_set_dp_overlap(self.handle, int(bool(on))), "recnet_set_dp_overlap")

    def step_ring(self):
        """[(start, end)] of the last (up to seven) replayed steps in microseconds, oldest first (recnet_read_step_ring)."""
        buf = (C.c_uint64 * 16)()
        _lib.check(self.lib.recnet_read_step_ring(self.handle, buf, _stream()), "recnet_read_step_ring")
        ns, ne = int(buf[0]), int(buf[8])
        k = min(ns, ne, 7)
        out = []
        for i in range(k):
            a, b = ns - k + i, ne - k + i
            out.append((buf[1 + a % 7] * 0.01, buf[9 + b % 7] * 0.01))
        return out

    def abort_step(self):
        """After an abandoned stream capture: the handle's enqueue-time bookkeeping back to "between two steps" (recnet_abort_step)."""
        _lib.check(self.lib.recnet_abort_step(self.handle), "recnet_abort_step")

    def join_side(self):
        """The current stream waits for the library's side stream (recnet_join_side)."""
        _lib.check(self.lib.recnet_join_side(self.handle, _stream()), "recnet_join_side")

    def mark_pending(self):
        _lib.check(self.lib.recnet_mark_pending(self.handle), "recnet_mark_pending")

    def flush(self):
        """Completes a pending deferred reconstructor update on the current stream (stream-ordered; no host sync)."""
        _lib.check(self.lib.recnet_flush(self.handle, _stream()), "recnet_flush")

    def debug_tensor(self, which, n):
        """Test hook: n floats of a saved tensor of the local reconstructor's forward pass (recnet_debug_offset)."""
        off = self.lib.recnet_debug_offset(self.handle, int(which))
        assert off >= 0
        base = (self._ws_ptr - self.workspace.data_ptr()) + off
        return self.workspace[base:base + 4 * n].view(torch.float32).clone()

    def images_stale(self):
        """Test hook: number of 16-bit words in which the packed operand images differ from a fresh re-pack of the master
        parameters (recnet_debug_images_stale); completes a pending update and synchronises."""
        nb = int(self.lib.recnet_debug_images_bytes(self.handle))
        assert nb > 0
        scratch = torch.empty(nb + 256, dtype=torch.uint8, device=self.device)      # (the caller's allocator, like every other buffer)
        base = (scratch.data_ptr() + 255) // 256 * 256
        n = C.c_int64(-1)
        _lib.check(self.lib.recnet_debug_images_stale(self.handle, C.c_void_p(base), nb, C.byref(n), _stream()), "recnet_debug_images_stale")
        return int(n.value)

    def debug_occupy(self, n_workgroups, microseconds, stream=None):
        """Test hook: n_workgroups CU-filling workgroups spinning for `microseconds` on `stream` (a torch stream; default: the current one)."""
        st = C.c_void_p(stream.cuda_stream) if stream is not None else _stream()
        _lib.check(self.lib.recnet_debug_occupy(self.handle, int(n_workgroups), int(microseconds), st), "recnet_debug_occupy")

    def chain_reset(self, disable_persistent=True):
        _lib.check(self.lib.recnet_chain_reset(self.handle, int(bool(disable_persistent)), _stream()), "recnet_chain_reset")

    def forward_decoder(self, enc, targets, T, step_weight, train=True, seed=0, want_hiddens=True):
        d = self.dims
        B = d["B"]
        _chk_tensor(enc, (B, d["F"], d["D"]), torch.float32, "encoder_outputs")
        _chk_tensor(targets, (self.hyper["caption_max_len"] + 1, B), torch.int64, "targets")
        _chk_tensor(step_weight, (T,), torch.float32, "step_weight")
        hid = torch.empty(T, 1, B, d["H"], dtype=torch.float32, device=self.device) if want_hiddens else None
        _lib.check(self.lib.recnet_forward_decoder(self.handle, _ptr(enc), _ptr(targets), int(T), _ptr(step_weight),
                                                   int(train), seed & 0xFFFFFFFF, _ptr(hid), _ptr(self.scalars),
                                                   _stream()), "recnet_forward_decoder")
        self.T = T
        return hid

    def forward_decoder_free(self, enc, targets, T, step_weight, train=False, seed=0):
        """Free-running pass (train.py:46-51): returns (hiddens [T,1,B,H], output_indices [T,B] int64).  Forward only."""
        d = self.dims
        B = d["B"]
        _chk_tensor(enc, (B, d["F"], d["D"]), torch.float32, "encoder_outputs")
        _chk_tensor(targets, (self.hyper["caption_max_len"] + 1, B), torch.int64, "targets")
        _chk_tensor(step_weight, (T,), torch.float32, "step_weight")
        hid = torch.empty(T, 1, B, d["H"], dtype=torch.float32, device=self.device)
        out = torch.empty(T, B, dtype=torch.int64, device=self.device)
        _lib.check(self.lib.recnet_forward_decoder_free(self.handle, _ptr(enc), _ptr(targets), int(T), _ptr(step_weight),
                                                        int(train), seed & 0xFFFFFFFF, _ptr(hid), _ptr(out),
                                                        _ptr(self.scalars), _stream()), "recnet_forward_decoder_free")
        self.T = T
        return hid, out

    def forward_reconstructor(self, enc, hiddens, T, train=True, seed=0):
        d = self.dims
        _chk_tensor(enc, (d["B"], d["F"], d["D"]), torch.float32, "encoder_outputs")
        if hiddens is not None:
            _chk_tensor(hiddens, (T, 1, d["B"], d["H"]), torch.float32, "decoder_hiddens")
        _lib.check(self.lib.recnet_forward_reconstructor(self.handle, _ptr(enc), _ptr(hiddens), int(T), int(train),
                                                         seed & 0xFFFFFFFF, _ptr(self.scalars), _stream()),
                   "recnet_forward_reconstructor")
        self.T = T

    def backward_reconstructor(self, enc, grad_scale=1.0, want_dhiddens=True):
        d = self.dims
        _chk_tensor(enc, (d["B"], d["F"], d["D"]), torch.float32, "encoder_outputs")
        dh = torch.empty(self.T, 1, d["B"], d["H"], dtype=torch.float32, device=self.device) if want_dhiddens else None
        _lib.check(self.lib.recnet_backward_reconstructor(self.handle, _ptr(enc), float(grad_scale), _ptr(dh),
                                                          _stream()), "recnet_backward_reconstructor")
        return dh

    def backward_decoder(self, enc, targets, dhiddens=None, grad_scale=1.0):
        d = self.dims
        _chk_tensor(enc, (d["B"], d["F"], d["D"]), torch.float32, "encoder_outputs")
        _chk_tensor(targets, (self.hyper["caption_max_len"] + 1, d["B"]), torch.int64, "targets")
        if dhiddens is not None:
            _chk_tensor(dhiddens, (self.T, 1, d["B"], d["H"]), torch.float32, "dhiddens")
        _lib.check(self.lib.recnet_backward_decoder(self.handle, _ptr(enc), _ptr(targets), _ptr(dhiddens),
                                                    float(grad_scale), _stream()), "recnet_backward_decoder")

    def add_reg_grad(self, which, grad_scale=1.0):
        _lib.check(self.lib.recnet_add_reg_grad(self.handle, int(which), float(grad_scale), _stream()),
                   "recnet_add_reg_grad")

    def clip_grad_norm(self, which, max_norm):
        out = torch.empty(1, dtype=torch.float32, device=self.device)
        _lib.check(self.lib.recnet_clip_grad_norm(self.handle, int(which), float(max_norm), _ptr(out), _stream()),
                   "recnet_clip_grad_norm")
        return out

    def optimizer_step(self, step, flags):
        _lib.check(self.lib.recnet_optimizer_step(self.handle, int(step), int(flags), _ptr(self.scalars), _stream()),
                   "recnet_optimizer_step")

    def _chk_step(self, enc, targets, T, step_weight):
        """The raw pointers go straight to the device: shapes, dtypes and residency are checked here."""
        d = self.dims
        _chk_tensor(enc, (d["B"], d["F"], d["D"]), torch.float32, "encoder_outputs")
        _chk_tensor(targets, (self.hyper["caption_max_len"] + 1, d["B"]), torch.int64, "targets")
        _chk_tensor(step_weight, (int(T),), torch.float32, "step_weight")

    def _chk_rec(self, enc, hiddens, T):
        from ._lib import RecNetError  # noqa: F401
        d = self.dims
        _chk_tensor(enc, (d["B"], d["F"], d["D"]), torch.float32, "encoder_outputs")
        if hiddens is not None:
            _chk_tensor(hiddens, (int(T), 1, d["B"], d["H"]), torch.float32, "decoder_hiddens")

    def train_step_fwd_bwd(self, enc, targets, T, step_weight, seed):
        self._chk_step(enc, targets, T, step_weight)
        _lib.check(self.lib.recnet_train_step_fwd_bwd(self.handle, _ptr(enc), _ptr(targets), int(T),
                                                      _ptr(step_weight), seed & 0xFFFFFFFF, _ptr(self.scalars),
                                                      _stream()), "recnet_train_step_fwd_bwd")
        self.T = T

    def train_step(self, enc, targets, T, step_weight, seed, step):
        self._chk_step(enc, targets, T, step_weight)
        _lib.check(self.lib.recnet_train_step(self.handle, _ptr(enc), _ptr(targets), int(T), _ptr(step_weight),
                                              seed & 0xFFFFFFFF, int(step), _ptr(self.scalars), _stream()),
                   "recnet_train_step")
        self.T = T

    # ---- hipGraph-replay-friendly forms (step counter and dropout seed live on the device)
    def set_step(self, step):
        _lib.check(self.lib.recnet_set_step(self.handle, int(step), _stream()), "recnet_set_step")

    def train_step_fwd_bwd_dev(self, enc, targets, T, step_weight, seed_base):
        self._chk_step(enc, targets, T, step_weight)
        _lib.check(self.lib.recnet_train_step_fwd_bwd_dev(self.handle, _ptr(enc), _ptr(targets), int(T),
                                                          _ptr(step_weight), seed_base & 0xFFFFFFFF,
                                                          _ptr(self.scalars), _stream()),
                   "recnet_train_step_fwd_bwd_dev")
        self.T = T

    def train_step_dev(self, enc, targets, T, step_weight, seed_base, flags):
        """Fused forward + backward + optimiser, step count / seed on the device (graph replay, single rank)."""
        self._chk_step(enc, targets, T, step_weight)
        _lib.check(self.lib.recnet_train_step_dev(self.handle, _ptr(enc), _ptr(targets), int(T), _ptr(step_weight),
                                                  seed_base & 0xFFFFFFFF, int(flags), _ptr(self.scalars), _stream()),
                   "recnet_train_step_dev")
        self.T = T

    def train_step_part_dev(self, part, enc, targets, T, step_weight, seed_base):
        self._chk_step(enc, targets, T, step_weight)
        _lib.check(self.lib.recnet_train_step_part_dev(self.handle, int(part), _ptr(enc), _ptr(targets), int(T),
                                                       _ptr(step_weight), seed_base & 0xFFFFFFFF, _ptr(self.scalars),
                                                       _stream()), "recnet_train_step_part_dev")
        self.T = T

    def optimizer_step_dev(self, flags):
        _lib.check(self.lib.recnet_optimizer_step_dev(self.handle, int(flags), _ptr(self.scalars), _stream()),
                   "recnet_optimizer_step_dev")

    # ---- live measurement of one recurrent-step GEMM site with HIP events on its own stream
    def profile_site(self, site, fn, iters=3):
        """Runs fn() `iters` times with hipEvents around every launch of `site` (1 dec fwd, 2 dec bwd, 3 rec
        fwd, 4 rec bwd, 5 local-attention, 7-10 the chain kernels); returns (launches, average ms per launch).  fn() launches
        EAGERLY: on a capturing stream no bracket is taken (csrc/host_common.inc: prof_take)."""
        _lib.check(self.lib.recnet_profile_begin(self.handle, int(site)), "recnet_profile_begin")
        for _ in range(iters):
            fn()
        n, ms = C.c_int32(0), C.c_double(0.0)
        _lib.check(self.lib.recnet_profile_end(self.handle, C.byref(n), C.byref(ms)), "recnet_profile_end")
        return n.value, (ms.value / n.value if n.value else 0.0)

    def profile_null_launch(self, count=1):
        _lib.check(self.lib.recnet_profile_null_launch(self.handle, int(count), _stream()), "recnet_profile_null_launch")

    def read_stamps(self):
        """Phase stamps of the last train step (recnet_read_stamps): dict of microsecond offsets from the step's start —
        `chains[name] = (begin, end)` for the chain kernels that ran inside it, `end` = the step's last kernel.  Written by the
        step's own kernels, so it describes a REPLAYED graph with no tracer attached.  Synchronises."""
        buf = (C.c_uint64 * 30)()
        _lib.check(self.lib.recnet_read_stamps(self.handle, buf, 30, _stream()), "recnet_read_stamps")
        t0, t1 = buf[0], buf[13]
        names = ("decoder_forward", "decoder_bptt", "global_forward", "global_backward", "local_forward", "local_backward")
        chains = {}
        for k, nm in enumerate(names):
            b, e = buf[1 + 2 * k], buf[2 + 2 * k]
            if t0 <= b <= e <= t1:                       # ran inside this step
                chains[nm] = ((b - t0) / 100.0, (e - t0) / 100.0)
        groups = {}
        for j, nm in enumerate(("prologue_products", "reconstructor_weight_gradients", "pending_recurrent_update", "decoder_weight_gradients")):
            b, e = buf[14 + 2 * j], buf[15 + 2 * j]
            if t0 <= b <= e <= t1:
                groups[nm] = ((b - t0) / 100.0, (e - t0) / 100.0)
        return {"end": (t1 - t0) / 100.0 if t1 >= t0 else None, "chains": chains, "groups": groups}

    def recurrent_step_bytes(self, which):
        return float(self.lib.recnet_recurrent_step_bytes(self.handle, int(which)))

    def chain_exchange_bytes(self, which):
        return float(self.lib.recnet_chain_exchange_bytes(self.handle, int(which)))

    def scalar_dict(self):
        """Host copy of the device scalars (synchronises)."""
        v = self.scalars.tolist()
        return dict(zip(_lib.SCALAR_NAMES, v))

    def gemm_bf16(self, A, B, a_col=False, b_col=False, bias=None, alpha=1.0, C_out=None, accumulate=False, splitk=1,
                  M=None, N=None, K=None, tag=0):
        """Test hook for the DMA-staged bf16 kernel: A, B are torch.bfloat16."""
        if M is None:
            M = A.shape[1] if a_col else A.shape[0]
            K = A.shape[0] if a_col else A.shape[1]
            N = B.shape[1] if b_col else B.shape[0]
        if C_out is None:
            C_out = torch.zeros(M, N, dtype=torch.float32, device=A.device)
        ws = torch.empty(max(1, splitk) * M * N, dtype=torch.float32, device=A.device) if splitk > 1 else None
        _lib.check(self.lib.recnet_gemm_bf16(_ptr(A), int(a_col), A.stride(0), _ptr(B), int(b_col), B.stride(0),
                                             _ptr(C_out), C_out.stride(0), _ptr(bias), M, N, K, float(alpha),
                                             int(accumulate), int(splitk), _ptr(ws), int(tag), _stream()),
                   "recnet_gemm_bf16")
        return C_out

    def gemm_group_bf16(self, As, Bs, a_col=False, b_col=False, biases=None, alphas=None, C_outs=None, accumulate=None, split=True):
        """Test hook for the grouped launch (recnet_gemm_group_bf16): lists of torch.bfloat16 operands of one layout; returns the
        fp32 results.  split=False: no slab workspace, i.e. no product is split along K."""
        n = len(As)
        Ms = [A.shape[1] if a_col else A.shape[0] for A in As]
        Ks = [A.shape[0] if a_col else A.shape[1] for A in As]
        Ns = [B.shape[1] if b_col else B.shape[0] for B in Bs]
        dev = As[0].device
        if C_outs is None:
            C_outs = [torch.zeros(m, nn_, dtype=torch.float32, device=dev) for m, nn_ in zip(Ms, Ns)]
        ws = torch.empty(16 * max(m * nn_ for m, nn_ in zip(Ms, Ns)) * n, dtype=torch.float32, device=dev) if split else None
        if not hasattr(self, "_gg_cnt"):
            self._gg_cnt = torch.zeros(1 << 16, dtype=torch.int32, device=dev)
        vp = lambda ts: (C.c_void_p * n)(*[C.c_void_p(t.data_ptr()) if t is not None else None for t in ts])
        ia = lambda xs: (C.c_int32 * n)(*[int(x) for x in xs])
        bl = biases if biases is not None else [None] * n
        _lib.check(self.lib.recnet_gemm_group_bf16(int(a_col), int(b_col), n, vp(As), ia([A.stride(0) for A in As]), vp(Bs),
                                                   ia([B.stride(0) for B in Bs]), vp(C_outs), ia([c.stride(0) for c in C_outs]), vp(bl),
                                                   ia(Ms), ia(Ns), ia(Ks), (C.c_float * n)(*[float(x) for x in (alphas or [1.0] * n)]),
                                                   ia(accumulate or [0] * n), _ptr(ws), ws.numel() if ws is not None else 0,
                                                   _ptr(self._gg_cnt) if split else None, (1 << 16) if split else 0, _stream()),
                   "recnet_gemm_group_bf16")
        return C_outs

    def gemm(self, A, B, a_col=False, b_col=False, bias=None, alpha=1.0, C_out=None, accumulate=False, splitk=1,
             M=None, N=None, K=None):
        """Test hook: C[M,N] (+)= alpha * op(A) op(B)^T + bias through the MFMA GEMM."""
        if M is None:
            M = A.shape[1] if a_col else A.shape[0]
            K = A.shape[0] if a_col else A.shape[1]
            N = B.shape[1] if b_col else B.shape[0]
        if C_out is None:
            C_out = torch.zeros(M, N, dtype=torch.float32, device=A.device)
        ws = torch.empty(max(1, splitk) * M * N, dtype=torch.float32, device=A.device) if splitk > 1 else None
        _lib.check(self.lib.recnet_gemm(_PREC[self.precision], _ptr(A), int(a_col), A.stride(0), _ptr(B), int(b_col),
                                        B.stride(0), _ptr(C_out), C_out.stride(0), _ptr(bias), M, N, K, float(alpha),
                                        int(accumulate), int(splitk), _ptr(ws), _stream()), "recnet_gemm")
        return C_out
