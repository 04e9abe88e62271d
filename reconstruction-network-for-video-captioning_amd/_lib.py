"""ctypes binding of librecnet_hip.so (C ABI: include/recnet_hip.h).

The library is the product: there is no CPU or eager-PyTorch fallback.  `load()` raises if the
shared object is missing or stale, so a GPU run can never silently take another path.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# RN_LIB_VARIANT=probe: the probe build (in-kernel time stamps); =acqinv: the cross-check build with acquire fences; =fault: fault injection
_VARIANT = ("_" + os.environ["RN_LIB_VARIANT"] if os.environ.get("RN_LIB_VARIANT") else "")
LIB_PATH = os.path.join(_HERE, "csrc", "librecnet_hip%s.so" % _VARIANT)
ABI_VERSION = 7

REC_NONE, REC_GLOBAL, REC_LOCAL = 0, 1, 2
PREC_F32, PREC_BF16 = 0, 1

_f, _d, _i = C.c_float, C.c_double, C.c_int32


class Config(C.Structure):
    """struct recnet_config"""
    _fields_ = [(n, _i) for n in (
        "batch_size", "encoder_output_len", "encoder_output_size", "embedding_size", "decoder_hidden_size",
        "decoder_attn_size", "n_vocabs", "reconstructor_hidden_size", "reconstructor_attn_size",
        "caption_max_len", "reconstructor_type", "precision", "global_batch_size", "batch_offset",
        "decoder_use_amsgrad", "reconstructor_use_amsgrad", "decoder_cell", "reconstructor_cell", "decoder_attn_normalize")] + [(n, _f) for n in (
        "embedding_scale", "embedding_dropout", "decoder_out_dropout", "reconstructor_decoder_dropout",
        "gradient_clip", "decoder_lambda_reg", "reconstructor_lambda_reg", "lambda_recon")] + [(n, _d) for n in (
        "decoder_learning_rate", "reconstructor_learning_rate", "decoder_weight_decay",
        "reconstructor_weight_decay", "adam_beta1", "adam_beta2", "adam_eps")]


DECODER_KEYS = ("attn_b", "embedding.weight", "attn_W.weight", "attn_U.weight", "attn_w.weight",
                "rnn.weight_ih_l0", "rnn.weight_hh_l0", "rnn.bias_ih_l0", "rnn.bias_hh_l0", "out.weight", "out.bias")
REC_KEYS = ("attn_b", "attn_W.weight", "attn_U.weight", "attn_w.weight", "rnn.weight_ih_l0", "rnn.weight_hh_l0",
            "rnn.bias_ih_l0", "rnn.bias_hh_l0", "out.weight", "out.bias")


class DecoderTensors(C.Structure):
    """struct recnet_decoder_tensors (field order == DECODER_KEYS)"""
    _fields_ = [(k.replace(".", "_"), C.c_void_p) for k in DECODER_KEYS]


class ReconstructorTensors(C.Structure):
    """struct recnet_reconstructor_tensors (field order == REC_KEYS)"""
    _fields_ = [(k.replace(".", "_"), C.c_void_p) for k in REC_KEYS]


SCALAR_NAMES = ("dec_ce", "dec_reg", "dec_loss", "rec_mse", "rec_reg", "rec_loss", "total_loss", "dec_grad_norm")

EXPORTS = {
    # name: (restype, argtypes)
    "recnet_abi_version": (_i, []),
    "recnet_last_error": (C.c_char_p, []),
    "recnet_create": (_i, [C.POINTER(Config), C.POINTER(C.c_void_p)]),
    "recnet_destroy": (None, [C.c_void_p]),
    "recnet_set_shard": (_i, [C.c_void_p, _i, _i]),
    "recnet_workspace_bytes": (C.c_size_t, [C.c_void_p]),
    "recnet_bind_workspace": (_i, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "recnet_bind_decoder": (_i, [C.c_void_p] + [C.POINTER(DecoderTensors)] * 5),
    "recnet_bind_reconstructor": (_i, [C.c_void_p] + [C.POINTER(ReconstructorTensors)] * 5),
    "recnet_pack_weights": (_i, [C.c_void_p, C.c_void_p]),
    "recnet_decoder_step": (_i, [C.c_void_p] + [C.c_void_p] * 7 + [_i, C.c_uint32, _i, C.c_void_p]),
    "recnet_forward_decoder": (_i, [C.c_void_p, C.c_void_p, C.c_void_p, _i, C.c_void_p, _i, C.c_uint32,
                                    C.c_void_p, C.c_void_p, C.c_void_p]),
    "recnet_forward_decoder_free": (_i, [C.c_void_p, C.c_void_p, C.c_void_p, _i, C.c_void_p, _i, C.c_uint32,
                                         C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "recnet_forward_reconstructor": (_i, [C.c_void_p, C.c_void_p, C.c_void_p, _i, _i, C.c_uint32, C.c_void_p,
                                          C.c_void_p]),
    "recnet_backward_reconstructor": (_i, [C.c_void_p, C.c_void_p, _f, C.c_void_p, C.c_void_p]),
    "recnet_backward_decoder": (_i, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, _f, C.c_void_p]),
    "recnet_add_reg_grad": (_i, [C.c_void_p, _i, _f, C.c_void_p]),
    "recnet_optimizer_step": (_i, [C.c_void_p, _i, _i, C.c_void_p, C.c_void_p]),
    "recnet_train_step_fwd_bwd": (_i, [C.c_void_p, C.c_void_p, C.c_void_p, _i, C.c_void_p, C.c_uint32,
                                       C.c_void_p, C.c_void_p]),
    "recnet_train_step": (_i, [C.c_void_p, C.c_void_p, C.c_void_p, _i, C.c_void_p, C.c_uint32, _i, C.c_void_p,
                               C.c_void_p]),
    "recnet_gemm": (_i, [_i, C.c_void_p, _i, _i, C.c_void_p, _i, _i, C.c_void_p, _i, C.c_void_p, _i, _i, _i, _f,
                         _i, _i, C.c_void_p, C.c_void_p]),
    "recnet_recurrent_step_bytes": (_d, [C.c_void_p, _i]),
    "recnet_chain_exchange_bytes": (_d, [C.c_void_p, _i]),
}

_lib = None


class RecNetLibraryError(RuntimeError):
    """librecnet_hip.so is missing, stale or does not match include/recnet_hip.h."""


class RecNetError(RuntimeError):
    """A library call returned a non-zero code; the message is recnet_last_error()'s."""


def load():
    """Load librecnet_hip.so; raise RecNetLibraryError (never fall back) when it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RecNetLibraryError(
            "HIP extension %s is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(make -C %s). There is no CPU fallback." % (LIB_PATH, os.path.dirname(LIB_PATH)))
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in EXPORTS.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise RecNetLibraryError("librecnet_hip.so does not export %s (stale build?)" % name) from e
        fn.restype = res
        fn.argtypes = args
    if lib.recnet_abi_version() != ABI_VERSION:
        raise RecNetLibraryError("librecnet_hip.so ABI version mismatch")
    _lib = lib
    return lib


def check(rc, what=""):
    if rc != 0:
        msg = load().recnet_last_error()
        raise RecNetError("%s failed (code %d): %s" % (what or "recnet call", rc, (msg or b"").decode()))

OPT_REG, OPT_CLIP, OPT_SKIP_DECODER, OPT_SKIP_RECONSTRUCTOR = 1, 2, 4, 8
EXPORTS["recnet_clip_grad_norm"] = (_i, [C.c_void_p, _i, _f, C.c_void_p, C.c_void_p])
EXPORTS["recnet_set_step"] = (_i, [C.c_void_p, _i, C.c_void_p])
EXPORTS["recnet_train_step_fwd_bwd_dev"] = (_i, [C.c_void_p, C.c_void_p, C.c_void_p, _i, C.c_void_p, C.c_uint32,
                                                 C.c_void_p, C.c_void_p])
EXPORTS["recnet_train_step_dev"] = (_i, [C.c_void_p, C.c_void_p, C.c_void_p, _i, C.c_void_p, C.c_uint32, _i, C.c_void_p, C.c_void_p])
EXPORTS["recnet_optimizer_step_dev"] = (_i, [C.c_void_p, _i, C.c_void_p, C.c_void_p])
EXPORTS["recnet_profile_begin"] = (_i, [C.c_void_p, _i])
EXPORTS["recnet_profile_end"] = (_i, [C.c_void_p, C.POINTER(_i), C.POINTER(_d)])
EXPORTS["recnet_profile_read"] = (_i, [C.c_void_p, C.POINTER(_i), C.POINTER(_d)])
EXPORTS["recnet_profile_null_launch"] = (_i, [C.c_void_p, _i, C.c_void_p])
EXPORTS["recnet_gemm_group_bf16"] = (_i, [_i, _i, _i, C.POINTER(C.c_void_p), C.POINTER(_i), C.POINTER(C.c_void_p), C.POINTER(_i),
                                         C.POINTER(C.c_void_p), C.POINTER(_i), C.POINTER(C.c_void_p), C.POINTER(_i), C.POINTER(_i),
                                         C.POINTER(_i), C.POINTER(C.c_float), C.POINTER(_i), C.c_void_p, C.c_int64, C.c_void_p, _i, C.c_void_p])
EXPORTS["recnet_set_dp_overlap"] = (_i, [C.c_void_p, _i])
EXPORTS["recnet_join_side"] = (_i, [C.c_void_p, C.c_void_p])
EXPORTS["recnet_abort_step"] = (_i, [C.c_void_p])
EXPORTS["recnet_dp_cast"] = (_i, [C.c_void_p, _i, C.c_void_p, _i, C.c_int64, C.c_void_p])
EXPORTS["recnet_dp_reduce"] = (_i, [C.c_void_p, _i, C.c_int64, C.c_void_p, _i, C.c_void_p])
EXPORTS["recnet_read_step_ring"] = (_i, [C.c_void_p, C.POINTER(C.c_uint64), C.c_void_p])
EXPORTS["recnet_debug_images_bytes"] = (C.c_int64, [C.c_void_p])
EXPORTS["recnet_debug_images_stale"] = (_i, [C.c_void_p, C.c_void_p, C.c_int64, C.POINTER(C.c_int64), C.c_void_p])
EXPORTS["recnet_read_stamps"] = (_i, [C.c_void_p, C.POINTER(C.c_uint64), _i, C.c_void_p])
EXPORTS["recnet_gemm_bf16"] = (_i, [C.c_void_p, _i, _i, C.c_void_p, _i, _i, C.c_void_p, _i, C.c_void_p, _i, _i, _i, _f, _i, _i,
                                  C.c_void_p, _i, C.c_void_p])
EXPORTS["recnet_train_step_part_dev"] = (_i, [C.c_void_p, _i, C.c_void_p, C.c_void_p, _i, C.c_void_p, C.c_uint32,
                                              C.c_void_p, C.c_void_p])
EXPORTS["recnet_decoder_prepare"] = (_i, [C.c_void_p, C.c_void_p, C.c_void_p])
EXPORTS["recnet_greedy_search"] = (_i, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p])
EXPORTS["recnet_beam_search"] = (_i, [C.c_void_p, C.c_void_p, _i, C.c_void_p, C.c_void_p, C.c_void_p])
EXPORTS["recnet_reconstructor_step"] = (_i, [C.c_void_p] * 5 + [_i] + [C.c_void_p] * 3 + [_i, C.c_uint32, _i, C.c_void_p])
EXPORTS["recnet_chain_status"] = (_i, [C.c_void_p, C.POINTER(_i), C.c_void_p])
EXPORTS["recnet_chain_reset"] = (_i, [C.c_void_p, _i, C.c_void_p])
EXPORTS["recnet_dim"] = (_i, [C.c_void_p, _i])
EXPORTS["recnet_probe_read"] = (_i, [C.c_void_p, C.c_void_p, _i])
EXPORTS["recnet_debug_poison_lds"] = (_i, [C.c_void_p, C.c_void_p])
EXPORTS["recnet_debug_raise_give_up"] = (_i, [C.c_void_p, _i, C.c_void_p])
EXPORTS["recnet_debug_occupy"] = (_i, [C.c_void_p, _i, _i, C.c_void_p])
EXPORTS["recnet_set_deferred_reconstructor_update"] = (_i, [C.c_void_p, _i, C.c_void_p])
EXPORTS["recnet_flush"] = (_i, [C.c_void_p, C.c_void_p])
EXPORTS["recnet_mark_pending"] = (_i, [C.c_void_p])
EXPORTS["recnet_debug_offset"] = (C.c_int64, [C.c_void_p, _i])
