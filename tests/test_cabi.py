"""CPU-only checks of the drop-in boundary: the C-ABI shared library loads, exports every symbol that
include/recnet_hip.h declares, agrees with the ctypes mirror on struct layout, and its host-only entry
points (create / workspace_bytes / error paths) behave.  No compute call is made (no GPU here)."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "recnet_hip.h")


def _declared_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    names = re.findall(r"\b(recnet_[a-z0-9_]+)\s*\(", src)
    return sorted(set(names))


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as g
    g.build()
    from recnet_amd import _lib
    return _lib.load()


def test_every_declared_symbol_is_exported(lib):
    from recnet_amd import _lib
    declared = _declared_functions()
    assert len(declared) >= 25
    for name in declared:
        assert hasattr(lib, name), "librecnet_hip.so does not export %s" % name
    # and the ctypes mirror binds exactly the declared set
    assert sorted(_lib.EXPORTS.keys()) == declared


def test_struct_layout_matches_header(lib):
    from recnet_amd import _lib
    # 18 int32 + 8 float + 7 double
    assert C.sizeof(_lib.Config) == 19 * 4 + 8 * 4 + 4 + 7 * 8      # 27 four-byte fields, 4 bytes of alignment padding, 7 doubles
    assert C.sizeof(_lib.DecoderTensors) == 11 * 8
    assert C.sizeof(_lib.ReconstructorTensors) == 10 * 8
    hdr = open(HEADER).read()
    cfg = hdr[hdr.index("typedef struct recnet_config {"):hdr.index("} recnet_config;")]
    cfg = re.sub(r"/\*.*?\*/", "", cfg, flags=re.S)
    fields = []
    for _, names in re.findall(r"\b(int32_t|float|double)\s+([^;{]+);", cfg):
        fields += [x.strip() for x in names.split(",")]
    assert fields == [f[0] for f in _lib.Config._fields_]
    for struct, cname in ((_lib.DecoderTensors, "recnet_decoder_tensors"),
                          (_lib.ReconstructorTensors, "recnet_reconstructor_tensors")):
        body = hdr[hdr.index("typedef struct %s {" % cname):hdr.index("} %s;" % cname)]
        names = re.findall(r"float\*\s+([a-z_0-9A-Z]+);", body)
        assert names == [f[0] for f in struct._fields_]


def _cfg(**over):
    from recnet_amd import _lib
    c = _lib.Config()
    c.batch_size, c.encoder_output_len, c.encoder_output_size, c.embedding_size = 100, 28, 1536, 468
    c.decoder_hidden_size, c.decoder_attn_size, c.n_vocabs = 512, 128, 4188
    c.reconstructor_hidden_size, c.reconstructor_attn_size, c.caption_max_len = 1536, 128, 30
    c.reconstructor_type, c.precision, c.global_batch_size = _lib.REC_GLOBAL, _lib.PREC_BF16, 100
    for k, v in over.items():
        setattr(c, k, v)
    return c


def test_host_only_entry_points(lib):
    from recnet_amd import _lib
    assert lib.recnet_abi_version() == _lib.ABI_VERSION
    h = C.c_void_p()
    c = _cfg()
    assert lib.recnet_create(C.byref(c), C.byref(h)) == 0
    nbytes = lib.recnet_workspace_bytes(h)
    assert 100e6 < nbytes < 8e9          # activations of a B=100, T=31 step: hundreds of MB, far below 288 GB
    assert lib.recnet_set_shard(h, 800, 300) == 0
    assert lib.recnet_set_shard(h, 50, 0) != 0      # global batch smaller than the local one
    # calling a compute entry point before binding a workspace is a state error, not a crash
    assert lib.recnet_pack_weights(h, None) == -2
    assert b"workspace" in lib.recnet_last_error()
    lib.recnet_destroy(h)


@pytest.mark.parametrize("over,msg", [
    (dict(batch_size=0), b"dimension"),
    (dict(reconstructor_type=7), b"reconstructor_type"),
    (dict(precision=5), b"precision"),
    (dict(reconstructor_type=2, reconstructor_hidden_size=1024), b"reconstructor_hidden_size == encoder_output_size"),
    (dict(reconstructor_type=2, reconstructor_attn_size=0), b"reconstructor_attn_size"),
])
def test_bad_configs_are_rejected(lib, over, msg):
    h = C.c_void_p()
    c = _cfg(**over)
    assert lib.recnet_create(C.byref(c), C.byref(h)) == -1
    assert msg in lib.recnet_last_error()


def test_product_path_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from recnet_amd.engine import Engine
    with pytest.raises(RuntimeError, match="needs a GPU"):
        Engine(dict(B=2, F=2, D=8, E=4, H=8, A=4, V=8), None, "bf16")


def test_missing_library_is_an_error_not_a_fallback(monkeypatch, tmp_path):
    from recnet_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_lib.RecNetLibraryError, match="no CPU fallback"):
        _lib.load()
