"""Host-side AddressSanitizer pass over the C-ABI library (sanitizers run on the CPU build only; GPU
ASan is not available on this pool).  `make -C csrc asan` instruments the HOST code of api.hip — handle creation, the
workspace carve, shard bounds, the optimiser / pack-descriptor tables, every argument check — and this test drives the
host-only entry points of that build under the ASan runtime in a child process: all five BASELINE shapes x the three
model kinds x both precisions, plus ragged and over-limit shapes and the error paths.  No GPU, no compute call."""
import glob
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "reconstruction-network-for-video-captioning_amd", "csrc")

CHILD = r"""
import ctypes as C, sys
sys.path.insert(0, %r)
from recnet_amd import _lib
lib = _lib.load()
assert lib.recnet_abi_version() == _lib.ABI_VERSION
n_ok = 0
for kind in (0, 1, 2):
    for prec in (0, 1):
        for B, F, D, cml in ((8, 28, 1536, 30), (100, 28, 1536, 30), (32, 40, 2048, 30), (64, 28, 3584, 30), (7, 5, 24, 30),
                             (113, 28, 1536, 30), (256, 40, 2048, 30), (100, 130, 1536, 30), (100, 28, 1536, 90)):
            c = _lib.Config()
            c.batch_size, c.encoder_output_len, c.encoder_output_size, c.embedding_size = B, F, D, 468
            c.decoder_hidden_size, c.decoder_attn_size, c.n_vocabs = 512, 128, 4188
            c.reconstructor_hidden_size, c.reconstructor_attn_size, c.caption_max_len = D, 128, cml
            c.reconstructor_type, c.precision, c.global_batch_size = kind, prec, B
            h = C.c_void_p()
            assert lib.recnet_create(C.byref(c), C.byref(h)) == 0, lib.recnet_last_error()
            assert lib.recnet_workspace_bytes(h) > 0
            assert lib.recnet_set_shard(h, 8 * B, 3 * B) == 0
            assert lib.recnet_set_shard(h, B - 1, 0) != 0
            assert lib.recnet_pack_weights(h, None) == -2            # no workspace bound: a state error, not a crash
            assert lib.recnet_flush(h, None) == -2
            assert lib.recnet_dim(h, 0) == B and lib.recnet_dim(h, 9) == cml + 1 and lib.recnet_dim(h, 99) == 0
            assert lib.recnet_recurrent_step_bytes(h, 0) > 0
            lib.recnet_destroy(h)
            n_ok += 1
for over in (dict(batch_size=0), dict(reconstructor_type=7), dict(precision=5)):
    c = _lib.Config()
    c.batch_size, c.encoder_output_len, c.encoder_output_size, c.embedding_size = 4, 5, 24, 8
    c.decoder_hidden_size, c.decoder_attn_size, c.n_vocabs = 16, 8, 41
    c.reconstructor_hidden_size, c.reconstructor_attn_size, c.caption_max_len = 24, 8, 30
    c.reconstructor_type, c.precision, c.global_batch_size = 1, 1, 4
    for k, v in over.items():
        setattr(c, k, v)
    h = C.c_void_p()
    assert lib.recnet_create(C.byref(c), C.byref(h)) == -1
lib.recnet_destroy(None)
print("ASAN-HOST-OK", n_ok)
"""


def test_host_entry_points_under_address_sanitizer():
    env = dict(os.environ)
    env.setdefault("HIPCC", "/opt/rocm/bin/hipcc")
    subprocess.check_call(["make", "-C", CSRC, "asan", "ARCH=gfx950"], env=env, stdout=subprocess.DEVNULL)
    rts = glob.glob("/opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so")
    if not rts:
        pytest.skip("no ASan runtime in this ROCm tree")
    env.update(LD_PRELOAD=rts[0], ASAN_OPTIONS="detect_leaks=0:abort_on_error=1:halt_on_error=1", RN_LIB_VARIANT="asan")
    out = subprocess.run([sys.executable, "-c", CHILD % ROOT], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    assert "ASAN-HOST-OK 54" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]
    assert "AddressSanitizer" not in out.stderr
