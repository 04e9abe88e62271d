"""TEST INFRASTRUCTURE ONLY — numpy restatement of the product's counter-based dropout.

The reference draws dropout masks from torch's global RNG (nn.Dropout at
models/decoder.py:23,43,48,69; models/global_reconstructor.py:17,38;
models/local_reconstructor.py:25,50).  Bitwise parity with that stream is not
possible from a HIP kernel, so the product defines its masks as a pure function
of (seed, site, step, global caption index, feature index) and this file
restates that function so the oracle (and the golden-vector generator, which
substitutes the reference's nn.Dropout sub-modules) can apply identical masks.

keep(seed, site, t, b, j) = fmix32(idx * 0x9E3779B1 + key) >= floor(p * 2^32)
  idx = (t * B_global + b) * N + j   (uint32 wrap-around arithmetic)
  key = seed * 0x632BE5AB + site * 0x7F4A7C15 + 0x1234567
"""
import numpy as np

SITE_DEC_EMBED = 0   # decoder.py:48  embedding dropout, tensor [B, E] at step t
SITE_DEC_LOGIT = 1   # decoder.py:69  logits dropout,    tensor [B, V] at step t
SITE_REC_INPUT = 2   # global_reconstructor.py:38 / local_reconstructor.py:50, tensor [B, H] at step t


def _fmix32(x):
    x = x.astype(np.uint32)
    x ^= x >> np.uint32(16)
    x = (x * np.uint32(0x85EBCA6B)).astype(np.uint32)
    x ^= x >> np.uint32(13)
    x = (x * np.uint32(0xC2B2AE35)).astype(np.uint32)
    x ^= x >> np.uint32(16)
    return x


def site_key(seed, site):
    return np.uint32((int(seed) * 0x632BE5AB + int(site) * 0x7F4A7C15 + 0x1234567) & 0xFFFFFFFF)


def threshold(p):
    """drop iff hash < threshold(p)."""
    return np.uint32(min(int(float(p) * 4294967296.0), 0xFFFFFFFF))


def keep_mask(seed, site, t, B_global, N, p, b_offset=0, B_local=None):
    """float32 array [B_local, N]: 1/(1-p) where kept, 0 where dropped (p==0 -> ones)."""
    if B_local is None:
        B_local = B_global - b_offset
    if p <= 0.0:
        return np.ones((B_local, N), dtype=np.float32)
    with np.errstate(over="ignore"):
        b = (np.arange(B_local, dtype=np.uint64) + np.uint64(b_offset))[:, None]
        j = np.arange(N, dtype=np.uint64)[None, :]
        idx = ((np.uint64(t) * np.uint64(B_global) + b) * np.uint64(N) + j) & np.uint64(0xFFFFFFFF)
        x = (idx.astype(np.uint32) * np.uint32(0x9E3779B1) + site_key(seed, site)).astype(np.uint32)
        h = _fmix32(x)
    keep = h >= threshold(p)
    return keep.astype(np.float32) * np.float32(1.0 / (1.0 - p))
