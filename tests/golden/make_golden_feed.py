"""Generates tests/golden/feed.npz by running the reference's own frame samplers / caption padding / collate rule
(dataset/transform.py, dataset/MSVD.py:53-74; imported from /root/reference, which exists only in the build
container) on seeded inputs.  Run:  python tests/golden/make_golden_feed.py"""
import os
import sys
sys.dont_write_bytecode = True      # the reference tree under /root/reference stays untouched (no __pycache__ beside its modules)
import types

import numpy as np
import torch

sys.path.insert(0, "/root/reference")
for name in ("h5py", "torchvision"):                      # imported by dataset/MSVD.py, unused by collate_fn
    sys.modules.setdefault(name, types.ModuleType(name))
sys.modules["torchvision"].transforms = types.SimpleNamespace(Compose=lambda fs: fs)
from dataset import transform as T                        # noqa: E402
from dataset.MSVD import MSVD                             # noqa: E402

out = {}
rs = np.random.RandomState(7)
cases = [(10, 28), (28, 28), (29, 28), (100, 28), (513, 28), (1200, 28), (64, 8)]
for ci, (n, ns) in enumerate(cases):
    frames = rs.randn(n, 6).astype(np.float32)
    out["frames_%d" % ci] = frames
    out["nsample_%d" % ci] = np.int64(ns)
    for mi, cls in enumerate((T.UniformSample, T.RandomSample, T.UniformJitterSample)):
        np.random.seed(100 + 10 * ci + mi)
        s = cls(ns)(list(frames))
        s = T.ZeroPadIfLessThan(ns)(list(s))
        out["sampled_%d_%d" % (ci, mi)] = T.ToTensor(torch.float)(s).numpy()

caps = [[5, 9, 4], [7], list(range(3, 33)), []]
for i, c in enumerate(caps):
    w = T.PadToLength(0, 31)(T.PadLast(2)(list(c)))
    out["cap_in_%d" % i] = np.asarray(c, dtype=np.int64)
    out["cap_out_%d" % i] = T.ToTensor(torch.long)(w).numpy()

ds = MSVD.__new__(MSVD)
ds.C = types.SimpleNamespace(batch_size=5)
batch = [("v%d" % i, torch.from_numpy(out["sampled_%d_0" % i]), torch.from_numpy(out["cap_out_%d" % i])) for i in range(3)]
_, vids, captions = ds.collate_fn(batch)
out["collate_videos"] = vids.numpy()
out["collate_captions"] = captions.numpy()                # float, [31, 5] (train.py casts with .long())
np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "feed.npz"), **out)
print("wrote feed.npz", len(out), "arrays")
