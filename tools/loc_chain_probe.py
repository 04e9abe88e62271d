#!/usr/bin/env python3
"""Where a step of the local reconstructor's chain kernels goes: in-kernel wall-clock stamps (probe build of the library:
`make -C reconstruction-network-for-video-captioning_amd/csrc probe`, loaded with RN_LIB_VARIANT=probe).

   RN_LIB_VARIANT=probe python tools/loc_chain_probe.py [B F D]
Prints, per role, the median over the steps of the intervals between consecutive stamps (us) and the step period."""
import ctypes as C
import os
import sys

import numpy as np

os.environ["RN_LIB_VARIANT"] = "probe"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import recnet_amd as R  # noqa: E402
from recnet_amd.synthetic import synthetic_features, synthetic_targets  # noqa: E402

B, F, D = (int(x) for x in sys.argv[1:4]) if len(sys.argv) > 3 else (100, 28, 1536)
KIND = sys.argv[4] if len(sys.argv) > 4 else "local"
V = 4188
Cc = R.make_config(batch_size=B, use_recon=True, reconstructor_type=KIND, encoder_output_len=F, encoder_output_size=D,
                   reconstructor_hidden_size=D, precision="bf16", device="cuda")
torch.manual_seed(0)
dec, rec = R.build_decoder(V, Cc), R.build_reconstructor(Cc)
step = R.TrainStep(dec, rec)
enc = synthetic_features(B, F, D).cuda()
tg = synthetic_targets(B, V)
T, w = step.prepare(tg.numpy())
for _ in range(3):
    step.fwd_bwd(enc, tg.cuda(), T, w, seed=1)
torch.cuda.synchronize()
eng = step.engine
n = 2 * 8 * 64 * 8
buf = (C.c_uint64 * n)()
R._lib.check(eng.lib.recnet_probe_read(eng.handle, buf, n), "recnet_probe_read")
raw = np.frombuffer(buf, dtype=np.uint64).astype(np.float64) / 100.0                      # 100 MHz -> us
ts = raw[:7 * 64 * 8].reshape(7, 64, 8)
# decoder forward chain (dec_chain.hpp DC_TS, workgroup 0 = phase-A owner of 16 columns AND caption 0): [t][12]
dts = raw[4096:4096 + T * 12].reshape(T, 12)
dn = [(0, "step start (phase A)"), (1, "A: panel . W + reduction"), (4, "hand-over: own words stamped"), (8, "scores"), (9, "context"),
      (5, "cell"), (6, "publish + ack (arrive)"), (7, "barrier passed")]
print("dec fwd    period %.2f us" % np.median(np.diff(dts[3:T - 2, 0])))
prev = 0
for i, nm in dn[1:]:
    print("    %-30s +%.2f" % (nm, np.median(dts[3:T - 2, i] - dts[3:T - 2, prev])))
    prev = i
print("    %-30s +%.2f" % ("(to the next step start)", np.median(dts[4:T - 1, 0] - dts[3:T - 2, 7])))
bts = raw[4096 + 512:4096 + 512 + T * 12].reshape(T, 12)
bn = [(0, "step start (phase A')"), (1, "A': rows . W + reduction, words stamped"), (2, "hand-over + cell backward"), (3, "da = P . dgates (MFMA)"),
      (4, "attention backward (f, k) plane"), (5, "publish + ack (arrive)"), (6, "barrier passed")]
print("dec bwd    period %.2f us" % np.median(np.diff(bts[3:T - 2, 0])))
prev = 0
for i, nm in bn[1:]:
    print("    %-40s +%.2f" % (nm, np.median(bts[3:T - 2, i] - bts[3:T - 2, prev])))
    prev = i
if KIND != "local":
    sys.exit(0)
names = {0: ("fwd U", ["x arrived", "x GEMM + red", "cell", "stores issued", "ack + arrive", "hr released", "hh GEMM"]),
         1: ("fwd C", ["Pw released", "Whr summed", "beta + x", "ack + arrive"]),
         2: ("fwd relay", ["U all arrived", "C all arrived"]),
         3: ("bwd relay", ["U' arrived", "X' arrived", "C' arrived"]),
         4: ("bwd U'", ["dG released", "big GEMM", "dWhr released", "small GEMM + cell", "ack + arrive"]),
         5: ("bwd X'", ["dG released", "GEMM + red", "ack + arrive"]),
         6: ("bwd C'", ["dx released", "attention bwd", "ack + arrive"])}
if D > 2048:
    # the phased backward chain of loc_big.hpp: role 3 = workgroup 0 (L + P + C phases), role 4 = a P-only workgroup
    pts = ["step start", "L done", "barrier 1 passed", "P products done", "P partial stored", "barrier 2 passed", "C done", "barrier 3 passed"]
    for role, nm in ((3, "big bwd, workgroup 0 (L, P, C)"), (4, "big bwd, a P-only workgroup")):
        t = ts[role, 2:F - 2, :8]
        print("%-32s period %.2f us" % (nm, np.median(np.diff(ts[role, 2:F - 2, 0]))))
        for i in range(1, 8):
            print("    %-22s +%.2f" % (pts[i], np.median(t[:, i] - t[:, i - 1])))
    w5 = raw[5120:5120 + 1280].reshape(5, 256)
    print("caption workgroups 0..7 at step 10: acked / prefetch issued / x flags seen", "  ".join("%4.1f/%4.1f/%4.1f" % tuple(w5[i, c] - w5[0].min() for i in (2, 3, 4)) for c in range(8)))
    w = raw[5120:5120 + 768].reshape(3, 256)
    t0 = w[0].min()
    print("per workgroup at step 10 (us after the first release): released / products done / partial acknowledged")
    for k in range(4):
        print("  K quarter %d" % k)
        for c0 in range(0, 64, 8):
            print("    cb %2d..%2d  " % (c0, c0 + 7) + "  ".join("%4.1f/%4.1f/%4.1f" % tuple(w[i, k * 64 + c] - t0 for i in range(3)) for c in range(c0, c0 + 8)))
    del names[3], names[4], names[5], names[6]
for role, (nm, pts) in names.items():
    t = ts[role, 2:F - 2, :len(pts)]
    period = np.median(np.diff(ts[role, 2:F - 2, 0]))
    print("%-10s period %.2f us" % (nm, period))
    for i in range(1, len(pts)):
        print("    %-22s +%.2f" % (pts[i], np.median(t[:, i] - t[:, i - 1])))
    print("    %-22s +%.2f (to the next step's first stamp)" % ("...", np.median(ts[role, 3:F - 1, 0] - t[:, len(pts) - 1])))
if D > 2048:
    sys.exit(0)
t4 = ts[4, 3:F - 2]
print("bwd U': dWhr released -> partial sums in LDS %.2f, -> cells done %.2f" % (np.median(t4[:, 5] - t4[:, 2]), np.median(t4[:, 3] - t4[:, 5])))
# cross-role offsets inside a step (forward): relay 'C all arrived' -> U 'x arrived', U 'ack + arrive' -> relay 'U all arrived'
s = slice(3, F - 2)
print("fwd: relay C-arrived -> U sees x      %.2f" % np.median(ts[0, s, 0] - ts[2, s, 1]))
print("fwd: U arrive -> relay sees all U     %.2f" % np.median(ts[2, 4:F - 1, 0] - ts[0, s, 4]))
print("fwd: relay U-arrived -> C released    %.2f" % np.median(ts[1, 4:F - 1, 0] - ts[2, 4:F - 1, 0]))
print("fwd: C arrive -> relay sees all C     %.2f" % np.median(ts[2, s, 1] - ts[1, s, 3]))
print("bwd: U' arrive -> relay               %.2f" % np.median(ts[3, s, 0] - ts[4, s, 4]))
print("bwd: relay dG -> X' released          %.2f" % np.median(ts[5, s, 0] - ts[3, s, 0]))
print("bwd: X' arrive -> relay               %.2f" % np.median(ts[3, s, 1] - ts[5, s, 2]))
print("bwd: relay dx -> C' released          %.2f" % np.median(ts[6, s, 0] - ts[3, s, 1]))
print("bwd: C' arrive -> relay               %.2f" % np.median(ts[3, s, 2] - ts[6, s, 2]))
print("bwd: relay dWhr -> U' released (next) %.2f" % np.median(ts[4, 4:F - 1, 2] - ts[3, s, 2]))
