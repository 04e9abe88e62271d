import sys, time, itertools, numpy as np, torch
sys.path.insert(0, ".")
from recnet_amd.feed import DeviceFeeder
from recnet_amd.synthetic import synthetic_features, synthetic_targets
B, F, D, V = 100, 28, 1536, 4188
tg = synthetic_targets(B, V, seed=1234)
host = [(synthetic_features(B, F, D, seed=77 + i).numpy(), tg.numpy()) for i in range(3)]
for threaded in (True, False):
    fd = DeviceFeeder(itertools.cycle(host), "cuda:0", 30, threaded=threaded, depth=4, ahead=2)
    for _ in range(8): next(fd)
    torch.cuda.synchronize()
    # host time of next() alone (the GPU has nothing else to do)
    t0 = time.perf_counter()
    for _ in range(50): next(fd)
    th = (time.perf_counter() - t0) / 50 * 1e3
    torch.cuda.synchronize()
    tt = (time.perf_counter() - t0) / 50 * 1e3
    print("threaded", threaded, "host per next() %.3f ms, with device %.3f ms" % (th, tt))
# pieces
pin = torch.empty(17203200 + 25000, dtype=torch.uint8).pin_memory(); dev = torch.empty_like(pin, device="cuda")
s = torch.cuda.Stream()
torch.cuda.synchronize()
t0 = time.perf_counter()
with torch.cuda.stream(s):
    for _ in range(20): dev.copy_(pin, non_blocking=True)
th = (time.perf_counter() - t0) / 20 * 1e3
torch.cuda.synchronize()
print("H2D enqueue host time per copy %.3f ms; total %.3f ms per copy" % (th, (time.perf_counter() - t0) / 20 * 1e3))
