"""Times the pieces of the host->device feature feed (pinned staging memcpy, H2D DMA) on this box."""
import time
import numpy as np
import torch

B, F, D = 100, 28, 1536
x = torch.randn(B, F, D)
xn = x.numpy()
pin = torch.empty(B, F, D).pin_memory()
dev = torch.empty(B, F, D, device="cuda")
s = torch.cuda.Stream()


def t(fn, n=20):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


mb = x.numel() * 4 / 1e6
print("bytes per batch: %.1f MB" % mb)
print("pageable->pinned copy_   %.3f ms" % t(lambda: pin.copy_(x)))
print("np.copyto into pinned    %.3f ms" % t(lambda: np.copyto(pin.numpy(), xn)))
print("pinned->device H2D       %.3f ms" % t(lambda: dev.copy_(pin, non_blocking=True)))
print("pageable->device H2D     %.3f ms" % t(lambda: dev.copy_(x)))
with torch.cuda.stream(s):
    print("pinned->device (side)    %.3f ms" % t(lambda: dev.copy_(pin, non_blocking=True)))
for th in (1, 4, 8):
    torch.set_num_threads(th)
    print("threads=%d pageable->pinned %.3f ms" % (th, t(lambda: pin.copy_(x))))
