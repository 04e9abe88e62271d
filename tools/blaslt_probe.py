"""How fast does the vendor library (hipBLASLt through torch) run the batched GEMM shapes of the step?  bf16 x bf16 ->
bf16 (torch's output type; our kernel writes fp32), timed with graphs of 20 calls."""
import time, torch
def bench(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n): fn()
    g.replay(); torch.cuda.synchronize()
    t0 = time.perf_counter(); g.replay(); g.replay(); torch.cuda.synchronize()
    return (time.perf_counter() - t0) / (2 * n)
shapes = [("Xg  NT", 3100, 6144, 512, "nt"), ("Xg2 NT", 3100, 6144, 1024, "nt"), ("dhid2 NN", 3100, 1024, 6144, "nn"), ("P   NT", 2800, 2048, 1536, "nt"), ("logit NT", 3100, 4188, 512, "nt"), ("Xe NT", 3100, 2048, 468, "nt"),
          ("dhid NN", 3100, 512, 6144, "nn"), ("dHs NN", 3100, 512, 4188, "nn"), ("demb NN", 3100, 468, 2048, "nn"),
          ("dWih_c TN", 2048, 1536, 3100, "tn"), ("dWo TN", 4188, 512, 3100, "tn"), ("dWih_rec TN", 6144, 512, 3100, "tn"), ("dWhh_rec TN", 6144, 1536, 3000, "tn")]
for name, M, N, K, lay in shapes:
    if lay == "nt": A = torch.randn(M, K, device="cuda").bfloat16(); B = torch.randn(N, K, device="cuda").bfloat16(); f = lambda: A @ B.t()
    elif lay == "nn": A = torch.randn(M, K, device="cuda").bfloat16(); B = torch.randn(K, N, device="cuda").bfloat16(); f = lambda: A @ B
    else: A = torch.randn(K, M, device="cuda").bfloat16(); B = torch.randn(K, N, device="cuda").bfloat16(); f = lambda: A.t() @ B
    t = bench(f)
    print("%-12s M=%5d N=%5d K=%5d  %6.1f us  %6.0f TFLOP/s" % (name, M, N, K, t * 1e6, 2.0 * M * N * K / t / 1e12))
