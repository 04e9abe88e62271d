import torch, time
x = torch.zeros(64, device="cuda")
y = torch.zeros(1<<20, device="cuda")
def run(n, t): 
    for _ in range(n): t.add_(1.0)
for name, t in (("tiny", x), ("4MB", y)):
    for n in (100, 1000):
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            run(10, t)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            run(n, t)
        g.replay(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20): g.replay()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 20
        print(name, n, "kernels/graph: %.2f us per kernel (graph)" % (dt / n * 1e6))
        t0 = time.perf_counter()
        for _ in range(5): run(n, t)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 5
        print(name, n, "eager: %.2f us per kernel" % (dt / n * 1e6))
