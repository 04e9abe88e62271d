import sys, time, torch
sys.path.insert(0, '.')
from recnet_amd.engine import Engine
eng = Engine(dict(B=2, F=2, D=8, E=4, H=8, A=4, V=8), None, "bf16")
def bench(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n): fn()
    g.replay(); torch.cuda.synchronize()
    t0 = time.perf_counter(); g.replay(); g.replay(); g.replay(); torch.cuda.synchronize()
    return (time.perf_counter() - t0) / (3 * n)
shapes = [("rec chain fwd NT", 100, 6144, 1536, 0, 0, 3, 4), ("dec chain fwd NT", 100, 2176, 512, 0, 0, 1, 8),
          ("rec chain bwd NN", 100, 1536, 6144, 0, 1, 4, 16), ("dec chain bwd NN", 100, 512, 2176, 0, 1, 2, 32),
          ("logits NT", 3100, 4188, 512, 0, 0, 0, 1), ("Xg NT", 3100, 6144, 512, 0, 0, 0, 1),
          ("dW_hh rec TN", 6144, 1536, 3000, 1, 1, 0, 1), ("dW_o TN", 4188, 512, 3100, 1, 1, 0, 1),
          ("dXg NN", 3100, 512, 6144, 0, 1, 0, 2), ("P NT", 2800, 2048, 1536, 0, 0, 0, 1)]
for name, M, N, K, ac, bc, tag, sk in shapes:
    A32 = torch.randn((K, M) if ac else (M, K), device="cuda"); B32 = torch.randn((K, N) if bc else (N, K), device="cuda")
    A16, B16 = A32.bfloat16(), B32.bfloat16()
    C = torch.zeros(M, N, device="cuda"); ws = None
    t_old = bench(lambda: eng.gemm(A32, B32, bool(ac), bool(bc), C_out=C, splitk=sk, M=M, N=N, K=K))
    t_new = bench(lambda: eng.gemm_bf16(A16, B16, bool(ac), bool(bc), C_out=C, splitk=sk, M=M, N=N, K=K, tag=tag))
    fl = 2.0 * M * N * K
    print("%-18s M=%5d N=%5d K=%5d sk=%2d  old %7.1f us (%6.1f TF)   new %7.1f us (%6.1f TF)" % (
        name, M, N, K, sk, t_old * 1e6, fl / t_old / 1e12, t_new * 1e6, fl / t_new / 1e12))
