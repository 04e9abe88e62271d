import sys, time, torch, ctypes as Cc
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from recnet_amd.engine import Engine
from recnet_amd import _lib
eng = Engine(dict(B=2, F=2, D=8, E=4, H=8, A=4, V=8), None, "bf16")
def bench(fn, n=20):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n): fn()
    g.replay(); torch.cuda.synchronize()
    t0 = time.perf_counter(); g.replay(); g.replay(); torch.cuda.synchronize()
    return (time.perf_counter() - t0) / (2 * n)
shapes = [("Uv NT", 2800, 128, 1536, 0, 0), ("P NT", 2800, 2048, 1536, 0, 0), ("Xe NT", 3100, 2048, 468, 0, 0),
          ("logits NT", 3100, 4188, 512, 0, 0), ("Xg NT", 3100, 6144, 512, 0, 0), ("Xg2 NT", 3100, 6144, 1024, 0, 0), ("dhid2 NN", 3100, 1024, 6144, 0, 1), ("dHs NN", 3100, 512, 4188, 0, 1),
          ("dW_o TN", 4188, 512, 3100, 1, 1), ("dhid NN", 3100, 512, 6144, 0, 1), ("dWih TN", 6144, 512, 3100, 1, 1),
          ("dWhh_r TN", 6144, 1536, 3000, 1, 1), ("demb NN", 3100, 468, 2048, 0, 1), ("dW_e TN", 2048, 468, 3100, 1, 1),
          ("dW_c TN", 2048, 1536, 3100, 1, 1), ("dW_hh TN", 2048, 512, 3000, 1, 1), ("dW_att TN", 128, 512, 3000, 1, 1),
          ("dU TN", 128, 1536, 2800, 1, 1)]
tot = {}
for name, M, N, K, ac, bc in shapes:
    ld = lambda n: (n + 7) // 8 * 8
    A16 = torch.randn((K, ld(M)) if ac else (M, ld(K)), device="cuda").bfloat16(); B16 = torch.randn((K, ld(N)) if bc else (N, ld(K)), device="cuda").bfloat16()
    C = torch.zeros(M, N, device="cuda")
    res = []
    for sk in (1, 2, 4, 8, 16):
        if sk > (K + 63) // 64 or sk * M * N > (80 << 20): continue
        ws = torch.empty(sk * M * N, device="cuda")
        def f():
            _lib.check(eng.lib.recnet_gemm_bf16(Cc.c_void_p(A16.data_ptr()), ac, A16.stride(0), Cc.c_void_p(B16.data_ptr()), bc, B16.stride(0),
                Cc.c_void_p(C.data_ptr()), N, None, M, N, K, 1.0, 0, sk, Cc.c_void_p(ws.data_ptr()), 0, Cc.c_void_p(torch.cuda.current_stream().cuda_stream)))
        t = bench(f) * 1e6
        res.append((t, sk))
    best = min(res)
    tiles = ((M + 127) // 128) * ((N + 127) // 128)
    print("%-10s M=%5d N=%5d K=%5d tiles=%4d  " % (name, M, N, K, tiles) + "  ".join("sk%d:%6.1f" % (sk, t) for t, sk in res) + "   best sk%d %.1f us %.0f TF" % (best[1], best[0], 2.0 * M * N * K / best[0] / 1e6))
