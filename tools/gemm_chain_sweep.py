import sys, time, torch, os
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
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
shapes = [("rec fwd NT", 100, 6144, 1536, 0, 0, 3), ("dec fwd NT", 100, 2176, 512, 0, 0, 1),
          ("rec bwd NN", 100, 1536, 6144, 0, 1, 4), ("dec bwd NN", 100, 512, 2176, 0, 1, 2)]
for name, M, N, K, ac, bc, tag in shapes:
    A16 = torch.randn((K, M) if ac else (M, K), device="cuda").bfloat16(); B16 = torch.randn((K, N) if bc else (N, K), device="cuda").bfloat16()
    C = torch.zeros(M, N, device="cuda")
    res = []
    for sk in (1, 2, 4, 6, 8, 12, 16, 24, 32):
        if sk > (K + 63) // 64: continue
        ws = torch.empty(sk * M * N, device="cuda")
        import ctypes as Cc
        from recnet_amd import _lib
        def f():
            _lib.check(eng.lib.recnet_gemm_bf16(Cc.c_void_p(A16.data_ptr()), ac, A16.stride(0), Cc.c_void_p(B16.data_ptr()), bc, B16.stride(0),
                Cc.c_void_p(C.data_ptr()), N, None, M, N, K, 1.0, 0, sk, Cc.c_void_p(ws.data_ptr()), tag, Cc.c_void_p(torch.cuda.current_stream().cuda_stream)))
        res.append("sk%2d:%5.1f" % (sk, bench(f) * 1e6))
    print("NS=%s %-11s" % (os.environ.get("RN_GEMM_NS_CHAIN", "4"), name), "  ".join(res), "(us, incl. splitk_reduce launch when sk>1)")
