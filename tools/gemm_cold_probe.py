"""Why do the step's batched products take 1.3-2.2 x their standalone time inside the step?  Times each shape
(a) warm: 20 back-to-back calls in a replayed graph (tools/gemm_shapes.py's number), (b) cold: every call behind a kernel that
streams 640 MB (the caches and TLBs hold none of the product's operands or output, as inside a train step: ~1.5 GB are touched
between two launches of the same product), the stream's own time subtracted; both for the library's kernel and for hipBLASLt
(torch matmul, bf16 output)."""
import sys, time, torch, ctypes as Cc
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from recnet_amd.engine import Engine
from recnet_amd import _lib
eng = Engine(dict(B=2, F=2, D=8, E=4, H=8, A=4, V=8), None, "bf16")
thrash = torch.empty(160 << 20, device="cuda")      # 640 MB of fp32
def graph_time(fns, n=10):
    for f in fns: f()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n):
            for f in fns: f()
    g.replay(); torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        t0 = time.perf_counter(); g.replay(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / n)
    return min(ts) * 1e6
fill = lambda: thrash.add_(1.0)
t_fill = graph_time([fill])
print("thrash kernel (640 MB read + write): %.1f us" % t_fill)
shapes = [("Xg2 NT", 3100, 6144, 1024, 0, 0, 1), ("dhid2 NN", 3100, 1024, 6144, 0, 1, 4), ("logits NT", 3100, 4188, 512, 0, 0, 1),
          ("dHs NN", 3100, 512, 4188, 0, 1, 4), ("dW_o TN", 4188, 512, 3100, 1, 1, 2), ("dWih TN", 6144, 1024, 3100, 1, 1, 1),
          ("dWhh_r TN", 6144, 1536, 3000, 1, 1, 1), ("P NT", 2800, 2048, 1536, 0, 0, 1), ("dW_c TN", 2048, 1536, 3100, 1, 1, 2)]
for name, M, N, K, ac, bc, sk in shapes:
    ld = lambda n: (n + 7) // 8 * 8
    A16 = torch.randn((K, ld(M)) if ac else (M, ld(K)), device="cuda").bfloat16(); B16 = torch.randn((K, ld(N)) if bc else (N, ld(K)), device="cuda").bfloat16()
    C = torch.zeros(M, N, device="cuda")
    ws = torch.empty(max(sk, 1) * M * N, device="cuda")
    def ours():
        _lib.check(eng.lib.recnet_gemm_bf16(Cc.c_void_p(A16.data_ptr()), ac, A16.stride(0), Cc.c_void_p(B16.data_ptr()), bc, B16.stride(0),
            Cc.c_void_p(C.data_ptr()), N, None, M, N, K, 1.0, 0, sk, Cc.c_void_p(ws.data_ptr()), 0, Cc.c_void_p(torch.cuda.current_stream().cuda_stream)))
    At = A16[:, :M].t() if ac else A16[:, :K]
    Bt = B16[:, :N] if bc else B16[:, :K].t()
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    def lib(): torch.matmul(At, Bt, out=out)
    w_o = graph_time([ours], 20); c_o = graph_time([fill, ours]) - t_fill
    w_l = graph_time([lib], 20); c_l = graph_time([fill, lib]) - t_fill
    fl = 2.0 * M * N * K / 1e6
    print("%-10s M=%5d N=%5d K=%5d sk%d | ours (fp32 out) warm %6.1f us (%4.0f TF) cold %6.1f | hipBLASLt (bf16 out) warm %6.1f cold %6.1f" % (
        name, M, N, K, sk, w_o, fl / w_o, c_o, w_l, c_l))
