"""One batched-GEMM shape of the step, launched N times back to back (no graph): the target of rocprofv3 --pmc passes.
   python3 tools/gemm_one.py M N K a_col b_col splitk [ours|lib] [n]"""
import sys, torch, ctypes as Cc
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from recnet_amd.engine import Engine
from recnet_amd import _lib
M, N, K, ac, bc, sk = (int(x) for x in sys.argv[1:7])
impl = sys.argv[7] if len(sys.argv) > 7 else "ours"
n = int(sys.argv[8]) if len(sys.argv) > 8 else 20
eng = Engine(dict(B=2, F=2, D=8, E=4, H=8, A=4, V=8), None, "bf16")
ld = lambda x: (x + 7) // 8 * 8
A16 = torch.randn((K, ld(M)) if ac else (M, ld(K)), device="cuda").bfloat16(); B16 = torch.randn((K, ld(N)) if bc else (N, ld(K)), device="cuda").bfloat16()
C = torch.zeros(M, N, device="cuda"); ws = torch.empty(max(sk, 1) * M * N, device="cuda")
At = A16[:, :M].t() if ac else A16[:, :K]; Bt = B16[:, :N] if bc else B16[:, :K].t()
out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
for _ in range(n):
    if impl == "ours":
        _lib.check(eng.lib.recnet_gemm_bf16(Cc.c_void_p(A16.data_ptr()), ac, A16.stride(0), Cc.c_void_p(B16.data_ptr()), bc, B16.stride(0),
            Cc.c_void_p(C.data_ptr()), N, None, M, N, K, 1.0, 0, sk, Cc.c_void_p(ws.data_ptr()), 0, Cc.c_void_p(torch.cuda.current_stream().cuda_stream)))
    else:
        torch.matmul(At, Bt, out=out)
torch.cuda.synchronize()
