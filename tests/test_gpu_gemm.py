"""MFMA GEMM (csrc/gemm.hpp) through the C ABI (recnet_gemm) against torch fp32 matmul on the same
device: every operand-layout combination, ragged sizes, split-K, bias / accumulate epilogues."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _engine(prec):
    from recnet_amd.engine import Engine
    return Engine(dict(B=2, F=2, D=8, E=4, H=8, A=4, V=8), None, prec)


def _ref(A, B, a_col, b_col):
    a = A.t() if a_col else A
    b = B.t() if b_col else B
    return a.double() @ b.double().t()


@pytest.mark.parametrize("prec", ["f32", "bf16"])
@pytest.mark.parametrize("a_col,b_col", [(0, 0), (0, 1), (1, 0), (1, 1)])
@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (100, 2048, 2048), (37, 97, 41), (300, 130, 1000), (5, 288, 112),
                                   (3100, 468, 2048)])
def test_gemm_layouts(prec, a_col, b_col, M, N, K):
    torch.manual_seed(M * 7 + N * 3 + K + a_col * 2 + b_col)
    eng = _engine(prec)
    A = torch.randn((K, M) if a_col else (M, K), device="cuda")
    B = torch.randn((K, N) if b_col else (N, K), device="cuda")
    ref = _ref(A, B, a_col, b_col)
    C = eng.gemm(A, B, bool(a_col), bool(b_col))
    torch.cuda.synchronize()
    scale = ref.abs().max().item()
    err = (C.double() - ref).abs().max().item()
    tol = (2e-5 if prec == "f32" else 2e-2) * scale
    assert err <= tol, (err, scale)


@pytest.mark.parametrize("prec", ["f32", "bf16"])
def test_gemm_exact_small_integers(prec):
    """Integer-valued operands are exact in bf16 and fp32: catches any fragment-layout slip bit-exactly,
    with an asymmetric B (guide: A=I checks need asymmetric B)."""
    eng = _engine(prec)
    M, N, K = 150, 200, 96
    g = torch.Generator().manual_seed(3)
    for a_col in (0, 1):
        for b_col in (0, 1):
            A = torch.randint(-4, 5, (K, M) if a_col else (M, K), generator=g).float().cuda()
            B = torch.randint(-4, 5, (K, N) if b_col else (N, K), generator=g).float().cuda()
            C = eng.gemm(A, B, bool(a_col), bool(b_col))
            assert torch.equal(C.double(), _ref(A, B, a_col, b_col)), (a_col, b_col)


@pytest.mark.parametrize("prec", ["f32", "bf16"])
@pytest.mark.parametrize("splitk", [1, 4, 7])
def test_gemm_epilogue_splitk_strided(prec, splitk):
    torch.manual_seed(5)
    eng = _engine(prec)
    M, N, K = 100, 200, 900
    Abig = torch.randn(M, K + 24, device="cuda")
    A = Abig[:, 8:8 + K]                       # row stride != K, 16-byte-aligned view
    B = torch.randn(N, K, device="cuda")
    bias = torch.randn(N, device="cuda")
    Cbig = torch.randn(M, N + 12, device="cuda")
    C = Cbig[:, 4:4 + N]
    ref = 0.5 * (A.double() @ B.double().t()) + bias.double() + C.double()
    eng.gemm(A, B, bias=bias, alpha=0.5, C_out=C, accumulate=True, splitk=splitk, M=M, N=N, K=K)
    torch.cuda.synchronize()
    tol = (3e-5 if prec == "f32" else 3e-2) * ref.abs().max().item()
    assert (C.double() - ref).abs().max().item() <= tol


def test_gemm_unaligned_falls_back_to_scalar_loads():
    eng = _engine("f32")
    M, N, K = 33, 45, 77            # K*4 bytes is not a multiple of 16 -> scalar path
    A = torch.randn(M, K, device="cuda")
    B = torch.randn(N, K, device="cuda")
    C = eng.gemm(A, B)
    ref = A.double() @ B.double().t()
    assert (C.double() - ref).abs().max().item() <= 2e-5 * ref.abs().max().item()


# ---------------------------------------------------------------- DMA-staged bf16 kernel (csrc/gemm_lds.hpp)
def _pad8(n):
    return (n + 7) // 8 * 8


def _bf16_operand(rows, cols, gen_int=False, g=None):
    """bf16 matrix [rows, cols] inside a zero-padded buffer whose leading dimension is a multiple of 8."""
    buf = torch.zeros(rows, _pad8(cols), dtype=torch.bfloat16, device="cuda")
    if gen_int:
        buf[:, :cols] = torch.randint(-4, 5, (rows, cols), generator=g).to(torch.bfloat16).cuda()
    else:
        buf[:, :cols] = torch.randn(rows, cols, device="cuda").to(torch.bfloat16)
    return buf, buf[:, :cols]


@pytest.mark.parametrize("tag_ns", [(0, "2"), (0, "3"), (3, "4")])
@pytest.mark.parametrize("a_col,b_col", [(0, 0), (0, 1), (1, 0), (1, 1)])
@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (100, 6144, 1536), (37, 97, 41), (300, 130, 1000), (5, 288, 112),
                                   (3100, 468, 2048), (2048, 468, 3100), (100, 2176, 512)])
def test_gemm_lds_exact_integers(tag_ns, a_col, b_col, M, N, K, monkeypatch):
    tag, ns = tag_ns
    if tag:                     # chain-site symbols exist only for A row / the site's own B layout
        if a_col or b_col:
            pytest.skip("tag 3 is the NT form")
    g = torch.Generator().manual_seed(M + N + K)
    eng = _engine("bf16")
    Ab, Av = _bf16_operand(K if a_col else M, M if a_col else K, True, g)
    Bb, Bv = _bf16_operand(K if b_col else N, N if b_col else K, True, g)
    ref = _ref(Av.float(), Bv.float(), a_col, b_col)
    for splitk in (1, 3):
        C = eng.gemm_bf16(Ab, Bb, bool(a_col), bool(b_col), splitk=splitk, M=M, N=N, K=K, tag=tag)
        torch.cuda.synchronize()
        assert torch.equal(C.double(), ref), (a_col, b_col, splitk, (C.double() - ref).abs().max().item())


@pytest.mark.parametrize("tag", [1, 3, 5])
@pytest.mark.parametrize("M,N,K,splitk", [(100, 6144, 1536, 4), (100, 2176, 512, 4), (100, 128, 1536, 16), (37, 97, 41, 1),
                                          (5, 288, 112, 1), (128, 130, 200, 1), (1, 64, 64, 1), (100, 200, 384, 2),
                                          (128, 16384, 384, 1), (77, 2048, 768, 2), (13, 40, 8, 1)])
def test_gemm_chain_exact_integers(tag, M, N, K, splitk):
    """csrc/gemm_chain.hpp (forward-form recurrent-step GEMM: weights straight into MFMA registers, one round trip):
    every K slice here is <= 6 k-tiles, so these shapes run that kernel — 64- and 128-column workgroups, partial tiles."""
    g = torch.Generator().manual_seed(M + N + K)
    eng = _engine("bf16")
    Ab, Av = _bf16_operand(M, K, True, g)
    Bb, Bv = _bf16_operand(N, K, True, g)
    ref = _ref(Av.float(), Bv.float(), 0, 0)
    C = eng.gemm_bf16(Ab, Bb, False, False, splitk=splitk, M=M, N=N, K=K, tag=tag)
    torch.cuda.synchronize()
    assert torch.equal(C.double(), ref), (C.double() - ref).abs().max().item()


@pytest.mark.parametrize("M,N,K,splitk", [(100, 1536, 6144, 16), (100, 1536, 6144, 3), (37, 96, 200, 1), (100, 192, 1000, 4),
                                          (128, 288, 64, 1), (100, 512, 2560, 8)])
def test_gemm_lds_backward_site_exact_integers(M, N, K, splitk):
    """Backward chain site (tag 4: activations x weights stored [K][N], read through the transposing LDS load)."""
    g = torch.Generator().manual_seed(M + N + K)
    eng = _engine("bf16")
    Ab, Av = _bf16_operand(M, K, True, g)
    Bb, Bv = _bf16_operand(K, N, True, g)
    ref = _ref(Av.float(), Bv.float(), 0, 1)
    C = eng.gemm_bf16(Ab, Bb, False, True, splitk=splitk, M=M, N=N, K=K, tag=4)
    torch.cuda.synchronize()
    assert torch.equal(C.double(), ref), (C.double() - ref).abs().max().item()


def test_gemm_lds_random_and_epilogue():
    torch.manual_seed(11)
    eng = _engine("bf16")
    M, N, K = 777, 1000, 1234
    Ab, Av = _bf16_operand(M, K)
    Bb, Bv = _bf16_operand(N, K)
    bias = torch.randn(N, device="cuda")
    C0 = torch.randn(M, N, device="cuda")
    C = C0.clone()
    ref = 0.25 * (Av.double() @ Bv.double().t()) + bias.double() + C0.double()
    eng.gemm_bf16(Ab, Bb, bias=bias, alpha=0.25, C_out=C, accumulate=True, M=M, N=N, K=K)
    torch.cuda.synchronize()
    assert (C.double() - ref).abs().max().item() <= 2e-4 * ref.abs().max().item()


# ---- grouped launch (csrc/gemm_lds.hpp: gemm_group_kernel): several products of one layout in one grid, split products summed
# inside the launch by their last-arriving slice
def _bf(x):
    return x.to(torch.bfloat16)


def _mk(M, N, K, a_col, b_col, g):
    A = torch.randn((K, M) if a_col else (M, K), generator=g).cuda()
    B = torch.randn((K, N) if b_col else (N, K), generator=g).cuda()
    # leading dimensions of the library's operand buffers are multiples of 8 elements (zero padded)
    def pad(t):
        c = (t.shape[1] + 7) // 8 * 8
        o = torch.zeros(t.shape[0], c, dtype=torch.bfloat16, device="cuda")
        o[:, :t.shape[1]] = _bf(t)
        return o[:, :t.shape[1]]
    return pad(A), pad(B)


GROUPS = {
    # the decoder's deferred weight gradients at the benchmark shape (train.py:264-268): dW_c, dW_ih[:, :E], dW_hh, dW_att, dU
    "tail": [(2048, 1536, 3100), (2048, 468, 3100), (2048, 512, 3000), (128, 512, 3000), (128, 1536, 2800)],
    # one long-K product alone: split by the scheduler, summed in the launch
    "split_one": [(3100, 1024, 6144)],
    # ragged members, one of them tiny, one with K below a k-tile
    "ragged": [(37, 97, 41), (300, 130, 1000), (5, 288, 112), (129, 257, 4100), (64, 8, 24)],
    "prologue": [(3100, 2048, 468), (2800, 128, 1536), (2800, 2048, 1536)],
}


@pytest.mark.parametrize("name", sorted(GROUPS))
@pytest.mark.parametrize("a_col,b_col", [(0, 0), (0, 1), (1, 1)])
@pytest.mark.parametrize("split", [True, False])
def test_gemm_group_matches_reference_and_single_launches(name, a_col, b_col, split):
    eng = _engine("bf16")
    g = torch.Generator().manual_seed(hash(name) % 1000 + 2 * a_col + b_col)
    shapes = GROUPS[name]
    ops = [_mk(M, N, K, a_col, b_col, g) for M, N, K in shapes]
    As, Bs = [o[0] for o in ops], [o[1] for o in ops]
    biases = [torch.randn(N, generator=g).cuda() if i % 2 == 0 else None for i, (M, N, K) in enumerate(shapes)]
    alphas = [1.0 if i % 3 else 0.5 for i in range(len(shapes))]
    outs = None
    for rep in range(3):           # the tile counters must come back to zero: a second and third launch see the same results
        Cs = eng.gemm_group_bf16(As, Bs, bool(a_col), bool(b_col), biases=biases, alphas=alphas, split=split)
        torch.cuda.synchronize()
        if outs is not None:
            for c0, c1 in zip(outs, Cs):
                assert torch.equal(c0, c1)
        outs = Cs
    for (M, N, K), A, B, bias, al, Cg in zip(shapes, As, Bs, biases, alphas, outs):
        ref = al * _ref(A.float(), B.float(), a_col, b_col)
        if bias is not None:
            ref = ref + bias.double()
        scale = ref.abs().max().item()
        assert (Cg.double() - ref).abs().max().item() <= 2e-5 * scale + 1e-4, (M, N, K)      # bf16 products are exact in fp32; only the summation order differs
        C1 = eng.gemm_bf16(A, B, bool(a_col), bool(b_col), bias=bias, alpha=al)
        assert (Cg - C1).abs().max().item() <= 2e-5 * scale + 1e-4


def test_gemm_group_accumulate_and_exact_integers():
    eng = _engine("bf16")
    g = torch.Generator().manual_seed(11)
    shapes = [(152, 200, 4096), (128, 128, 64), (264, 72, 2048)]      # (col operands: M and N are the leading dimensions, multiples of 8)
    As = [_bf(torch.randint(-2, 3, (K, M), generator=g).float()).cuda() for M, N, K in shapes]
    Bs = [_bf(torch.randint(-2, 3, (K, N), generator=g).float()).cuda() for M, N, K in shapes]
    C0 = [torch.randint(-5, 6, (M, N), generator=g).float().cuda() for M, N, K in shapes]
    Cs = eng.gemm_group_bf16(As, Bs, True, True, C_outs=[c.clone() for c in C0], accumulate=[1, 1, 0])
    torch.cuda.synchronize()
    for A, B, c0, c, acc in zip(As, Bs, C0, Cs, (1, 1, 0)):
        ref = A.double().t() @ B.double()
        if acc:
            ref = ref + c0.double()
        assert torch.equal(c.double(), ref)
