// hipBLASLt with bf16 operands and fp32 OUTPUT on the step's whole-chip products (the library column of tools/gemm_cold_probe.py writes
// bf16): row-major C[M][N] = A . B^T (NT: A [M][K], B [N][K]) or A . B (NN: B [K][N]) or A^T . B (TN: A [K][M], B [K][N]).
#include <hip/hip_runtime.h>
#include <hipblaslt/hipblaslt.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CK(x) do { auto e_ = (x); if (e_ != 0) { printf("error %d at %s:%d\n", (int)e_, __FILE__, __LINE__); exit(1); } } while (0)
int main() {
  hipblasLtHandle_t lt; CK(hipblasLtCreate(&lt));
  struct S { const char* n; int M, N, K, acol, bcol, bias; } shapes[] = {
    {"Xg2 NT + bias", 3100, 6144, 1024, 0, 0, 1}, {"dhid2 NN", 3100, 1024, 6144, 0, 1, 0}, {"logits NT + bias", 3100, 4188, 512, 0, 0, 1},
    {"P NT", 2800, 2048, 1536, 0, 0, 0}, {"dWih TN", 6144, 1024, 3100, 1, 1, 0}, {"dWhh_r TN", 6144, 1536, 3000, 1, 1, 0}, {"dW_c TN", 2048, 1536, 3100, 1, 1, 0}};
  const size_t wsz = 64u << 20; void* ws; CK(hipMalloc(&ws, wsz));
  hipStream_t st; CK(hipStreamCreate(&st));
  for (auto& s : shapes) {
    auto pad8 = [](int n) { return (n + 7) / 8 * 8; };
    const int lda = s.acol ? pad8(s.M) : pad8(s.K), ldb = s.bcol ? pad8(s.N) : pad8(s.K), ldc = s.N;
    const size_t na = (size_t)(s.acol ? s.K : s.M) * lda, nb = (size_t)(s.bcol ? s.K : s.N) * ldb;
    void *A, *B; float *C, *bias; CK(hipMalloc(&A, na * 2)); CK(hipMalloc(&B, nb * 2)); CK(hipMalloc(&C, (size_t)s.M * ldc * 4)); CK(hipMalloc(&bias, s.N * 4));
    CK(hipMemset(A, 0, na * 2)); CK(hipMemset(B, 0, nb * 2)); CK(hipMemset(bias, 0, s.N * 4));
    // column-major view: C^T [N x M] = op(B') . op(A'),  B' = our B as a column-major matrix, A' = our A
    hipblasLtMatmulDesc_t d; CK(hipblasLtMatmulDescCreate(&d, HIPBLAS_COMPUTE_32F, HIP_R_32F));
    hipblasOperation_t ta = s.bcol ? HIPBLAS_OP_N : HIPBLAS_OP_T, tb = s.acol ? HIPBLAS_OP_T : HIPBLAS_OP_N;
    CK(hipblasLtMatmulDescSetAttribute(d, HIPBLASLT_MATMUL_DESC_TRANSA, &ta, sizeof(ta)));
    CK(hipblasLtMatmulDescSetAttribute(d, HIPBLASLT_MATMUL_DESC_TRANSB, &tb, sizeof(tb)));
    if (s.bias) {
      hipblasLtEpilogue_t ep = HIPBLASLT_EPILOGUE_BIAS; hipDataType bt = HIP_R_32F;
      CK(hipblasLtMatmulDescSetAttribute(d, HIPBLASLT_MATMUL_DESC_EPILOGUE, &ep, sizeof(ep)));
      CK(hipblasLtMatmulDescSetAttribute(d, HIPBLASLT_MATMUL_DESC_BIAS_POINTER, &bias, sizeof(bias)));
      CK(hipblasLtMatmulDescSetAttribute(d, HIPBLASLT_MATMUL_DESC_BIAS_DATA_TYPE, &bt, sizeof(bt)));
    }
    hipblasLtMatrixLayout_t la, lb, lc;
    // "A" = our B: stored column-major as [K x N] (bcol = 0: row-major [N][K]) or [N x K] (bcol = 1: row-major [K][N])
    CK(hipblasLtMatrixLayoutCreate(&la, HIP_R_16BF, s.bcol ? s.N : s.K, s.bcol ? s.K : s.N, ldb));
    CK(hipblasLtMatrixLayoutCreate(&lb, HIP_R_16BF, s.acol ? s.M : s.K, s.acol ? s.K : s.M, lda));
    CK(hipblasLtMatrixLayoutCreate(&lc, HIP_R_32F, s.N, s.M, ldc));
    hipblasLtMatmulPreference_t pref; CK(hipblasLtMatmulPreferenceCreate(&pref));
    CK(hipblasLtMatmulPreferenceSetAttribute(pref, HIPBLASLT_MATMUL_PREF_MAX_WORKSPACE_BYTES, &wsz, sizeof(wsz)));
    hipblasLtMatmulHeuristicResult_t h[4]; int nh = 0;
    CK(hipblasLtMatmulAlgoGetHeuristic(lt, d, la, lb, lc, lc, pref, 4, h, &nh));
    if (nh == 0) { printf("%-18s no algorithm\n", s.n); continue; }
    const float one = 1.f, zero = 0.f;
    for (int a = 0; a < nh && a < 2; ++a) {
      for (int i = 0; i < 3; ++i) CK(hipblasLtMatmul(lt, d, &one, B, la, A, lb, &zero, C, lc, C, lc, &h[a].algo, ws, wsz, st));
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      hipEventRecord(e0, st);
      for (int i = 0; i < 20; ++i) CK(hipblasLtMatmul(lt, d, &one, B, la, A, lb, &zero, C, lc, C, lc, &h[a].algo, ws, wsz, st));
      hipEventRecord(e1, st); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      printf("%-18s M=%5d N=%5d K=%5d  algo %d: %6.1f us  (%4.0f TF), workspace %zu\n", s.n, s.M, s.N, s.K, a, ms * 1000 / 20, 2.0 * s.M * s.N * s.K / (ms / 20 * 1e-3) / 1e12, h[a].workspaceSize);
    }
    hipFree(A); hipFree(B); hipFree(C); hipFree(bias);
  }
  return 0;
}
