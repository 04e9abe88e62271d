// Where does the time of one recurrent-step GEMM launch go?  Runs the product's gemm_chain_kernel / gemm_lds_kernel on
// the reconstructor shape (M=100, N=6144, K=1536, split-K 4) inside a graph, alternating with a consumer that reads the
// slabs (like lstm_pw), with pieces compiled out (-DGC_PROBE_SKIP_*).  Build variants with tools/micro/build_probe.sh.
#include "../../reconstruction-network-for-video-captioning_amd/csrc/gemm_chain.hpp"
#include <stdio.h>
#include <stdlib.h>

__global__ __launch_bounds__(256) void consumer(const float* __restrict__ slab, int S, int n, float* __restrict__ out, bf16_t* __restrict__ hlp) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float v = 0.f;
  for (int z = 0; z < S; ++z) v += slab[(size_t)z * n + i];
  out[i] = v;
  if (i < 100 * 1536) hlp[i] = (bf16_t)(v * 1e-3f);
}

int main(int argc, char** argv) {
  const int M = 100, N = 6144, K = 1536, S = 6, chain = 30;   // 6 slices of 4 k-tiles (the chain kernel holds <= 4)
  const int which = argc > 1 ? atoi(argv[1]) : 0;     // 0 chain kernel, 1 ring kernel
  bf16_t *A, *B; float *ws, *out;
  hipMalloc(&A, (size_t)M * K * 2 * 2); hipMalloc(&B, (size_t)N * K * 2); hipMalloc(&ws, (size_t)S * M * N * 4); hipMalloc(&out, (size_t)M * N * 4);
  hipMemset(A, 0, (size_t)M * K * 2 * 2); hipMemset(B, 0, (size_t)N * K * 2);
  GemmArgs a{};
  a.A = A; a.B = B; a.C = nullptr; a.bias = nullptr; a.M = M; a.N = N; a.K = K; a.lda = K; a.ldb = K; a.ldc = N;
  a.alpha = 1.f; a.accumulate = 0; a.splitk = S; a.kchunk = K / S; a.ws = ws; a.a_vec = a.b_vec = 1; a.c_bf16 = 0; a.C2 = nullptr; a.ldc2 = 0;
  hipStream_t st; hipStreamCreate(&st);
  auto fc = gemm_chain_kernel<2, 4, 3>;
  auto fr = gemm_lds_kernel<false, false, 4, 3>;
  hipFuncSetAttribute((const void*)fc, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 16384);
  hipFuncSetAttribute((const void*)fr, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * GL_STAGE_BYTES);
  for (int mode = 0; mode < 2; ++mode) {             // 0: GEMM launches only, 1: GEMM + consumer pairs
    hipGraph_t g; hipGraphExec_t ge;
    hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
    for (int i = 0; i < chain; ++i) {
      if (which == 0) hipLaunchKernelGGL(fc, dim3(N / 128, 1, S), dim3(256), 4 * 16384, st, a);
      else hipLaunchKernelGGL(fr, dim3(N / 128, 1, S), dim3(256), 4 * GL_STAGE_BYTES, st, a);
      if (mode) hipLaunchKernelGGL(consumer, dim3((M * N + 255) / 256), dim3(256), 0, st, ws, S, M * N, out, A);
    }
    hipStreamEndCapture(st, &g); hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipGraphLaunch(ge, st); hipStreamSynchronize(st);
    hipEventRecord(e0, st); for (int r = 0; r < 5; ++r) hipGraphLaunch(ge, st); hipEventRecord(e1, st); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%s %s: %.2f us per %s\n", which ? "ring " : "chain", mode ? "gemm+consumer" : "gemm only    ", ms * 1e3f / (5 * chain), mode ? "pair" : "launch");
  }
  return 0;
}
