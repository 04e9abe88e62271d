// (3100 x 6144 x 1024, row/row) with shader-clock stamps around the segments of the K loop (GL_PROBE, gemm_lds.hpp):
// wait for the DMA of this tile (s_waitcnt vmcnt), barrier, issue of the next tile's DMAs, LDS reads + MFMAs.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -I../../include -DGL_PROBE tools/micro/gemm_probe.hip -o tools/micro/gemm_probe
#include "../../reconstruction-network-for-video-captioning_amd/csrc/gemm_lds.hpp"
#include <stdio.h>
#include <string.h>
#include <stdlib.h>
#include <vector>
template <typename T> T* dalloc(size_t n) { T* p; hipMalloc(&p, n * sizeof(T)); hipMemset(p, 0x11, n * sizeof(T)); return p; }
template <typename K> static void run(const char* name, K kern, dim3 grid, int threads, int lds, GemmArgs a, int nw, int col) {
  hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipStream_t st; hipStreamCreate(&st);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e9f;
  for (int rep = 0; rep < 5; ++rep) {
    hipEventRecord(e0, st);
    hipLaunchKernelGGL(kern, grid, dim3(threads), lds, st, a);
    hipEventRecord(e1, st);
    hipStreamSynchronize(st);
    float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
  }
  std::vector<unsigned long long> b(64 * 8 * 8);
  hipMemcpyFromSymbol(b.data(), HIP_SYMBOL(gl_probe_buf), b.size() * 8);
  const int nkt = (a.K + 63) / 64;
  double s[8] = {0, 0, 0, 0, 0, 0, 0, 0}; int n = 0;
  for (int bl = 0; bl < 64; ++bl) for (int w = 0; w < nw; ++w) { for (int i = 0; i < 8; ++i) s[i] += (double)b[(bl * 8 + w) * 8 + i]; ++n; }
  printf("%-40s %7.1f us | per k-tile and wave (cycles): vmcnt wait %6.0f  barrier %6.0f  DMA issue %6.0f  ds_read+MFMA %6.0f  | per tile: loop+prologue %7.0f (%d k-tiles)  epilogue issued %6.0f  acknowledged %6.0f\n",
         name, best * 1e3, s[0] / n / nkt, s[1] / n / nkt, s[2] / n / nkt, s[3] / n / nkt, s[4] / n, nkt, s[6] / n, s[7] / n);
}
int main() {
  const int M = 3100, N = 6144, K = 1024;
  GemmArgs a; memset(&a, 0, sizeof(a));
  a.M = M; a.N = N; a.K = K; a.alpha = 1.f; a.splitk = 1; a.kchunk = K; a.a_vec = a.b_vec = 1; a.mse_B = 1;
  a.C = dalloc<float>((size_t)M * N); a.ldc = N;
  // row/row (NT): A [M][K], B [N][K]
  a.A = dalloc<bf16_t>((size_t)M * K); a.lda = K; a.B = dalloc<bf16_t>((size_t)N * K); a.ldb = K;
  run("128x128, 4 waves, 2 stages  NT", gemm_lds_kernel<false, false, 2, 0, 128>, dim3((N + 127) / 128, (M + 127) / 128, 1), 256, 2 * GL_STAGE_BYTES, a, 4, 0);
  a.c_bf16 = 1;
  run("128x128  NT, bf16 output", gemm_lds_kernel<false, false, 2, 0, 128>, dim3((N + 127) / 128, (M + 127) / 128, 1), 256, 2 * GL_STAGE_BYTES, a, 4, 0);
  a.c_bf16 = 0;
  { GemmArgs q = a; q.K = 512; q.kchunk = 512; q.N = 4188; q.ldc = 4188;
    run("128x128  NT, K = 512, N = 4188 (logits)", gemm_lds_kernel<false, false, 2, 0, 128>, dim3((4188 + 127) / 128, (M + 127) / 128, 1), 256, 2 * GL_STAGE_BYTES, q, 4, 0); }
  // col/col (TN): A [K][M], B [K][N]   (the weight-gradient form)
  a.lda = (M + 7) / 8 * 8; a.ldb = N;
  a.A = dalloc<bf16_t>((size_t)K * a.lda); a.B = dalloc<bf16_t>((size_t)K * N);
  run("128x128, 4 waves, 2 stages  TN", gemm_lds_kernel<true, true, 2, 0, 128>, dim3((N + 127) / 128, (M + 127) / 128, 1), 256, 2 * GL_STAGE_BYTES, a, 4, 1);
  return 0;
}
