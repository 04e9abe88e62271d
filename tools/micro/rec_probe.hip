// Time breakdown of the fused reconstructor step (rec_step.hpp) at B=100, R=1536: chain of 30 launches in a graph, with
// pieces compiled out (-DRS_PROBE_SKIP_*).
#include "../../reconstruction-network-for-video-captioning_amd/csrc/kernels.hpp"
#include "../../reconstruction-network-for-video-captioning_amd/csrc/rec_step.hpp"
#include <stdio.h>
#include <stdlib.h>
int main(int argc, char** argv) {
  const int B = 100, R = 1536, chain = 30; const int LDA = R + (argc > 1 ? atoi(argv[1]) : 0);
  bf16_t *A, *W; float *X, *C, *Hh, *acts;
  hipMalloc(&A, (size_t)2 * B * LDA * 2); hipMalloc(&W, (size_t)4 * R * R * 2); hipMalloc(&X, (size_t)B * 4 * R * 4);
  hipMalloc(&C, (size_t)2 * B * R * 4); hipMalloc(&Hh, (size_t)B * R * 4); hipMalloc(&acts, (size_t)B * 4 * R * 4);
  hipMemset(A, 0, (size_t)2 * B * LDA * 2); hipMemset(W, 0, (size_t)4 * R * R * 2); hipMemset(X, 0, (size_t)B * 4 * R * 4); hipMemset(C, 0, (size_t)2 * B * R * 4);
  hipStream_t st; hipStreamCreate(&st);
  hipGraph_t g; hipGraphExec_t ge;
  hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
  for (int i = 0; i < chain; ++i) {
    RecStepArgs a;
    a.B = B; a.R = R; a.K = R; a.A = A + (size_t)(i & 1) * B * LDA; a.lda = LDA; a.W = W; a.ldw = R; a.X = X; a.ldx = 4 * R;
    a.c_prev = C + (size_t)(i & 1) * B * R; a.h_out = Hh; a.c_out = C + (size_t)((i + 1) & 1) * B * R; a.acts = acts;
    a.h_lp = A + (size_t)((i + 1) & 1) * B * LDA; a.ld_hlp = LDA;
    hipLaunchKernelGGL((rec_step_fused_kernel<12, 3>), dim3(R / 8), dim3(256), 0, st, a);
  }
  hipStreamEndCapture(st, &g); hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipGraphLaunch(ge, st); hipStreamSynchronize(st);
  hipEventRecord(e0, st); for (int r = 0; r < 5; ++r) hipGraphLaunch(ge, st); hipEventRecord(e1, st); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("fused step: %.2f us per launch\n", ms * 1e3f / (5 * chain));
  return 0;
}
