// Per-step time of the persistent reconstructor chain (rec_chain.hpp) at B=100, R=1536, T=31 in a replayed graph, with
// pieces compiled out (-DRC_PROBE_*).
#include "../../reconstruction-network-for-video-captioning_amd/csrc/kernels.hpp"
#include "../../reconstruction-network-for-video-captioning_amd/csrc/rec_chain.hpp"
#include <stdio.h>
#ifndef RC_PF
#define RC_PF 2
#endif
#include <stdlib.h>
int main(int argc, char** argv) {
  const int B = 100, R = 1536, T = argc > 1 ? atoi(argv[1]) : 31;
  bf16_t *L, *W, *Pn; float *X, *C, *Hh, *acts; unsigned* bar;
  hipMalloc(&L, (size_t)T * B * R * 2); hipMalloc(&W, (size_t)4 * R * R * 2); hipMalloc(&X, (size_t)T * B * 4 * R * 4);
  hipMalloc(&C, (size_t)T * B * R * 4); hipMalloc(&Hh, (size_t)T * B * R * 4); hipMalloc(&acts, (size_t)T * B * 4 * R * 4);
  hipMalloc(&bar, 4096); hipMalloc(&Pn, (size_t)T * R * RC_PAN_ROWS * 2); hipMemset(Pn, 0, (size_t)T * R * RC_PAN_ROWS * 2);
  hipMemset(L, 0, (size_t)T * B * R * 2); hipMemset(W, 0, (size_t)4 * R * R * 2); hipMemset(X, 0, (size_t)T * B * 4 * R * 4);
  hipStream_t st; hipStreamCreate(&st);
  hipGraph_t g; hipGraphExec_t ge;
  hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
  RecChainArgs a;
  a.T = T; a.B = B; a.R = R; a.gru = 0; a.W = W; a.ldw = R; a.Hlp = L; a.ld_hlp = R; a.Pan = Pn; a.Xg = X; a.H = Hh; a.C = C; a.acts = acts; a.bar = bar; a.epoch = bar + 600; a.master = getenv("MASTER") ? atoi(getenv("MASTER")) : 1; a.poison = (float*)(bar + 700); a.hmean = nullptr; a.hmean_lp = nullptr; a.ld_hmean = R;
  hipMemsetAsync(bar, 0, 4096, st);
#ifdef RC_PROBE_MS2
  hipFuncSetAttribute(reinterpret_cast<const void*>(rec_chain_kernel<12, RC_PF, 4, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)rc_smem_bytes<4, 4>());
  const size_t sm = rc_smem_bytes<4, 4>();
  hipLaunchKernelGGL((rec_chain_kernel<12, RC_PF, 4, 4>), dim3(R / 16 + a.master, 2), dim3(256), sm, st, a);
#else
  const size_t sm = rc_smem_bytes<7, 2>();
  hipLaunchKernelGGL((rec_chain_kernel<12, RC_PF, 7, 2>), dim3(R / 8 + a.master, 1), dim3(256), sm, st, a);
#endif
  hipStreamEndCapture(st, &g); hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipGraphLaunch(ge, st); hipStreamSynchronize(st);
  hipEventRecord(e0, st); for (int r = 0; r < 5; ++r) hipGraphLaunch(ge, st); hipEventRecord(e1, st); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("persistent chain T=%d: %.1f us per launch, %.2f us per step\n", T, ms * 1e3f / 5, ms * 1e3f / (5 * T));
  return 0;
}
