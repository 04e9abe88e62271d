// Phase timeline of the persistent decoder forward chain (dec_chain.hpp) at B=100, H=512, A=128, F=28, T=31:
// timestamps (100 MHz wall clock) of one workgroup, averaged over the steps.
#define DC_PROBE_TS
#ifndef DC_PROBE_WG
#define DC_PROBE_WG 0
#endif
#include "../../reconstruction-network-for-video-captioning_amd/csrc/kernels.hpp"
#include "../../reconstruction-network-for-video-captioning_amd/csrc/dec_chain.hpp"
#include <stdio.h>
#include <stdlib.h>
#include <vector>
template <typename T> T* dalloc(size_t n) { T* p; hipMalloc(&p, n * sizeof(T)); hipMemset(p, 0, n * sizeof(T)); return p; }
int main() {
  const int B = 100, H = 512, A = 128, F = 28, T = 31, N = 4 * H + A;
  DecChainArgs c;
  c.T = T; c.B = B; c.F = F; c.H = H; c.A = A; c.gru = 0;
  c.W = dalloc<bf16_t>((size_t)(4 * H + 4 * A) * H); c.ldw = H;
  c.P = dalloc<bf16_t>((size_t)B * F * 4 * H); c.ldp = 4 * H;
  c.Uv = dalloc<float>((size_t)B * F * A); c.ab = dalloc<float>(A); c.w = dalloc<float>(A);
  c.Xe = dalloc<float>((size_t)T * B * 4 * H); c.G1 = dalloc<float>((size_t)2 * T * B * N); c.epoch = dalloc<unsigned>(16); c.ll = getenv("LL") ? atoi(getenv("LL")) : 1;
  c.Pan = dalloc<bf16_t>((size_t)T * rc_pan_elems(H));
  c.Hs = dalloc<float>((size_t)T * B * H); c.Cs = dalloc<float>((size_t)T * B * H); c.acts = dalloc<float>((size_t)T * B * 4 * H);
  c.Hlp = dalloc<bf16_t>((size_t)T * B * H); c.ld_hlp = H; c.Wh = dalloc<float>((size_t)T * B * A); c.att = dalloc<float>((size_t)T * B * F);
  c.bar = dalloc<unsigned>(1024); c.poison = dalloc<float>(4); c.mp = nullptr; c.mp_scale = 1.f; c.master = getenv("MASTER") ? atoi(getenv("MASTER")) : 1; c.ts = dalloc<unsigned long long>((size_t)T * 12);
  hipStream_t st; hipStreamCreate(&st);
  const int NA = N / 16;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep) {
    
    hipEventRecord(e0, st);
    hipLaunchKernelGGL(dec_chain_kernel<false>, dim3((NA > B ? NA : B) + (c.master ? 1 : 0) - (getenv("SHORT") ? 1 : 0)), dim3(256), 0, st, c);
    hipEventRecord(e1, st);
    hipStreamSynchronize(st);
  }
  float ms; hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> ts((size_t)T * 12);
  hipMemcpy(ts.data(), c.ts, ts.size() * 8, hipMemcpyDeviceToHost);
  const int order[11] = {0, 1, 2, 3, 4, 8, 9, 5, 6, 7, 12};
  const char* name[10] = {"A: loads+MFMA+reduce", "A: G1 stores acked", "barrier 1", "B: G1 loaded", "B: scores", "B: context", "B: cell + hl",
                          "B: panel acked", "barrier 2 (+side stores)", "next step start"};
  double d[10] = {0};
  for (int t = 2; t < T - 1; ++t)
    for (int i = 0; i < 10; ++i) {
      const unsigned long long a = ts[t * 12 + order[i]], b = order[i + 1] == 12 ? ts[(t + 1) * 12] : ts[t * 12 + order[i + 1]];
      d[i] += (double)(b - a);
    }
  { float pz; hipMemcpy(&pz, c.poison, 4, hipMemcpyDeviceToHost); if (pz != pz) printf("gave up waiting: poison = NaN\n"); }
  printf("dec chain: %.1f us per launch, %.2f us per step (wg %d)\n", ms * 1e3, ms * 1e3 / T, DC_PROBE_WG);
  for (int i = 0; i < 10; ++i) printf("  %-28s %.2f us\n", name[i], d[i] / (T - 3) * 0.01);
  return 0;
}
