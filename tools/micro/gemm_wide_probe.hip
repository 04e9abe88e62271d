// gemm_wide_kernel (csrc/gemm_wide.hpp) against gemm_lds_kernel (128 x 128 tiles) and hipBLASLt (fp32 output) on the step's
// large products: correctness of every (tile width, split) candidate against a plain fp32 reference kernel on random operands,
// then warm (20 back-to-back launches) and cold (each launch behind a 640 MB streaming kernel) times.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -I../../include tools/micro/gemm_wide_probe.hip -o tools/micro/gemm_wide_probe -lhipblaslt
// The vendor library is linked into this TOOL only (the yardstick); the product library does not use it.
#include "../../reconstruction-network-for-video-captioning_amd/csrc/gemm_wide.hpp"
#include <hipblaslt/hipblaslt.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <algorithm>
#define CK(x) do { auto e_ = (x); if ((int)e_ != 0) { printf("error %d at %s:%d\n", (int)e_, __FILE__, __LINE__); exit(1); } } while (0)

__global__ void fill_kernel(bf16_t* p, size_t n, unsigned seed) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const unsigned h = rn_fmix32((unsigned)i * 0x9E3779B1u + seed);
    p[i] = (bf16_t)((float)(h >> 8) * (2.0f / 16777216.0f) - 1.0f);
  }
}
__global__ void zero_pad_kernel(bf16_t* p, int rows, int ext, int ld) {      // zeros between the logical extent and ld
  const int r = blockIdx.x;
  for (int c = ext + threadIdx.x; c < ld; c += blockDim.x) p[(size_t)r * ld + c] = (bf16_t)0.f;
  (void)rows;
}
__global__ void ref_kernel(const bf16_t* A, int acol, int lda, const bf16_t* B, int bcol, int ldb, const float* bias, float* C, int M, int N, int K) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x, m = blockIdx.y;
  if (n >= N) return;
  float s = 0.f;
  for (int k = 0; k < K; ++k) {
    const float a = (float)(acol ? A[(size_t)k * lda + m] : A[(size_t)m * lda + k]);
    const float b = (float)(bcol ? B[(size_t)k * ldb + n] : B[(size_t)n * ldb + k]);
    s += a * b;
  }
  C[(size_t)m * N + n] = s + (bias ? bias[n] : 0.f);
}
__global__ void diff_kernel(const float* a, const float* b, size_t n, float* out) {      // out[0] = max |a - b|, out[1] = max |b|
  float d = 0.f, mx = 0.f;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    d = fmaxf(d, fabsf(a[i] - b[i])); mx = fmaxf(mx, fabsf(b[i]));
    if (a[i] != a[i]) d = 1e30f;
  }
  d = wave_max(d); mx = wave_max(mx);
  if ((threadIdx.x & 63) == 0) { atomicMax((unsigned*)out, __float_as_uint(d)); atomicMax((unsigned*)out + 1, __float_as_uint(mx)); }
}
__global__ void thrash_kernel(float* p, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] += 1.f;
}

struct Shape { const char* name; int M, N, K, acol, bcol, bias; };
static hipStream_t st;
static float* thrash; static const size_t thrash_n = 160u << 20;
static float t_thrash = 0.f;

template <typename F> static float time_warm(F f, int n = 20) {
  for (int i = 0; i < 3; ++i) f();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e9f;
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0, st);
    for (int i = 0; i < n; ++i) f();
    hipEventRecord(e1, st); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); best = std::min(best, ms * 1000.f / n);
  }
  return best;
}
template <typename F> static float time_cold(F f, int n = 8) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e9f;
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0, st);
    for (int i = 0; i < n; ++i) { hipLaunchKernelGGL(thrash_kernel, dim3(2048), dim3(256), 0, st, thrash, thrash_n); f(); }
    hipEventRecord(e1, st); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); best = std::min(best, ms * 1000.f / n);
  }
  return best - t_thrash;
}

struct Bufs { bf16_t *A, *B; float *C, *Cref, *bias, *ws, *dif; unsigned* cnt; int lda, ldb; size_t ws_floats; };

template <bool ACOL, bool BCOL, int NI, int WAVES>
static void launch_wide(const Shape& s, const Bufs& b, int splitk, int P) {
  typedef GwCfg<NI, BCOL, WAVES> Cf;
  static bool attr = false;
  auto fn = gemm_wide_kernel<ACOL, BCOL, NI, WAVES>;
  if (!attr) { CK(hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, Cf::LDS)); attr = true; }
  GwArgs g; memset(&g, 0, sizeof(g));
  GwProb& p = g.p[0];
  p.A = b.A; p.B = b.B; p.C = b.C; p.bias = s.bias ? b.bias : nullptr; p.C2 = nullptr; p.ws = b.ws; p.cnt = b.cnt;
  p.M = s.M; p.N = s.N; p.K = s.K; p.lda = b.lda; p.ldb = b.ldb; p.ldc = s.N; p.ldc2 = 0; p.alpha = 1.f; p.accumulate = 0; p.c_bf16 = 0;
  p.tn = (s.N + Cf::BN - 1) / Cf::BN; p.tiles = p.tn * ((s.M + 255) / 256);
  const int nkt = (s.K + 63) / 64;
  int per = (nkt + splitk - 1) / splitk; splitk = (nkt + per - 1) / per;
  p.splitk = splitk; p.kchunk = per * 64; p.first = 0; g.first[0] = 0;
  g.np = 1; g.items = p.tiles * splitk; g.stamp = nullptr;
  if (splitk > 1 && (size_t)splitk * p.tiles * Cf::BM * Cf::BN > b.ws_floats) { printf("ws too small\n"); exit(1); }
  const int grid = std::min(P, g.items);
  hipLaunchKernelGGL(fn, dim3(grid), dim3(WAVES * 64), Cf::LDS, st, g);
}
template <bool ACOL, bool BCOL, int W>
static void launch_wide_ni(int ni, const Shape& s, const Bufs& b, int splitk, int P) {
  if constexpr (!BCOL) {
    switch (ni) {
      case 2: launch_wide<ACOL, BCOL, 2, W>(s, b, splitk, P); return;
      case 3: launch_wide<ACOL, BCOL, 3, W>(s, b, splitk, P); return;
      case 4: launch_wide<ACOL, BCOL, 4, W>(s, b, splitk, P); return;
      case 5: launch_wide<ACOL, BCOL, 5, W>(s, b, splitk, P); return;
      case 6: launch_wide<ACOL, BCOL, 6, W>(s, b, splitk, P); return;
      case 8: launch_wide<ACOL, BCOL, 8, W>(s, b, splitk, P); return;
    }
  } else {
    switch (ni) {
      case 4: launch_wide<ACOL, BCOL, 4, W>(s, b, splitk, P); return;
      case 8: launch_wide<ACOL, BCOL, 8, W>(s, b, splitk, P); return;
    }
  }
  printf("no instance NI=%d\n", ni); exit(1);
}
#ifndef GW_W
#define GW_W 4
#endif
static void launch_wide_any(const Shape& s, const Bufs& b, int ni, int splitk, int P) {
  if (!s.acol && !s.bcol) launch_wide_ni<false, false, GW_W>(ni, s, b, splitk, P);
  else if (!s.acol && s.bcol) launch_wide_ni<false, true, GW_W>(ni, s, b, splitk, P);
  else launch_wide_ni<true, true, GW_W>(ni, s, b, splitk, P);
}

template <bool ACOL, bool BCOL>
static void launch_old_t(const Shape& s, const Bufs& b, int splitk) {
  static bool attr = false;
  auto fn = gemm_lds_kernel<ACOL, BCOL, 2, 0, 128>;
  if (!attr) { CK(hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * GL_STAGE_BYTES)); attr = true; }
  GemmArgs a; memset(&a, 0, sizeof(a));
  a.A = b.A; a.B = b.B; a.C = b.C; a.bias = s.bias ? b.bias : nullptr; a.M = s.M; a.N = s.N; a.K = s.K; a.lda = b.lda; a.ldb = b.ldb; a.ldc = s.N;
  a.alpha = 1.f; a.a_vec = a.b_vec = 1; a.mse_B = 1;
  const int nkt = (s.K + 63) / 64; int per = (nkt + splitk - 1) / splitk; splitk = (nkt + per - 1) / per;
  a.splitk = splitk; a.kchunk = per * 64; a.ws = b.ws;
  hipLaunchKernelGGL(fn, dim3((s.N + 127) / 128, (s.M + 127) / 128, splitk), dim3(256), 2 * GL_STAGE_BYTES, st, a);
  if (splitk > 1) {
    size_t total = (size_t)s.M * s.N; int blocks = (int)std::min<size_t>((total + 255) / 256, 2048);
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, st, b.ws, splitk, s.M, s.N, b.C, s.N, a.bias, 1.f, 0);
  }
}
static void launch_old(const Shape& s, const Bufs& b, int splitk) {
  if (!s.acol && !s.bcol) launch_old_t<false, false>(s, b, splitk);
  else if (!s.acol && s.bcol) launch_old_t<false, true>(s, b, splitk);
  else launch_old_t<true, true>(s, b, splitk);
}
static int old_splitk(int M, int N, int K) {      // gemm.hip: rn_pick_splitk, batched
  const int tiles = ((M + 127) / 128) * ((N + 127) / 128); int nkt = (K + 63) / 64, sp = 1;
  while (sp * 2 <= 16 && tiles * sp * 2 <= 512 && nkt / (sp * 2) >= 5) sp *= 2;
  return sp;
}

static double check(const Shape& s, const Bufs& b) {
  CK(hipMemsetAsync(b.dif, 0, 8, st));
  hipLaunchKernelGGL(diff_kernel, dim3(1024), dim3(256), 0, st, b.C, b.Cref, (size_t)s.M * s.N, b.dif);
  float h[2]; CK(hipMemcpyAsync(h, b.dif, 8, hipMemcpyDeviceToHost, st)); CK(hipStreamSynchronize(st));
  return h[0] / (h[1] > 0 ? h[1] : 1.f);
}

int main(int argc, char** argv) {
  const int quick = argc > 1 ? atoi(argv[1]) : 0;
  CK(hipStreamCreate(&st));
  CK(hipMalloc(&thrash, thrash_n * 4)); CK(hipMemset(thrash, 0, thrash_n * 4));
  t_thrash = time_warm([&] { hipLaunchKernelGGL(thrash_kernel, dim3(2048), dim3(256), 0, st, thrash, thrash_n); }, 8);
  printf("thrash kernel: %.1f us\n", t_thrash);
  hipblasLtHandle_t lt; CK(hipblasLtCreate(&lt));
  const size_t lt_wsz = 64u << 20; void* lt_ws; CK(hipMalloc(&lt_ws, lt_wsz));
  Shape shapes[] = {
    {"Xg2 NT+b", 3100, 6144, 1024, 0, 0, 1}, {"dhid2 NN", 3100, 1024, 6144, 0, 1, 0}, {"logits NT+b", 3100, 4188, 512, 0, 0, 1},
    {"dHs NN", 3100, 512, 4188, 0, 1, 0}, {"dW_o TN", 4188, 512, 3100, 1, 1, 0}, {"dWih TN", 6144, 1024, 3100, 1, 1, 0},
    {"dWhh_r TN", 6144, 1536, 3000, 1, 1, 0}, {"P NT", 2800, 2048, 1536, 0, 0, 0}, {"dW_c TN", 2048, 1536, 3100, 1, 1, 0},
    {"Xe NT+b", 3100, 2048, 468, 0, 0, 1}};
  Bufs b; b.ws_floats = (size_t)96 << 20;
  CK(hipMalloc(&b.ws, b.ws_floats * 4)); CK(hipMalloc(&b.cnt, 65536 * 4)); CK(hipMemset(b.cnt, 0, 65536 * 4)); CK(hipMalloc(&b.dif, 8));
  for (auto& s : shapes) {
    auto pad8 = [](int n) { return (n + 7) / 8 * 8; };
    b.lda = s.acol ? pad8(s.M) : pad8(s.K); b.ldb = s.bcol ? pad8(s.N) : pad8(s.K);
    const int ra = s.acol ? s.K : s.M, rb = s.bcol ? s.K : s.N;
    CK(hipMalloc(&b.A, (size_t)ra * b.lda * 2)); CK(hipMalloc(&b.B, (size_t)rb * b.ldb * 2));
    CK(hipMalloc(&b.C, (size_t)s.M * s.N * 4)); CK(hipMalloc(&b.Cref, (size_t)s.M * s.N * 4)); CK(hipMalloc(&b.bias, s.N * 4));
    hipLaunchKernelGGL(fill_kernel, dim3(1024), dim3(256), 0, st, b.A, (size_t)ra * b.lda, 1u);
    hipLaunchKernelGGL(fill_kernel, dim3(1024), dim3(256), 0, st, b.B, (size_t)rb * b.ldb, 2u);
    hipLaunchKernelGGL(zero_pad_kernel, dim3(ra), dim3(64), 0, st, b.A, ra, s.acol ? s.M : s.K, b.lda);
    hipLaunchKernelGGL(zero_pad_kernel, dim3(rb), dim3(64), 0, st, b.B, rb, s.bcol ? s.N : s.K, b.ldb);
    { std::vector<float> hb(s.N); for (int i = 0; i < s.N; ++i) hb[i] = 0.01f * (i % 37) - 0.2f; CK(hipMemcpy(b.bias, hb.data(), s.N * 4, hipMemcpyHostToDevice)); }
    hipLaunchKernelGGL(ref_kernel, dim3((s.N + 255) / 256, s.M), dim3(256), 0, st, b.A, s.acol, b.lda, b.B, s.bcol, b.ldb, s.bias ? b.bias : nullptr, b.Cref, s.M, s.N, s.K);
    CK(hipStreamSynchronize(st));
    const double gf = 2.0 * s.M * s.N * s.K / 1e6;
    printf("---- %-12s M=%5d N=%5d K=%5d\n", s.name, s.M, s.N, s.K);
    // old kernel
    { const int sk = old_splitk(s.M, s.N, s.K);
      CK(hipMemsetAsync(b.C, 0xff, (size_t)s.M * s.N * 4, st)); launch_old(s, b, sk); const double e = check(s, b);
      const float w = time_warm([&] { launch_old(s, b, sk); }), c = quick ? 0.f : time_cold([&] { launch_old(s, b, sk); });
      printf("  gemm_lds 128x128 sk%-2d          warm %6.1f us (%4.0f TF)  cold %6.1f   err %.1e\n", sk, w, gf / w, c, e); }
    // hipBLASLt, fp32 out
    {
      hipblasLtMatmulDesc_t d; CK(hipblasLtMatmulDescCreate(&d, HIPBLAS_COMPUTE_32F, HIP_R_32F));
      hipblasOperation_t ta = s.bcol ? HIPBLAS_OP_N : HIPBLAS_OP_T, tb = s.acol ? HIPBLAS_OP_T : HIPBLAS_OP_N;
      CK(hipblasLtMatmulDescSetAttribute(d, HIPBLASLT_MATMUL_DESC_TRANSA, &ta, sizeof(ta)));
      CK(hipblasLtMatmulDescSetAttribute(d, HIPBLASLT_MATMUL_DESC_TRANSB, &tb, sizeof(tb)));
      if (s.bias) {
        hipblasLtEpilogue_t ep = HIPBLASLT_EPILOGUE_BIAS; hipDataType bt = HIP_R_32F;
        CK(hipblasLtMatmulDescSetAttribute(d, HIPBLASLT_MATMUL_DESC_EPILOGUE, &ep, sizeof(ep)));
        CK(hipblasLtMatmulDescSetAttribute(d, HIPBLASLT_MATMUL_DESC_BIAS_POINTER, &b.bias, sizeof(b.bias)));
        CK(hipblasLtMatmulDescSetAttribute(d, HIPBLASLT_MATMUL_DESC_BIAS_DATA_TYPE, &bt, sizeof(bt)));
      }
      hipblasLtMatrixLayout_t la, lb, lc;
      CK(hipblasLtMatrixLayoutCreate(&la, HIP_R_16BF, s.bcol ? s.N : s.K, s.bcol ? s.K : s.N, b.ldb));
      CK(hipblasLtMatrixLayoutCreate(&lb, HIP_R_16BF, s.acol ? s.M : s.K, s.acol ? s.K : s.M, b.lda));
      CK(hipblasLtMatrixLayoutCreate(&lc, HIP_R_32F, s.N, s.M, s.N));
      hipblasLtMatmulPreference_t pref; CK(hipblasLtMatmulPreferenceCreate(&pref));
      CK(hipblasLtMatmulPreferenceSetAttribute(pref, HIPBLASLT_MATMUL_PREF_MAX_WORKSPACE_BYTES, &lt_wsz, sizeof(lt_wsz)));
      hipblasLtMatmulHeuristicResult_t h[2]; int nh = 0;
      CK(hipblasLtMatmulAlgoGetHeuristic(lt, d, la, lb, lc, lc, pref, 2, h, &nh));
      const float one = 1.f, zero = 0.f;
      for (int a = 0; a < nh; ++a) {
        auto f = [&] { CK(hipblasLtMatmul(lt, d, &one, b.B, la, b.A, lb, &zero, b.C, lc, b.C, lc, &h[a].algo, lt_ws, lt_wsz, st)); };
        CK(hipMemsetAsync(b.C, 0xff, (size_t)s.M * s.N * 4, st)); f(); const double e = check(s, b);
        const float w = time_warm(f), c = quick ? 0.f : time_cold(f);
        printf("  hipBLASLt algo %d (fp32 out)     warm %6.1f us (%4.0f TF)  cold %6.1f   err %.1e\n", a, w, gf / w, c, e);
      }
    }
    // wide kernel candidates
    std::vector<std::pair<int, int>> cand;
    const int nkt = (s.K + 63) / 64;
    for (int ni : {2, 3, 4, 5, 6, 8}) {
      if (s.bcol && ni != 4 && ni != 8) continue;
      const int tiles = ((s.M + 255) / 256) * ((s.N + 32 * ni - 1) / (32 * ni));
      std::vector<int> sks = {1};
      for (int sk = 2; sk <= 16; ++sk) if (tiles * sk <= 1024 && nkt / sk >= 4 && (tiles < 256 || tiles * (sk - 1) < 512)) sks.push_back(sk);
      for (int sk : sks) {
        if ((size_t)sk * tiles * 256 * 32 * ni > b.ws_floats) continue;
        // keep: unsplit, and splits that end near a whole number of rounds of 256
        const int items = tiles * sk; const double rounds = items / 256.0, fr = rounds - (int)rounds;
        if (sk > 1 && !(fr == 0.0 || fr > 0.8 || items <= 256)) continue;
        cand.push_back({ni, sk});
      }
    }
    for (auto& c : cand) {
      const int ni = c.first, sk = c.second;
      const int tiles = ((s.M + 255) / 256) * ((s.N + 32 * ni - 1) / (32 * ni));
      CK(hipMemsetAsync(b.C, 0xff, (size_t)s.M * s.N * 4, st));
      launch_wide_any(s, b, ni, sk, 256); const double e = check(s, b);
      const float w = time_warm([&] { launch_wide_any(s, b, ni, sk, 256); });
      const float cc = quick ? 0.f : time_cold([&] { launch_wide_any(s, b, ni, sk, 256); });
#ifdef GW_PROBE
      { std::vector<unsigned long long> pb(256 * 8 * 8); CK(hipMemcpyFromSymbol(pb.data(), HIP_SYMBOL(gw_probe_buf), pb.size() * 8));
        double sm[8] = {0, 0, 0, 0, 0, 0, 0, 0}; for (int w = 0; w < 256 * 8; ++w) for (int i = 0; i < 8; ++i) sm[i] += (double)pb[w * 8 + i];
        const double it = sm[6] > 0 ? sm[6] : 1, its = sm[7] > 0 ? sm[7] : 1;
        printf("      per k-tile and wave (cycles): top %5.0f  rows 0-3 %5.0f  waits %5.0f  barrier %5.0f  reads + rows 4-7 %5.0f = %5.0f | per item: epilogue %6.0f (%.0f k-tiles, %.1f items per wave)\n",
               sm[0] / it, sm[1] / it, sm[2] / it, sm[3] / it, sm[4] / it, (sm[0] + sm[1] + sm[2] + sm[3] + sm[4]) / it, sm[5] / its, it / (256 * GW_W), its / (256 * GW_W)); }
#endif
      printf("  wide%d 256x%-3d sk%-2d (%4d items)  warm %6.1f us (%4.0f TF)  cold %6.1f   err %.1e%s\n", GW_W, 32 * ni, sk, tiles * sk, w, gf / w, cc, e, e > 2e-3 ? "  <-- WRONG" : "");
    }
    hipFree(b.A); hipFree(b.B); hipFree(b.C); hipFree(b.Cref); hipFree(b.bias);
  }
  return 0;
}
