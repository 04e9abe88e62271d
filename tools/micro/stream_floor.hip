// Micro-benchmark: how fast can one kernel launch of a dependent chain pull ~19 MB of weights (cache state as in the
// train step: another 40+ MB touched between launches), as a function of workgroup count and loads in flight?
// Build: hipcc -O3 --offload-arch=gfx950 stream_floor.hip -o stream_floor ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int L>   // L 16-byte loads per thread, all issued before the first use
__global__ __launch_bounds__(256) void stream_kernel(const f32x4* __restrict__ w, size_t n16, float* __restrict__ out) {
  const size_t per_wg = (size_t)256 * L;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (size_t base = (size_t)blockIdx.x * per_wg; base < n16; base += (size_t)gridDim.x * per_wg) {
    f32x4 v[L];
#pragma unroll
    for (int i = 0; i < L; ++i) {
      const size_t idx = base + (size_t)i * 256 + threadIdx.x;
      v[i] = idx < n16 ? __builtin_nontemporal_load(w + idx) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int i = 0; i < L; ++i) acc += v[i];
  }
  out[(size_t)blockIdx.x * 256 + threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
}
__global__ void touch_kernel(float* p, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] += 1.0f;
}
__global__ void empty_kernel() {}

template <int L>
float run(const f32x4* w, size_t n16, float* out, int grid, float* other, size_t nother, int chain, hipStream_t st) {
  hipGraph_t g; hipGraphExec_t ge;
  hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
  for (int i = 0; i < chain; ++i) {
    hipLaunchKernelGGL(stream_kernel<L>, dim3(grid), dim3(256), 0, st, w, n16, out);
    if (other) hipLaunchKernelGGL(touch_kernel, dim3(512), dim3(256), 0, st, other, nother);
  }
  hipStreamEndCapture(st, &g);
  hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  hipGraphLaunch(ge, st); hipStreamSynchronize(st);
  hipEventRecord(a, st);
  for (int r = 0; r < 5; ++r) hipGraphLaunch(ge, st);
  hipEventRecord(b, st); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  hipGraphExecDestroy(ge); hipGraphDestroy(g);
  return ms * 1e3f / (5 * chain);
}

int main() {
  const size_t bytes = (size_t)6144 * 1536 * 2;     // packed W_hh of the global reconstructor, bf16
  const size_t n16 = bytes / 16;
  f32x4* w; float* out; float* other;
  const size_t nother = (size_t)3 << 20;             // 12 MB read + written between launches (slabs, gates)
  hipMalloc(&w, bytes); hipMalloc(&out, 4096 * 256 * 4); hipMalloc(&other, nother * 4);
  hipMemset(w, 0, bytes); hipMemset(other, 0, nother * 4);
  hipStream_t st; hipStreamCreate(&st);
  const int chain = 30;
  // launch floor
  {
    hipGraph_t g; hipGraphExec_t ge;
    hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
    for (int i = 0; i < 60; ++i) hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(64), 0, st);
    hipStreamEndCapture(st, &g); hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipGraphLaunch(ge, st); hipStreamSynchronize(st);
    hipEventRecord(a, st); for (int r = 0; r < 5; ++r) hipGraphLaunch(ge, st); hipEventRecord(b, st); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("empty kernel in graph: %.2f us\n", ms * 1e3f / 300);
  }
  {
    float t = 0;
    hipGraph_t g; hipGraphExec_t ge;
    hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
    for (int i = 0; i < chain; ++i) hipLaunchKernelGGL(touch_kernel, dim3(512), dim3(256), 0, st, other, nother);
    hipStreamEndCapture(st, &g); hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipGraphLaunch(ge, st); hipStreamSynchronize(st);
    hipEventRecord(a, st); for (int r = 0; r < 5; ++r) hipGraphLaunch(ge, st); hipEventRecord(b, st); hipEventSynchronize(b);
    hipEventElapsedTime(&t, a, b);
    printf("touch kernel alone (12 MB r+w): %.2f us\n", t * 1e3f / (5 * chain));
  }
  const int grids[] = {96, 128, 192, 256, 384, 512, 768, 1024, 2048};
  printf("%-8s %10s %10s %10s %10s   (us per stream launch, W only | with a 12 MB touch kernel in between, pair time)\n", "grid", "L=4", "L=8", "L=16", "L=32");
  for (int gsz : grids) {
    printf("%-8d", gsz);
    printf(" %9.2f", run<4>(w, n16, out, gsz, nullptr, 0, chain, st));
    printf(" %9.2f", run<8>(w, n16, out, gsz, nullptr, 0, chain, st));
    printf(" %9.2f", run<16>(w, n16, out, gsz, nullptr, 0, chain, st));
    printf(" %9.2f", run<32>(w, n16, out, gsz, nullptr, 0, chain, st));
    printf("  |");
    printf(" %9.2f", run<8>(w, n16, out, gsz, other, nother, chain, st));
    printf(" %9.2f", run<16>(w, n16, out, gsz, other, nother, chain, st));
    printf(" %9.2f\n", run<32>(w, n16, out, gsz, other, nother, chain, st));
  }
  return 0;
}
