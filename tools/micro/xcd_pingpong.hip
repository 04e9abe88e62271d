// Hand-over latency between two workgroups through memory, same XCD vs different XCDs, by scope of the accesses.
// 256 workgroups (one per CU: 64 KB of LDS each) are launched; workgroup 0 ping-pongs a counter with workgroup `peer`
// (8 -> normally the same XCD, 1 -> the next XCD); everyone else exits.  Prints XCC ids and ns per one-way hand-over.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
template <int MODE>   // 0: agent-scope atomics (sc1: memory side)   1: workgroup-scope atomics (sc0: L2 of the XCD)
__global__ __launch_bounds__(256) void pingpong(unsigned* flag, int peer, int iters, unsigned* xcc, unsigned long long* out) {
  __shared__ char pad[60 * 1024];
  if (threadIdx.x == 0) pad[0] = 1;
  const int me = blockIdx.x;
  unsigned id;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
  if (threadIdx.x == 0) xcc[me] = id;
  if (me != 0 && me != peer) return;
  if (threadIdx.x != 0) return;
  const unsigned long long t0 = wall_clock64();
  for (int i = 0; i < iters; ++i) {
    const unsigned want = 2 * i + (me == 0 ? 0 : 1);      // wg 0 waits for even, writes odd; peer waits odd, writes even
    if (MODE == 0) {
      int spin = 0;
      while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != want) { if (++spin > 2000000) { out[1] = 1; return; } }
      __hip_atomic_store(flag, want + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      unsigned v;
      int spin = 0;
      do { asm volatile("global_load_dword %0, %1, off sc0\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(flag) : "memory"); if (++spin > 200000) { out[1] = 1; return; } } while (v != want);
      asm volatile("global_store_dword %0, %1, off sc0\n s_waitcnt vmcnt(0)" :: "v"(flag), "v"(want + 1) : "memory");
    }
  }
  if (me == 0) out[0] = wall_clock64() - t0;
}
int main() {
  unsigned *flag, *xcc; unsigned long long* out;
  hipMalloc(&flag, 256); hipMalloc(&xcc, 1024); hipMalloc(&out, 64);
  const int iters = 2000;
  for (int mode = 0; mode < 2; ++mode)
    for (int peer : {8, 1, 16, 3}) {
      hipMemset(flag, 0, 256); hipMemset(out, 0, 64);
      if (mode == 0) hipLaunchKernelGGL(pingpong<0>, dim3(256), dim3(256), 0, 0, flag, peer, iters, xcc, out);
      else hipLaunchKernelGGL(pingpong<1>, dim3(256), dim3(256), 0, 0, flag, peer, iters, xcc, out);
      if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 1; }
      unsigned hx[256]; unsigned long long t, bad;
      hipMemcpy(hx, xcc, 1024, hipMemcpyDeviceToHost); hipMemcpy(&t, out, 8, hipMemcpyDeviceToHost); hipMemcpy(&bad, out + 1, 8, hipMemcpyDeviceToHost);
      if (bad) { printf("mode %d peer %2d: xcc %u vs %u: no progress (not coherent at this scope)\n", mode, peer, hx[0] & 0xf, hx[peer] & 0xf); fflush(stdout); continue; }
      printf("mode %d (%s) peer %2d: xcc %u vs %u: %.0f ns per one-way hand-over\n", mode, mode ? "sc0 loads/stores (L2 of the XCD)" : "agent scope (sc1)",
             peer, hx[0] & 0xf, hx[peer] & 0xf, (double)t * 10.0 / (2.0 * iters)); fflush(stdout);
    }
  return 0;
}
