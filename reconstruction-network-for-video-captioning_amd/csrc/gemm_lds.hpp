// bf16 MFMA GEMM with direct global->LDS staging (global_load_lds_dwordx4) and an NS-deep LDS ring.
//
//   C[M,N] (+)= alpha * sum_k A(m,k) B(n,k) (+ bias[n]),  A and B stored as bf16.
//
// Same contract as gemm.hpp (row / col operand layouts, split-K slabs, fp32 or bf16 output); this is
// the production kernel of the bf16 path, gemm.hpp stays as the exact-fp32 path.  What differs:
//   * operands are already bf16 in HBM (producers write bf16 operand copies; weights are packed once per
//     optimiser step), so a tile is staged by 8 global_load_lds_dwordx4 per wave with no VGPR round trip;
//   * NS-stage ring, prefetch distance NS-1, ONE raw s_barrier per 64-deep k-tile and a counted
//     s_waitcnt vmcnt((NS-2)*8): loads of later tiles stay in flight across the barrier;
//   * LDS images are lane-linear (the DMA writes base + lane*16), bank conflicts are removed by
//     XOR-swizzling the per-lane SOURCE chunk and applying the same XOR on the fragment read:
//       row operand  [128 m][64 k]  (128-byte rows):  chunk' = chunk ^ (m & 7)            -> ds_read_b128
//       col operand  [64 k][128 m]  (256-byte rows):  chunk' = chunk ^ (((k&3)<<1)|(k&8)) -> ds_read_b64_tr_b16
// Requirements (guaranteed by construction for every bf16 buffer of the library): 16-byte aligned bases,
// leading dimensions that are multiples of 8 elements and >= the extent rounded up to 8, zero padding
// between the logical extent and its multiple of 8.
#pragma once
#include "gemm.hpp"
#include <type_traits>

#ifndef GL_TR_ASM
#define GL_TR_ASM 1      // col-operand fragments through asm transposing reads (gl_frag_tr_asm below); 0: the builtin
#endif
#define GL_STAGE_BYTES 32768   // A image 16 KiB + B image 16 KiB
#ifndef RC_ACQUIRE_INV
#define RC_ACQUIRE_INV 0       // 1: cross-check build with the conventional acquire fences (rec_chain.hpp; Makefile: acqinv)
#endif

__device__ __forceinline__ int gl_col_swz(int k) { return ((k & 3) << 1) | (k & 8); }

template <int N> __device__ __forceinline__ void gl_wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// Source address of this lane's 16 bytes of image piece `piece` (the image is lane-linear: piece * 1024 + lane * 16).
template <bool COL>
__device__ __forceinline__ const bf16_t* gl_piece_src(const bf16_t* base, int ld, int row0, int rext, int k0, int K, int piece, int lane) {
  if (!COL) {
    const int r = piece * 8 + (lane >> 3), cp = lane & 7;
    int gr = row0 + r; gr = gr < rext ? gr : rext - 1;
    int gk = k0 + ((cp ^ (r & 7)) << 3); gk = gk < K ? gk : 0;         // beyond K: any valid address, zero-fixed later
    return base + (size_t)gr * ld + gk;
  } else {
    const int kk = piece * 4 + (lane >> 4), cp = lane & 15;
    int gk = k0 + kk; gk = gk < K ? gk : K - 1;
    const int r8 = (rext + 7) & ~7;
    int gm = row0 + ((cp ^ gl_col_swz(kk)) << 3); gm = gm + 8 <= r8 ? gm : r8 - 8;
    return base + (size_t)gk * ld + gm;
  }
}
// One operand tile: NP DMA instructions per wave (4 for a 128-row tile, 3 for the 96-row weight tile of the row/row
// form), each 64 lanes x 16 B = 1 KiB of the image.
template <bool COL, int NP = 4>
__device__ __forceinline__ void gl_stage(char* img, const bf16_t* base, int ld, int row0, int rext, int k0, int K,
                                         int wave, int lane) {
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    const int piece = wave * NP + i;
    const bf16_t* src;
    if (!COL) {
      const int r = piece * 8 + (lane >> 3), cp = lane & 7;
      int gr = row0 + r; gr = gr < rext ? gr : rext - 1;
      int gk = k0 + ((cp ^ (r & 7)) << 3); gk = gk < K ? gk : 0;         // beyond K: any valid address, zero-fixed later
      src = base + (size_t)gr * ld + gk;
    } else {
      const int kk = piece * 4 + (lane >> 4), cp = lane & 15;
      int gk = k0 + kk; gk = gk < K ? gk : K - 1;
      const int r8 = (rext + 7) & ~7;
      int gm = row0 + ((cp ^ gl_col_swz(kk)) << 3); gm = gm + 8 <= r8 ? gm : r8 - 8;
      src = base + (size_t)gk * ld + gm;
    }
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)(img + piece * 1024), 16, 0, 0);
  }
}
// zero the chunks of the (last, partial) k-tile that lie beyond K
template <bool COL, int NP = 4>
__device__ __forceinline__ void gl_zero_tail(char* img, int k0, int K, int wave, int lane) {
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    const int piece = wave * NP + i;
    bool bad;
    if (!COL) {
      const int r = piece * 8 + (lane >> 3), cp = lane & 7;
      bad = k0 + ((cp ^ (r & 7)) << 3) >= K;
    } else {
      bad = k0 + piece * 4 + (lane >> 4) >= K;
    }
    if (bad) *reinterpret_cast<f32x4*>(img + piece * 1024 + lane * 16) = f32x4{0.f, 0.f, 0.f, 0.f};
  }
}

template <bool COL>
__device__ __forceinline__ bf16x8 gl_frag(const char* img, int row /*first of the 16 m-rows*/, int ks /*0 or 32*/, int lane) {
  if (!COL) {
    const int r = row + (lane & 15), c = (ks >> 3) + (lane >> 4);
    return *reinterpret_cast<const bf16x8*>(img + r * 128 + ((c ^ (r & 7)) << 4));
  } else {
    const int li = lane & 15, q = li >> 2, pp = li & 3, g = lane >> 4;
    const int kk = ks + 8 * g + q;                      // kk + 4 has the same swizzle (bits 0,1,3 unchanged)
    const int c = (row >> 3) + (pp >> 1);
    const char* a = img + kk * 256 + ((c ^ gl_col_swz(kk)) << 4) + ((pp & 1) << 3);
    typedef __attribute__((address_space(3))) bf16x4 lds_b4;
    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_b4*)(a));
    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_b4*)(a + 4 * 256));
    bf16x8 r;
    r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
    r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
    return r;
  }
}

// Col-operand fragments WITHOUT the transposing-read builtin (round 6).  hipcc's wait-count pass treats an LDS access that carries a
// memory operand as a possible reader of every LDS-DMA in flight and puts s_waitcnt vmcnt(0) in front of it; ordinary ds_read_b128
// fragment loads lose their memory operand on the way and are left alone, the ds_read_tr16_b64 builtin keeps it — so every k-tile of
// a product with a col operand (the weight-gradient form dY^T . X, and dY . W) waited for the DMAs of the NEXT stage it had just
// requested: no prefetch at all (the 1.3 - 1.6 x of the TN rows of r05_gemm_cold_vs_hipblaslt.txt).  The asm form is invisible to
// that pass; its completion is waited for explicitly (gl_tr_wait: lgkmcnt(0) with the fragment halves as operands, so no consumer
// can move above it) before the halves are put together.
struct GlTrFrag { bf16x4 lo, hi; };
__device__ __forceinline__ GlTrFrag gl_frag_tr_asm(const char* img, int row, int ks, int lane) {
  const int li = lane & 15, q = li >> 2, pp = li & 3, g = lane >> 4;
  const int kk = ks + 8 * g + q;
  const int c = (row >> 3) + (pp >> 1);
  const unsigned a = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) char*)(img + kk * 256 + ((c ^ gl_col_swz(kk)) << 4) + ((pp & 1) << 3));
  GlTrFrag f;
  asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(f.lo) : "v"(a) : "memory");
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:1024" : "=v"(f.hi) : "v"(a) : "memory");
  return f;
}
template <int N> __device__ __forceinline__ void gl_tr_wait(GlTrFrag (&f)[N]) {
  static_assert(N == 3 || N == 4, "gl_tr_wait: operand list");
  if constexpr (N == 4) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f[0].lo), "+v"(f[0].hi), "+v"(f[1].lo), "+v"(f[1].hi), "+v"(f[2].lo), "+v"(f[2].hi), "+v"(f[3].lo), "+v"(f[3].hi) :: "memory");
  else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f[0].lo), "+v"(f[0].hi), "+v"(f[1].lo), "+v"(f[1].hi), "+v"(f[2].lo), "+v"(f[2].hi) :: "memory");
}
__device__ __forceinline__ bf16x8 gl_tr_join(const GlTrFrag& f) {
  bf16x8 r;
  r[0] = f.lo[0]; r[1] = f.lo[1]; r[2] = f.lo[2]; r[3] = f.lo[3]; r[4] = f.hi[0]; r[5] = f.hi[1]; r[6] = f.hi[2]; r[7] = f.hi[3];
  return r;
}

// GL_PROBE (tools/micro/gemm_probe.hip only): shader-clock stamps around the segments of the K loop and the epilogue, per wave
#ifdef GL_PROBE
__device__ unsigned long long gl_probe_buf[64 * 8 * 8];
#define GL_PR(i) do { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); pr_acc[i] += n_ - pr_t; pr_t = n_; } while (0)
#else
#define GL_PR(i)
#endif
// BN = columns per workgroup: 128, or 96 (row/row form only) — N = 6144 with 4 K slices is 192 workgroups of 128
// columns but 256 of 96, one per CU, each with a quarter less weight data and MFMA work on the chain's critical path.
// One output tile (bx, by) of K slice z.  FIX: the launch is a grouped one (gemm_group_kernel) and a split product is finished
// inside it — see the fix-up block of the epilogue.
template <bool ACOL, bool BCOL, int NS, int BN, bool FIX, bool EPI = false, typename ArgsT>
__device__ __forceinline__ void gemm_lds_tile(const ArgsT& p, const int bx, const int by, const int z, char* gl_smem,
                                              const AdamShared* sh = nullptr) {
  static_assert(BN == 128 || (BN == 96 && !BCOL), "96-column tiles: row-layout weights only");
  constexpr int NPB = BN / 32;                 // DMA pieces per wave of the B tile = 16-column groups per wave
  constexpr int STAGE = 16384 + BN * 128;      // A image 16 KiB + B image
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = (wave >> 1) * 64, wn = (wave & 1) * (BN / 2);
  const int m0 = by * GEMM_TILE, n0 = bx * BN;
  const int kbeg = z * p.kchunk;
  int kend = kbeg + p.kchunk;
  if (kend > p.K) kend = p.K;
  const bf16_t* A = reinterpret_cast<const bf16_t*>(p.A);
  const bf16_t* B = reinterpret_cast<const bf16_t*>(p.B);
  constexpr int D = NS - 1;     // prefetch distance in k-tiles

  f32x4 acc[4][NPB];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < NPB; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nkt = (kend > kbeg) ? (kend - kbeg + 63) / 64 : 0;
#ifdef GL_PROBE
  unsigned long long pr_acc[6] = {0, 0, 0, 0, 0, 0}, pr_t = __builtin_amdgcn_s_memtime();
  const unsigned long long pr_t0 = pr_t;
#endif
#pragma unroll
  for (int d = 0; d < D; ++d)
    if (d < nkt) {
      char* st = gl_smem + d * STAGE;
      gl_stage<ACOL>(st, A, p.lda, m0, p.M, kbeg + d * 64, kend, wave, lane);
      gl_stage<BCOL, NPB>(st + 16384, B, p.ldb, n0, p.N, kbeg + d * 64, kend, wave, lane);
    }
  for (int kt = 0; kt < nkt; ++kt) {
    // this wave's DMA of tile kt has landed once at most `ahead` later tiles (8 DMAs each) are outstanding
    const int issued = (kt + D < nkt) ? kt + D : nkt;
    const int ahead = issued - (kt + 1);
    if (NS >= 4 && ahead >= 2) gl_wait_vmcnt<2 * (4 + NPB)>();
    else if (NS >= 3 && ahead >= 1) gl_wait_vmcnt<4 + NPB>();
    else gl_wait_vmcnt<0>();
    GL_PR(0);
    __builtin_amdgcn_s_barrier();           // every wave's part of tile kt landed; everyone is done with tile kt-1
    asm volatile("" ::: "memory");
    GL_PR(1);
    if (kt + D < nkt) {
      char* st = gl_smem + ((kt + D) % NS) * STAGE;
      gl_stage<ACOL>(st, A, p.lda, m0, p.M, kbeg + (kt + D) * 64, kend, wave, lane);
      gl_stage<BCOL, NPB>(st + 16384, B, p.ldb, n0, p.N, kbeg + (kt + D) * 64, kend, wave, lane);
    }
    char* cur = gl_smem + (kt % NS) * STAGE;
    const int k0 = kbeg + kt * 64;
    GL_PR(2);
    if (k0 + 64 > kend) {                    // partial last tile: zero what lies beyond K (block-uniform branch)
      gl_zero_tail<ACOL>(cur, k0, kend, wave, lane);
      gl_zero_tail<BCOL, NPB>(cur + 16384, k0, kend, wave, lane);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
    }
#pragma unroll
    for (int ks = 0; ks < 64; ks += 32) {
      bf16x8 fa[4], fb[NPB];
#if GL_TR_ASM
      GlTrFrag ta[4], tb[NPB];
      if constexpr (ACOL) {
#pragma unroll
        for (int i = 0; i < 4; ++i) ta[i] = gl_frag_tr_asm(cur, wm + i * 16, ks, lane);
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) fa[i] = gl_frag<false>(cur, wm + i * 16, ks, lane);
      }
      if constexpr (BCOL) {
#pragma unroll
        for (int j = 0; j < NPB; ++j) tb[j] = gl_frag_tr_asm(cur + 16384, wn + j * 16, ks, lane);
      } else {
#pragma unroll
        for (int j = 0; j < NPB; ++j) fb[j] = gl_frag<false>(cur + 16384, wn + j * 16, ks, lane);
      }
      if constexpr (ACOL) { gl_tr_wait(ta);
#pragma unroll
        for (int i = 0; i < 4; ++i) fa[i] = gl_tr_join(ta[i]); }
      if constexpr (BCOL) { gl_tr_wait(tb);
#pragma unroll
        for (int j = 0; j < NPB; ++j) fb[j] = gl_tr_join(tb[j]); }
#else
#pragma unroll
      for (int i = 0; i < 4; ++i) fa[i] = gl_frag<ACOL>(cur, wm + i * 16, ks, lane);
#pragma unroll
      for (int j = 0; j < NPB; ++j) fb[j] = gl_frag<BCOL>(cur + 16384, wn + j * 16, ks, lane);
#endif
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < NPB; ++j) acc[i][j] = gemm_mma(fa[i], fb[j], acc[i][j]);
    }
    GL_PR(3);
  }
#ifdef GL_PROBE
  {
    const int bl = by * gridDim.x + bx;
    if (bl < 64 && lane == 0 && z == 0) {
      pr_acc[4] = pr_t - pr_t0;      // prologue + K loop
      for (int i = 0; i < 5; ++i) gl_probe_buf[(bl * 8 + wave) * 8 + i] = pr_acc[i];
    }
  }
#endif

  // ---- epilogue.  The MFMA C/D fragment is 4 rows x 16 columns per register, i.e. 64-byte row segments per
  // store instruction; going through a wave-private LDS staging block turns the wave's 64x64 tile into whole
  // 256-byte row segments written with one 16-byte store per lane (4x fewer store instructions).
  const int cr = (lane >> 4) * 4, cc = lane & 15;
  __builtin_amdgcn_s_barrier();                 // every wave has finished reading the operand ring
  asm volatile("" ::: "memory");
  float* stg = reinterpret_cast<float*>(gl_smem) + wave * (32 * 68);
  const bool to_slab = p.splitk > 1;
  float* Cb = to_slab ? p.ws + (size_t)z * p.M * p.N : p.C;
  const int ldc = to_slab ? p.N : p.ldc;
  const bool vec4 = ((ldc & 3) == 0) && ((((uintptr_t)Cb) & 15) == 0) && !p.c_bf16 &&
                    (!p.C2 || (((p.ldc2 & 3) == 0) && ((((uintptr_t)p.C2) & 7) == 0)));
  if (FIX && to_slab && p.cnt) {
    // ---- split product finished inside the launch.  Every slice writes its fp32 partial tile THROUGH to memory (sc1 16-byte
    // stores: no release fence, nothing else of the L2 is written back), waits for the stores, and takes a ticket on the
    // tile's counter; the slice that draws the last ticket reads all slabs back in slice order (sc1 loads: served by L2 /
    // memory, never by this CU's L1) and runs the epilogue.  Slab addresses are written once and read once per launch, and
    // the kernel boundary in front of the launch has dropped whatever an L2 held of them — the same argument as for the
    // exchange panels of the chain kernels (rec_chain.hpp); RC_ACQUIRE_INV builds add the conventional acquire fence.
    // The sum runs in slice order whoever arrives last: results do not depend on the schedule.  The host splits only products
    // with N % 4 == 0 (every slab access is a whole 16-byte quad).
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)p.ws, 0, 0x7ffffffc, 0x00020000);
    const unsigned slab_b = (unsigned)p.M * (unsigned)p.N * 4u;          // bytes of one slab (the host keeps splitk * slab_b < 2 GiB)
#pragma unroll
    for (int half = 0; half < 2; ++half) {
#pragma unroll
      for (int ii = 0; ii < 2; ++ii)
#pragma unroll
        for (int j = 0; j < NPB; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r) stg[(ii * 16 + cr + r) * 68 + j * 16 + cc] = acc[half * 2 + ii][j][r];
#pragma unroll
      for (int it = 0; it < 8; ++it) {
        const int rl = it * 4 + (lane >> 4), c4 = (lane & 15) * 4;
        const f32x4 v = *reinterpret_cast<const f32x4*>(stg + rl * 68 + c4);
        const int row = m0 + wm + half * 32 + rl, col = n0 + wn + c4;
        if (row >= p.M || col >= p.N || c4 >= BN / 2) continue;
        const unsigned off = (unsigned)z * slab_b + ((unsigned)row * (unsigned)p.N + (unsigned)col) * 4u;
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4_t, v), rs, off, 0, 16);      // (N % 4 == 0: whole quads)
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    unsigned* flag = reinterpret_cast<unsigned*>(gl_smem + 4 * 32 * 68 * 4);      // behind the four staging blocks
    unsigned* cn = p.cnt + (size_t)by * ((p.N + BN - 1) / BN) + bx;
    if (tid == 0) *flag = __hip_atomic_fetch_add(cn, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (*flag != (unsigned)(p.splitk - 1)) return;
    if (tid == 0) __hip_atomic_store(cn, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // ready for the next launch
#if RC_ACQUIRE_INV
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
#endif
    const int S = p.splitk;
    const int nrow = (p.M - m0 < GEMM_TILE) ? p.M - m0 : GEMM_TILE;
    const int ncol = (p.N - n0 < BN) ? p.N - n0 : BN;
    // 16 quads per thread and slab; 16 loads of 16 bytes in flight per thread at a time (NBQ quads x ZS slices): the sum is
    // bound by round trips, not by bytes — one quad at a time took 16 dependent trips per tile
    auto sum_tile = [&](auto nbq_c, auto zs_c) {
      constexpr int NBQ = decltype(nbq_c)::value, ZS = decltype(zs_c)::value;
      for (int q0 = 0; q0 < GEMM_TILE * (BN / 4) / 256; q0 += NBQ) {
        f32x4 part[NBQ * ZS];
        unsigned off[NBQ]; bool ok[NBQ];
#pragma unroll
        for (int u = 0; u < NBQ; ++u) {
          const int q = (q0 + u) * 256 + tid;
          const int rl = q / (BN / 4), c4 = (q - rl * (BN / 4)) * 4;
          ok[u] = rl < nrow && c4 < ncol;
          off[u] = ((unsigned)(m0 + rl) * (unsigned)p.N + (unsigned)(n0 + c4)) * 4u;
#pragma unroll
          for (int zz = 0; zz < ZS; ++zz)
            if (zz < S && ok[u]) part[u * ZS + zz] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, off[u] + (unsigned)zz * slab_b, 0, 16));
        }
#pragma unroll
        for (int u = 0; u < NBQ; ++u) {
          if (!ok[u]) continue;
          f32x4 sum = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int zz = 0; zz < ZS; ++zz) if (zz < S) sum += part[u * ZS + zz];
          const int q = (q0 + u) * 256 + tid;
          const int rl = q / (BN / 4), c4 = (q - rl * (BN / 4)) * 4;
          const int row = m0 + rl, col = n0 + c4;
          float o[4];
          if (p.bias && ((((uintptr_t)p.bias) & 15) == 0)) {      // (N % 4 == 0 here: the quad is whole)
            const f32x4 b4 = *reinterpret_cast<const f32x4*>(p.bias + col);
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = p.alpha * sum[e] + b4[e];
          } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = p.alpha * sum[e] + ((p.bias && col + e < p.N) ? p.bias[col + e] : 0.f);
          }
          if (p.c_bf16) {
            bf16_t* dst = reinterpret_cast<bf16_t*>(p.C) + (size_t)row * p.ldc + col;
            if (((p.ldc & 3) == 0) && ((((uintptr_t)p.C) & 7) == 0) && col + 3 < p.N) {
              bf16x4 hb; hb[0] = (bf16_t)o[0]; hb[1] = (bf16_t)o[1]; hb[2] = (bf16_t)o[2]; hb[3] = (bf16_t)o[3];
              *reinterpret_cast<bf16x4*>(dst) = hb;
            } else
            for (int e = 0; e < 4 && col + e < p.N; ++e) dst[e] = (bf16_t)o[e];
          } else {
            float* dst = p.C + (size_t)row * p.ldc + col;
            if (((p.ldc & 3) == 0) && ((((uintptr_t)p.C) & 15) == 0) && !p.C2) {      // (N % 4 == 0: the quad is whole)
              f32x4 w = f32x4{o[0], o[1], o[2], o[3]};
              if (p.accumulate) w += *reinterpret_cast<const f32x4*>(dst);
              *reinterpret_cast<f32x4*>(dst) = w;
            } else {
              for (int e = 0; e < 4 && col + e < p.N; ++e) {
                float w = o[e];
                if (p.accumulate) w += dst[e];
                dst[e] = w;
                if (p.C2) reinterpret_cast<bf16_t*>(p.C2)[(size_t)row * p.ldc2 + col + e] = (bf16_t)w;
              }
            }
          }
        }
      }
    };
    if (S <= 2) sum_tile(std::integral_constant<int, 8>(), std::integral_constant<int, 2>());
    else if (S <= 4) sum_tile(std::integral_constant<int, 4>(), std::integral_constant<int, 4>());
    else if (S <= 8) sum_tile(std::integral_constant<int, 2>(), std::integral_constant<int, 8>());
    else sum_tile(std::integral_constant<int, 1>(), std::integral_constant<int, 16>());
    return;
  }
  if constexpr (EPI) {
    if (!to_slab && p.ad_p && sh) {
      // ---- the product is a weight gradient and this launch also applies its Adam update (single-rank fused step): every
      // quad of the tile is written as the gradient and, in the same pass, updates p / m / v and the bf16 operand image; the
      // transposed image (the chain kernels' K-contiguous form) goes through a per-wave LDS block so that every lane writes
      // 64 contiguous bytes.  What a separate Adam kernel (28 B per parameter through ~120 CUs) and two transpose kernels did.
      float* hy = reinterpret_cast<float*>(gl_smem + 4 * 32 * 68 * 4 + 16);
      if (tid == 0) {
        const double st = (double)(*sh->step_ptr + sh->step_off);
        const double bc1 = 1.0 - pow(sh->hp.beta1, st), bc2 = 1.0 - pow(sh->hp.beta2, st);
        hy[0] = (float)(sh->hp.lr / bc1); hy[1] = (float)sqrt(bc2);
        // a chain kernel of this step gave up (poison), or no step left an update pending: gradient only
        hy[2] = ((sh->poison && *sh->poison != 0.f) || (sh->pending && *sh->pending == 0u)) ? 0.f : 1.f;
      }
      __syncthreads();
      const float step_size = hy[0], bc2s = hy[1];
      const bool active = hy[2] != 0.f;
#ifdef RN_FAULT_INJECT
      // fault-injection build only (make fault -> librecnet_hip_fault.so, tests/test_gpu_faults.py): the named store is left out so
      // that the tests which pin this update path can be shown to notice.  1: parameters, 2: bf16 operand image, 3: transposed
      // image, 4: Adam moments.  Never compiled into the product library.
      const int fault = sh->pad;
#else
      constexpr int fault = 0;
#endif
      const float nrm = p.ad_pnorm ? *p.ad_pnorm : 0.f;
      const float kreg = nrm > 0.f ? sh->hp.reg_coef / nrm : 0.f;
      const bool ams = sh->hp.amsgrad != 0;
      constexpr int TS = 40;                                  // row stride of the transposed block (bf16): 80 bytes
      bf16_t* tT = reinterpret_cast<bf16_t*>(gl_smem + 36864 + wave * (64 * TS * 2));
#pragma unroll
      for (int half = 0; half < 2; ++half) {
#pragma unroll
        for (int ii = 0; ii < 2; ++ii)
#pragma unroll
          for (int j = 0; j < NPB; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) stg[(ii * 16 + cr + r) * 68 + j * 16 + cc] = acc[half * 2 + ii][j][r];
        // every load of the half tile first (24 x 16 bytes per lane in flight: the accumulators are in LDS by now, the registers
        // are free) — load / update / store quad by quad was a chain of eight dependent round trips per half tile (the stores to
        // p / m / v may alias the next quad's loads as far as the compiler knows): 91 us of epilogue on a 221 us product
        const int c4 = (lane & 15) * 4;
        const int colq = n0 + wn + c4;
        const bool colok = colq < p.N && c4 < BN / 2;
        f32x4 p8[8], m8[8], v8[8], x8[8];
#pragma unroll
        for (int it = 0; it < 8; ++it) {
          const int row = m0 + wm + half * 32 + it * 4 + (lane >> 4);
          const size_t off = (size_t)row * p.ldc + colq;
          const bool ok = active && colok && row < p.M;
          p8[it] = ok ? *reinterpret_cast<const f32x4*>(p.ad_p + off) : f32x4{0.f, 0.f, 0.f, 0.f};
          m8[it] = ok ? *reinterpret_cast<const f32x4*>(p.ad_m + off) : f32x4{0.f, 0.f, 0.f, 0.f};
          v8[it] = ok ? *reinterpret_cast<const f32x4*>(p.ad_v + off) : f32x4{0.f, 0.f, 0.f, 0.f};
          x8[it] = (ok && ams) ? *reinterpret_cast<const f32x4*>(p.ad_vmax + off) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int it = 0; it < 8; ++it) {
          const int rl = it * 4 + (lane >> 4);
          const f32x4 v = *reinterpret_cast<const f32x4*>(stg + rl * 68 + c4);
          const int row = m0 + wm + half * 32 + rl, col = colq;
          if (row >= p.M || !colok) continue;
          const size_t off = (size_t)row * p.ldc + col;
          const f32x4 g4 = p.alpha * v;
          *reinterpret_cast<f32x4*>(p.C + off) = g4;
          if (!active) continue;
          const f32x4 p4 = p8[it];
          f32x4 m4 = m8[it], v4 = v8[it], x4 = x8[it];
          f32x4 n4;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float mm = m4[e], vv = v4[e], xx = x4[e];
            n4[e] = rn_adam_update(p4[e], g4[e], mm, vv, xx, kreg, 1.f, sh->hp, step_size, bc2s);
            m4[e] = mm; v4[e] = vv; x4[e] = xx;
          }
          if (fault != 4) { *reinterpret_cast<f32x4*>(p.ad_m + off) = m4; *reinterpret_cast<f32x4*>(p.ad_v + off) = v4; }
          if (ams) *reinterpret_cast<f32x4*>(p.ad_vmax + off) = x4;
          if (fault != 1) *reinterpret_cast<f32x4*>(p.ad_p + off) = n4;
          bf16x4 hb; hb[0] = (bf16_t)n4[0]; hb[1] = (bf16_t)n4[1]; hb[2] = (bf16_t)n4[2]; hb[3] = (bf16_t)n4[3];
          if (p.ad_img && fault != 2) *reinterpret_cast<bf16x4*>(reinterpret_cast<bf16_t*>(p.ad_img) + (size_t)row * p.ad_ld_img + col) = hb;
          if (p.ad_imgt) {
#pragma unroll
            for (int e = 0; e < 4; ++e) tT[(c4 + e) * TS + rl] = hb[e];
          }
        }
        if (p.ad_imgt && active && fault != 3) {
          // lane c owns column c of the wave's 32 x 64 block: 32 consecutive rows = 64 contiguous bytes of the transposed image
          const int col = n0 + wn + lane, r0 = m0 + wm + half * 32;
          if (col < p.N && r0 + 32 <= p.M && lane < BN / 2) {
            bf16_t* dst = reinterpret_cast<bf16_t*>(p.ad_imgt) + (size_t)col * p.ad_ld_imgt + r0;
#pragma unroll
            for (int q = 0; q < 4; ++q)
              *reinterpret_cast<f32x4*>(dst + q * 8) = *reinterpret_cast<const f32x4*>(tT + lane * TS + q * 8);
          }
        }
      }
      return;
    }
  }
  // ---- the common case on a straight path (round 5): whole 16-byte quads of an fp32 tile (or slab) or of a bf16 tile, nothing else
  // to do per element.  The general loop below re-tests every option per quad and fetched the bias element by element behind
  // branches — 16 dependent round trips per tile: 12.7k cycles of epilogue beside a 25k-cycle K loop at K = 1024, 12k beside 16k at
  // K = 512 (tools/micro/gemm_probe.hip).  Here the lane's bias quad is loaded once (its four columns are the same in all 16 rows)
  // and the 16 stores of a wave follow one another.
  {
    const bool plain = !p.accumulate && !p.C2 && !p.mse_ref && (p.N & 3) == 0 && (to_slab || !p.bias || ((((uintptr_t)p.bias) & 15) == 0));
    const bool f32_ok = plain && vec4;
    const bool b16_ok = plain && p.c_bf16 && !to_slab && ((p.ldc & 3) == 0) && ((((uintptr_t)p.C) & 7) == 0);
    if (f32_ok || b16_ok) {      // (block-uniform)
      const int c4 = (lane & 15) * 4, col = n0 + wn + c4;
      const bool colok = col < p.N && c4 < BN / 2;
      f32x4 b4 = f32x4{0.f, 0.f, 0.f, 0.f};
      if (!to_slab && p.bias && colok) b4 = *reinterpret_cast<const f32x4*>(p.bias + col);
      const float al = to_slab ? 1.f : p.alpha;
#pragma unroll
      for (int half = 0; half < 2; ++half) {
#pragma unroll
        for (int ii = 0; ii < 2; ++ii)
#pragma unroll
          for (int j = 0; j < NPB; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) stg[(ii * 16 + cr + r) * 68 + j * 16 + cc] = acc[half * 2 + ii][j][r];
        f32x4 v8[8];
#pragma unroll
        for (int it = 0; it < 8; ++it) v8[it] = *reinterpret_cast<const f32x4*>(stg + (it * 4 + (lane >> 4)) * 68 + c4);
#pragma unroll
        for (int it = 0; it < 8; ++it) {
          const int row = m0 + wm + half * 32 + it * 4 + (lane >> 4);
          if (row >= p.M || !colok) continue;
          const f32x4 w = al * v8[it] + b4;
          if (b16_ok) {
            bf16x4 hb; hb[0] = (bf16_t)w[0]; hb[1] = (bf16_t)w[1]; hb[2] = (bf16_t)w[2]; hb[3] = (bf16_t)w[3];
            *reinterpret_cast<bf16x4*>(reinterpret_cast<bf16_t*>(p.C) + (size_t)row * p.ldc + col) = hb;
          } else {
            *reinterpret_cast<f32x4*>(Cb + (size_t)row * ldc + col) = w;
          }
        }
      }
#ifdef GL_PROBE
      {
        const unsigned long long pe0 = __builtin_amdgcn_s_memtime();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned long long pe1 = __builtin_amdgcn_s_memtime();
        const int bl = by * gridDim.x + bx;
        if (bl < 64 && lane == 0 && z == 0) { gl_probe_buf[(bl * 8 + wave) * 8 + 6] = pe0 - pr_t; gl_probe_buf[(bl * 8 + wave) * 8 + 7] = pe1 - pr_t; }
      }
#endif
      return;
    }
  }
  float msq = 0.f;               // MSE epilogue: this lane's sum of squared differences
#pragma unroll
  for (int half = 0; half < 2; ++half) {
#pragma unroll
    for (int ii = 0; ii < 2; ++ii)
#pragma unroll
      for (int j = 0; j < NPB; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) stg[(ii * 16 + cr + r) * 68 + j * 16 + cc] = acc[half * 2 + ii][j][r];
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int rl = it * 4 + (lane >> 4), c4 = (lane & 15) * 4;
      const f32x4 v = *reinterpret_cast<const f32x4*>(stg + rl * 68 + c4);
      const int row = m0 + wm + half * 32 + rl, col = n0 + wn + c4;
      if (row >= p.M || col >= p.N || c4 >= BN / 2) continue;
      if (to_slab) {
        float* dst = Cb + (size_t)row * ldc + col;
        if (vec4 && col + 3 < p.N) *reinterpret_cast<f32x4*>(dst) = v;
        else for (int q = 0; q < 4 && col + q < p.N; ++q) dst[q] = v[q];
      } else {
        float o[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) o[q] = p.alpha * v[q] + ((p.bias && col + q < p.N) ? p.bias[col + q] : 0.f);
        if (p.c_bf16) {
          bf16_t* dst = reinterpret_cast<bf16_t*>(p.C) + (size_t)row * p.ldc + col;
          // one 8-byte store per lane (round 5: the element-wise form made this epilogue 40 % SLOWER than the fp32 one at half the
          // bytes — 18.0k against 12.7k cycles per tile, tools/micro/gemm_probe.hip)
          if (((p.ldc & 3) == 0) && ((((uintptr_t)p.C) & 7) == 0) && col + 3 < p.N) {
            bf16x4 hb; hb[0] = (bf16_t)o[0]; hb[1] = (bf16_t)o[1]; hb[2] = (bf16_t)o[2]; hb[3] = (bf16_t)o[3];
            *reinterpret_cast<bf16x4*>(dst) = hb;
          } else
          for (int q = 0; q < 4 && col + q < p.N; ++q) dst[q] = (bf16_t)o[q];
        } else {
          float* dst = Cb + (size_t)row * ldc + col;
          if (vec4 && col + 3 < p.N) {
            f32x4 w = f32x4{o[0], o[1], o[2], o[3]};
            if (p.accumulate) { const f32x4 old = *reinterpret_cast<const f32x4*>(dst); w += old; }
            if (p.mse_ref) {      // (host: N % 4 == 0, aligned reference rows, C2 set — see GemmArgs)
              const int s_ = row / p.mse_B, b_ = row - s_ * p.mse_B;
              const f32x4 e = *reinterpret_cast<const f32x4*>(p.mse_ref + (size_t)b_ * p.mse_bstride + (size_t)s_ * p.mse_sstride + col);
              const f32x4 d = w - e;
              msq += (d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3]);
              w = p.mse_gcoef * d;
              *reinterpret_cast<f32x4*>(dst) = w;
              const f32x4 wl = p.mse_lp * w;
              bf16x4 hb; hb[0] = (bf16_t)wl[0]; hb[1] = (bf16_t)wl[1]; hb[2] = (bf16_t)wl[2]; hb[3] = (bf16_t)wl[3];
              *reinterpret_cast<bf16x4*>(reinterpret_cast<bf16_t*>(p.C2) + (size_t)row * p.ldc2 + col) = hb;
              continue;
            }
            *reinterpret_cast<f32x4*>(dst) = w;
            if (p.C2) {
              bf16x4 hb; hb[0] = (bf16_t)w[0]; hb[1] = (bf16_t)w[1]; hb[2] = (bf16_t)w[2]; hb[3] = (bf16_t)w[3];
              *reinterpret_cast<bf16x4*>(reinterpret_cast<bf16_t*>(p.C2) + (size_t)row * p.ldc2 + col) = hb;
            }
          } else {
            for (int q = 0; q < 4 && col + q < p.N; ++q) {
              float w = o[q];
              if (p.accumulate) w += dst[q];
              dst[q] = w;
              if (p.C2) reinterpret_cast<bf16_t*>(p.C2)[(size_t)row * p.ldc2 + col + q] = (bf16_t)w;
            }
          }
        }
      }
    }
  }
#ifdef GL_PROBE
  {
    const unsigned long long pe0 = __builtin_amdgcn_s_memtime();      // stores issued
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long pe1 = __builtin_amdgcn_s_memtime();      // ... and acknowledged
    const int bl = by * gridDim.x + bx;
    if (bl < 64 && lane == 0 && z == 0) { gl_probe_buf[(bl * 8 + wave) * 8 + 6] = pe0 - pr_t; gl_probe_buf[(bl * 8 + wave) * 8 + 7] = pe1 - pr_t; }
  }
#endif
  if (p.mse_ref && !to_slab) {      // (block-uniform) the tile's sum of squares -> its slot of the partial sums
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) msq += __shfl_down(msq, off, 64);
    __syncthreads();               // every wave is done with its staging block
    float* red = reinterpret_cast<float*>(gl_smem);
    if (lane == 0) red[wave] = msq;
    __syncthreads();
    if (tid == 0) p.mse_part[(size_t)by * ((p.N + BN - 1) / BN) + bx] = (red[0] + red[1]) + (red[2] + red[3]);
  }
}

template <bool ACOL, bool BCOL, int NS, int TAG, int BN = 128>
__global__ __launch_bounds__(256) void gemm_lds_kernel(const GemmArgs p) {
  extern __shared__ __attribute__((aligned(16))) char gl_smem[];
  gemm_lds_tile<ACOL, BCOL, NS, BN, false>(p, blockIdx.x, blockIdx.y, blockIdx.z, gl_smem);
}

// ---- grouped launch: the tiles of up to GG_MAX products of one operand layout in ONE grid.  The hardware dispatcher deals the
// workgroups out in index order as CUs free up, i.e. the grid IS a work queue: the host orders the products by slice length
// (longest first), so the short tiles of the small products fill the tail of the schedule instead of running as launches of a
// few dozen workgroups each, and a split product is summed by its last-arriving slice (no reduction launch, no memset node).
#ifndef GG_MAX
#define GG_MAX 8
#endif
// stamp: optional pair of 100 MHz wall-clock words {first workgroup started, last workgroup left} (recnet_read_stamps)
struct GemmGroupArgs { GemmArgs p[GG_MAX]; int first[GG_MAX + 1]; int np; AdamShared ad; unsigned long long* stamp; };
template <bool ACOL, bool BCOL, int NS, bool EPI = false>
__global__ __launch_bounds__(256) void gemm_group_kernel(const GemmGroupArgs g) {
  extern __shared__ __attribute__((aligned(16))) char gl_smem[];
  int i = 0;
  const int bid = blockIdx.x;
#pragma unroll
  for (int k = 1; k < GG_MAX; ++k) if (k < g.np && bid >= g.first[k]) i = k;
  const GemmArgs& p = g.p[i];
  const int local = bid - g.first[i];
  const int tn = (p.N + GEMM_TILE - 1) / GEMM_TILE, tm = (p.M + GEMM_TILE - 1) / GEMM_TILE;
  // slices of a tile are neighbours in the queue (they finish together: the last arriver does not wait long for the others)
  const int z = local % p.splitk, t = local / p.splitk;
  if (g.stamp && bid == 0 && threadIdx.x == 0) __hip_atomic_store(g.stamp, (unsigned long long)wall_clock64(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  gemm_lds_tile<ACOL, BCOL, NS, 128, true, EPI>(p, t % tn, t / tn, z, gl_smem, &g.ad);
  if (g.stamp && threadIdx.x == 0) __hip_atomic_fetch_max(g.stamp + 1, (unsigned long long)wall_clock64(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  (void)tm;
}
