// Wide-tile bf16 MFMA GEMM for the step's large batched products (round 6).
//
//   C[M,N] (+)= alpha * sum_k A(m,k) B(n,k) (+ bias[n]),  A and B stored as bf16 (row or col operand layouts, gemm_lds.hpp).
//
// What differs from gemm_lds_kernel (128 x 128 tiles, 4 waves, two workgroups per CU):
//   * tile 256 x (32 NI), NI = 2 .. 8, ONE workgroup per CU with one wave per SIMD (WAVES = 4: wave grid 2 x 2, a wave owns
//     128 x 16 NI) — the form the vendor library's kernels have.  Per staged byte and per fragment byte read from LDS the tile
//     does 1.3 - 2 x the FLOPs of the 128 x 128 tile (256 x 160: 98 FLOP per staged byte against 64), which is what bounded the
//     small tile (fill path, LDS read path and MFMA pipe at their limits at once).  The tile WIDTH is the
//     per-shape knob: the host picks NI so that the product's tiles fill whole rounds of the chip (3100 x 6144: 507 tiles of
//     256 x 160 on 256 CUs).  (WAVES = 8, two waves per SIMD on 64 x 16 NI each, is kept as a template variant: measured slower —
//     the two waves of a SIMD meet at the same barrier, so their non-MFMA sections coincide instead of covering each other.)
//   * a wave's instruction stream is laid out by hand (sched_barrier between every group): every LDS fragment read and every
//     DMA request sits between two MFMAs, whose 16 pipe cycles cover its issue; the second k-step's fragments are read during
//     the first k-step's MFMAs, the NEXT stage's first fragments during the second's (software pipeline, one barrier per
//     64-deep k-tile in its middle).
//   * persistent: the grid is one workgroup per CU (or per free CU beside a chain kernel), workgroup p runs items p, p + P, ...;
//     the ring keeps prefetching ACROSS items, so the first stages of the next tile land while this tile's epilogue stores
//     drain.  No workgroup ever waits for another one (a split product is finished by the slice that arrives last, as in
//     gemm_group_kernel), so the launch is correct with any number of resident workgroups.
//   * accumulators are kept TRANSPOSED (mfma(B fragment, A fragment)): a lane then holds four consecutive columns of one output
//     row, i.e. the epilogue is one 16-byte store per fragment straight from the accumulator registers — no LDS staging pass.
//     The stores are raw-buffer stores that every lane executes (out-of-range lanes are dropped by the buffer's range check), so
//     their count is exact and the next wait for a stage can be a COUNTED s_waitcnt vmcnt: the stores stay in flight.
//   * split products exchange their partial tiles fragment-major (a lane re-reads exactly the 16-byte words it wrote: every
//     access a whole 1 KiB wave instruction), written through (sc1), summed in slice order by the last arriver.
// Operand requirements are those of gemm_lds.hpp (16-byte aligned bases, leading dimensions multiples of 8, zero padding);
// C: 16-byte aligned, ldc % 4 == 0, N % 4 == 0, M * ldc * 4 < 2^31; bias 16-byte aligned.
#pragma once
#include "gemm_lds.hpp"

struct GwProb {
  const void* A; const void* B; float* C; const float* bias; void* C2; float* ws; unsigned* cnt;
  int M, N, K, lda, ldb, ldc, ldc2;
  float alpha; int accumulate, c_bf16;
  int tn, tiles;            // tiles along N, tiles in all
  int splitk, kchunk;       // slices per tile; k extent of a slice (multiple of 64)
  int first;                // first item of this product in the launch's item list (items of a product: tile-major, slices adjacent)
};
struct GwArgs { GwProb p[GG_MAX]; int first[GG_MAX]; int np, items; unsigned long long* stamp; };      // first[k] = p[k].first (one scalar load for the scan)

template <int NI_, bool BCOL, int WAVES> struct GwCfg {
  static_assert(WAVES == 4 || WAVES == 8, "wave grid 2 x 2 or 4 x 2");
  static constexpr int MI = WAVES == 4 ? 8 : 4, NI = NI_, BM = 256, BN = 32 * NI_;
  static constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, STAGE = A_BYTES + B_BYTES;
  static constexpr int NS = (3 * STAGE + 64 <= 160 * 1024) ? 3 : 2;
  static constexpr int LDS = NS * STAGE + 64;
  static constexpr int PA = BM / 8 / WAVES;                        // DMA pieces per wave and stage, A (32 pieces of 1 KiB)
  static constexpr int PB = (BN / 8 + WAVES - 1) / WAVES;          // ... B (BN / 8 pieces; a wave short of a piece repeats one of its own)
  static constexpr int PW = PA + PB;
  static constexpr int NST = MI * NI_;                             // 16-byte stores per wave of the direct epilogue
  static_assert(!BCOL || (BN % 128) == 0, "col-layout B: whole 128-column images");
};

struct GwItem { int pi, t, z, m0, n0, kbeg, kend, nkt; };

template <int BN>
__device__ __forceinline__ GwItem gw_decode(const GwArgs& g, int item) {
  int i = 0;
#pragma unroll
  for (int k = 1; k < GG_MAX; ++k) if (k < g.np && item >= g.first[k]) i = k;
  const GwProb& p = g.p[i];
  GwItem it;
  it.pi = i;
  const int local = item - p.first;
  it.z = local % p.splitk; it.t = local / p.splitk;
  const int by = it.t / p.tn, bx = it.t - by * p.tn;
  it.m0 = by * 256; it.n0 = bx * BN;
  it.kbeg = it.z * p.kchunk;
  int ke = it.kbeg + p.kchunk; it.kend = ke < p.K ? ke : p.K;
  it.nkt = it.kend > it.kbeg ? (it.kend - it.kbeg + 63) >> 6 : 0;
  return it;
}

// one operand region of a stage: NPIECES DMA pieces of 1 KiB, wave w issues pieces w, w + WAVES, ... (general address form)
template <bool COL, int NPIECES, int WAVES>
__device__ __forceinline__ void gw_stage(char* region, const bf16_t* base, int ld, int row0, int rext, int k0, int K, int wave, int lane) {
  constexpr int PER = (NPIECES + WAVES - 1) / WAVES;
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    int piece = wave + WAVES * i;
    if (piece >= NPIECES) piece -= WAVES;      // (wave-uniform) repeats the wave's previous piece: same bytes to the same place
    const bf16_t* src;
    if (!COL) src = gl_piece_src<false>(base, ld, row0, rext, k0, K, piece, lane);
    else src = gl_piece_src<true>(base, ld, row0 + (piece >> 4) * 128, rext, k0, K, piece & 15, lane);
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)(region + piece * 1024), 16, 0, 0);
  }
}
template <bool COL, int NPIECES, int WAVES>
__device__ __forceinline__ void gw_zero_tail(char* region, int k0, int K, int wave, int lane) {
  constexpr int PER = (NPIECES + WAVES - 1) / WAVES;
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    int piece = wave + WAVES * i;
    if (piece >= NPIECES) piece -= WAVES;
    bool bad;
    if (!COL) { const int r = piece * 8 + (lane >> 3), cp = lane & 7; bad = k0 + ((cp ^ (r & 7)) << 3) >= K; }
    else bad = k0 + (piece & 15) * 4 + (lane >> 4) >= K;
    if (bad) *reinterpret_cast<f32x4*>(region + piece * 1024 + lane * 16) = f32x4{0.f, 0.f, 0.f, 0.f};
  }
}
template <bool COL>
__device__ __forceinline__ bf16x8 gw_frag(const char* region, int row, int ks, int lane) {
  if (!COL) return gl_frag<false>(region, row, ks, lane);
  return gl_frag<true>(region + (row >> 7) * 16384, row & 127, ks, lane);
}

// VGPR-destination loads inside the persistent loop are issued from inline asm: with an ordinary load anywhere in the loop hipcc's
// wait-count pass puts s_waitcnt vmcnt(0) in front of the first ds_read of EVERY k-tile (it cannot see the counted waits below
// and merges the epilogue's pending loads over the back edge) — the stage just requested would be waited for at once.  The asm
// forms are invisible to that pass; each group is followed by its own s_waitcnt that carries the loaded registers as operands.
__device__ __forceinline__ f32x4 gw_load16(const float* p) {
  f32x4 v; asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(p) : "memory"); return v;
}
__device__ __forceinline__ f32x4 gw_load16_sc1(const float* p) {
  f32x4 v; asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory"); return v;
}
template <int N> __device__ __forceinline__ void gw_wait_loads(f32x4 (&v)[N]) {
  if constexpr (N == 2) asm volatile("s_waitcnt vmcnt(0)" : "+v"(v[0]), "+v"(v[1]) :: "memory");
  else if constexpr (N == 3) asm volatile("s_waitcnt vmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]) :: "memory");
  else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]) :: "memory");
  else if constexpr (N == 5) asm volatile("s_waitcnt vmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]) :: "memory");
  else if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]) :: "memory");
  else if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]) :: "memory");
  else static_assert(N == 2, "gw_wait_loads: add the operand list");
}
// GW_PROBE (tools/micro/gemm_wide_probe.hip only): shader-clock stamps around the sections of the loop, summed per wave
#ifdef GW_PROBE
__device__ unsigned long long gw_probe_buf[256 * 8 * 8];
#define GW_PR(i) do { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); pr_acc[i] += n_ - pr_t; pr_t = n_; } while (0)
#else
#define GW_PR(i)
#endif
#ifndef GW_COUNTED_STORES
#define GW_COUNTED_STORES 1
#endif
#define GW_SB() __builtin_amdgcn_sched_barrier(0)
// One wave per SIMD (WAVES = 4) needs more than 256 registers per lane: the accumulators live in the accumulator file.  Left to
// itself hipcc's allocator kept part of them in VGPRs across the loop and copied four registers into AGPRs in front of an MFMA and
// back behind it (hundreds of v_accvgpr moves per k-tile); the "+a" constraint of an asm MFMA pins them.  What hipcc does not pad
// for an asm statement (cdna_hip_programming.md 5.7): the first MFMA behind the zeroing v_accvgpr_writes and the epilogue's reads
// of the last MFMA's result — both get an explicit s_nop run (gw_acc_settle).  An accumulate chain (D of one MFMA = C of the next
// on the same registers) needs no wait states.
template <bool PIN> __device__ __forceinline__ void gw_mma(f32x4& c, const bf16x8& b, const bf16x8& a) {
  if constexpr (PIN) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(b), "v"(a));
  else c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b, a, c, 0, 0, 0);
}
__device__ __forceinline__ void gw_acc_settle() { asm volatile("s_nop 15\n\ts_nop 15" ::: "memory"); }

template <bool ACOL, bool BCOL, int NI, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void gemm_wide_kernel(const GwArgs g) {
  typedef GwCfg<NI, BCOL, WAVES> Cf;
  constexpr int MI = Cf::MI;
  extern __shared__ __attribute__((aligned(16))) char gw_smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = (wave >> 1) * (MI * 16), wn = (wave & 1) * (NI * 16);
  constexpr int D = Cf::NS - 1;
  const int P = gridDim.x;
  if (g.stamp && blockIdx.x == 0 && tid == 0) __hip_atomic_store(g.stamp, (unsigned long long)wall_clock64(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);

  int item_c = blockIdx.x;
  if (item_c >= g.items) return;
  GwItem ci = gw_decode<Cf::BN>(g, item_c);
  int kt_c = 0;
  // load cursor (runs D stages ahead of the compute cursor, across items).  Per-lane source pointers of its item's pieces at
  // k = kbeg are formed once per item (lpa / lpb); a whole k-tile then costs one 64-bit add per piece.  Only the partial last
  // k-tile of a slice goes through the general address form with its clamps (gw_stage).
  int item_l = item_c; GwItem li = ci; int kt_l = 0; bool lvalid = true;
  const bf16_t* lpa[Cf::PA]; const bf16_t* lpb[Cf::PB];
  long lstep_a = 0, lstep_b = 0;      // elements per k-tile along the cursor's operands
  auto load_setup = [&]() {
    const GwProb& p = g.p[li.pi];
#pragma unroll
    for (int i = 0; i < Cf::PA; ++i) {
      const int piece = wave + WAVES * i;
      lpa[i] = ACOL ? gl_piece_src<true>(reinterpret_cast<const bf16_t*>(p.A), p.lda, li.m0 + (piece >> 4) * 128, p.M, li.kbeg, 0x7fffffff, piece & 15, lane)
                    : gl_piece_src<false>(reinterpret_cast<const bf16_t*>(p.A), p.lda, li.m0, p.M, li.kbeg, 0x7fffffff, piece, lane);
    }
#pragma unroll
    for (int i = 0; i < Cf::PB; ++i) {
      int piece = wave + WAVES * i;
      if (piece >= Cf::BN / 8) piece -= WAVES;
      lpb[i] = BCOL ? gl_piece_src<true>(reinterpret_cast<const bf16_t*>(p.B), p.ldb, li.n0 + (piece >> 4) * 128, p.N, li.kbeg, 0x7fffffff, piece & 15, lane)
                    : gl_piece_src<false>(reinterpret_cast<const bf16_t*>(p.B), p.ldb, li.n0, p.N, li.kbeg, 0x7fffffff, piece, lane);
    }
    lstep_a = ACOL ? 64l * p.lda : 64l; lstep_b = BCOL ? 64l * p.ldb : 64l;
  };
  // piece q (0 .. PW-1) of the cursor's k-tile, fast form
  auto issue_piece = [&](auto qc, int slot) {
    constexpr int q = decltype(qc)::value;
    char* st = gw_smem + slot * Cf::STAGE;
    if constexpr (q < Cf::PA) {
      const bf16_t* src = lpa[q] + (long)kt_l * lstep_a;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(st + (wave + WAVES * q) * 1024), 16, 0, 0);
    } else {
      constexpr int i = q - Cf::PA;
      int piece = wave + WAVES * i;
      if (piece >= Cf::BN / 8) piece -= WAVES;
      const bf16_t* src = lpb[i] + (long)kt_l * lstep_b;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(st + Cf::A_BYTES + piece * 1024), 16, 0, 0);
    }
  };
  auto load_tail = [&]() { return li.kbeg + kt_l * 64 + 64 > li.kend; };      // the cursor's k-tile is a partial one
  auto stage_general = [&](int slot) {
    const GwProb& p = g.p[li.pi];
    char* st = gw_smem + slot * Cf::STAGE;
    const int k0 = li.kbeg + kt_l * 64;
    gw_stage<ACOL, Cf::BM / 8, WAVES>(st, reinterpret_cast<const bf16_t*>(p.A), p.lda, li.m0, p.M, k0, li.kend, wave, lane);
    gw_stage<BCOL, Cf::BN / 8, WAVES>(st + Cf::A_BYTES, reinterpret_cast<const bf16_t*>(p.B), p.ldb, li.n0, p.N, k0, li.kend, wave, lane);
  };
  auto load_advance = [&]() {
    if (++kt_l == li.nkt) {
      kt_l = 0; item_l += P;
      if (item_l >= g.items) lvalid = false; else { li = gw_decode<Cf::BN>(g, item_l); load_setup(); }
    }
  };
  load_setup();

  f32x4 acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // ---- software pipeline.  Iteration s multiplies stage s (slot `slot`); its first k-step's fragments are ALREADY in registers
  // (fa0 / fb0, read in the second half of iteration s - 1).  In the middle of iteration s — behind the first k-step's MFMA rows,
  // between which the second k-step's fragments were requested — the wave waits for its DMAs of stage s + 1, meets the other
  // waves at the one barrier of the iteration and reads stage s + 1's first fragments between the second k-step's MFMAs.
  // Ring hazards: a slot is rewritten (DMAs of stage s + D, requested between the MFMAs of iteration s) only after every wave
  // has passed the barrier in the middle of iteration s - 1, in front of which it has waited for all its reads of stage s - 1
  // (lgkmcnt(0)); a stage is read only behind the barrier that follows every wave's wait for its own DMAs of it.
  constexpr int EARLY = D == 1 ? Cf::PW : (Cf::PW < MI ? Cf::PW : MI);      // pieces requested in front of the barrier
  auto wait_dyn = [&](int n) {      // s_waitcnt vmcnt(n), n one of the six values the schedule produces (block-uniform)
    if (n == 0) gl_wait_vmcnt<0>();
    else if (n == EARLY) gl_wait_vmcnt<EARLY>();
    else if (n == Cf::PW) gl_wait_vmcnt<Cf::PW>();
    else if (n == Cf::NST) gl_wait_vmcnt<Cf::NST>();
    else if (n == Cf::NST + EARLY) gl_wait_vmcnt<(Cf::NST + EARLY <= 63 ? Cf::NST + EARLY : 0)>();
    else if (n == Cf::NST + Cf::PW) gl_wait_vmcnt<(Cf::NST + Cf::PW <= 63 ? Cf::NST + Cf::PW : 0)>();
    else gl_wait_vmcnt<0>();
  };
  constexpr bool COUNTED = GW_COUNTED_STORES && D >= 2 && Cf::NST + Cf::PW <= 63;
  int nst1 = 0;      // DMAs of the stage behind the one the prologue waits for
#pragma unroll
  for (int d = 0; d < D; ++d)
    if (lvalid) { stage_general(d); load_advance(); if (d == 1) nst1 = Cf::PW; }
  int slot = 0, lslot = D % Cf::NS;
  bool pend = false;      // the direct epilogue's NST stores are younger than the stage the next wait is for
  bf16x8 fa0[MI], fb0[NI], fa1[MI], fb1[NI];
  auto fix_tail = [&](char* stg, int k0, int kend) {      // stage landed for every wave: zero what a partial k-tile has beyond K
    if (k0 + 64 > kend) {      // (block-uniform)
      gw_zero_tail<ACOL, Cf::BM / 8, WAVES>(stg, k0, kend, wave, lane);
      gw_zero_tail<BCOL, Cf::BN / 8, WAVES>(stg + Cf::A_BYTES, k0, kend, wave, lane);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
    }
  };
  wait_dyn(nst1);
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  fix_tail(gw_smem, ci.kbeg, ci.kend);
#pragma unroll
  for (int i = 0; i < MI; ++i) fa0[i] = gw_frag<ACOL>(gw_smem, wm + i * 16, 0, lane);
#pragma unroll
  for (int j = 0; j < NI; ++j) fb0[j] = gw_frag<BCOL>(gw_smem + Cf::A_BYTES, wn + j * 16, 0, lane);
  GwItem nx = ci;

#ifdef GW_PROBE
  unsigned long long pr_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, pr_t = __builtin_amdgcn_s_memtime();
#endif
  for (;;) {
   // ---- the k-tiles of one item: nothing but MFMAs touches the accumulators in here (they stay in the accumulator registers;
   // with the epilogue inside this loop hipcc kept them in VGPRs across the back edge and copied every one of them into an AGPR
   // in front of its MFMA and back, 4 + 4 moves per MFMA)
   for (;;) {
    char* cur = gw_smem + slot * Cf::STAGE;
    slot = slot + 1 == Cf::NS ? 0 : slot + 1;
    char* nxt = gw_smem + slot * Cf::STAGE;
    // request of the stage D ahead: a partial k-tile in one go here, a whole one piece by piece between the MFMAs below (the
    // CU's address path takes ~16 clocks per piece: all waves asking at once queued for ~1000 clocks with the matrix pipes idle)
    bool spread = false, staged = false; int early = 0;
    if (lvalid) { staged = true; if (load_tail()) { stage_general(lslot); load_advance(); early = Cf::PW; } else { spread = true; early = EARLY; } }
    GW_PR(0);
    GW_SB();
    // one MFMA row = the NI MFMAs of fragment row r of A; between its MFMAs: up to two fragment reads and one DMA request
#define GW_PIECE(q) { if constexpr ((q) >= 0 && (q) < Cf::PW) { if (spread) issue_piece(std::integral_constant<int, ((q) >= 0 && (q) < Cf::PW ? (q) : 0)>(), lslot); GW_SB(); } }
#define GW_ROW(FA, FB, r, RD_A, RD_B, Q0, Q1)                                                                                   \
    {                                                                                                                          \
      gw_mma<WAVES == 4>(acc[r][0], FB[0], FA[r]); GW_SB();                                                                  \
      RD_A; GW_SB();                                                                                                           \
      gw_mma<WAVES == 4>(acc[r][1], FB[1], FA[r]); GW_SB();                                                                  \
      RD_B; GW_SB();                                                                                                           \
      _Pragma("unroll") for (int j = 2; j < NI; ++j) { gw_mma<WAVES == 4>(acc[r][j], FB[j], FA[r]); GW_SB();                 \
        if (j == 2) GW_PIECE(Q0) if (j == 3) GW_PIECE(Q1) }                                                                    \
      if (NI < 3) GW_PIECE(Q0) if (NI < 4) GW_PIECE(Q1)                                                                        \
    }
#define GW_RA1(r) fa1[r] = gw_frag<ACOL>(cur, wm + (r) * 16, 32, lane)
#define GW_RB1(r) { if constexpr ((r) < NI) fb1[(r) < NI ? (r) : 0] = gw_frag<BCOL>(cur + Cf::A_BYTES, wn + ((r) < NI ? (r) : 0) * 16, 32, lane); }
#define GW_RA0(r) fa0[r] = gw_frag<ACOL>(nxt, wm + (r) * 16, 0, lane)
#define GW_RB0(r) { if constexpr ((r) < NI) fb0[(r) < NI ? (r) : 0] = gw_frag<BCOL>(nxt + Cf::A_BYTES, wn + ((r) < NI ? (r) : 0) * 16, 0, lane); }
    // first k-step; the second k-step's fragments between its MFMAs.  DMA pieces: D == 1: two per row, D >= 2: one per row
#define GW_H0(r) GW_ROW(fa0, fb0, r, GW_RA1(r), GW_RB1(r), (D == 1 ? 2 * (r) : (r)), (D == 1 ? 2 * (r) + 1 : -1))
    GW_H0(0) GW_H0(1) GW_H0(2) GW_H0(3)
    if constexpr (MI == 8) { GW_H0(4) GW_H0(5) GW_H0(6) GW_H0(7) }
    if constexpr (NI > MI) {      // (NI = 8 with MI = 4) the B fragments the rows did not reach
#pragma unroll
      for (int j = MI; j < NI; ++j) fb1[j] = gw_frag<BCOL>(cur + Cf::A_BYTES, wn + j * 16, 32, lane);
      GW_SB();
    }
    GW_PR(1);
    // ---- middle of the iteration: the next stage
    const bool last_kt = kt_c + 1 == ci.nkt;
    const bool has_next = !last_kt || item_c + P < g.items;
    __builtin_amdgcn_s_waitcnt(0xC07F);      // lgkmcnt(0): every LDS read of this stage has returned (ring hazard, above).  The builtin,
                                             // not inline asm: hipcc's wait-count pass sees it and does not wait again for the second
                                             // k-step's fragments behind the next stage's fragment reads
    GW_SB();
    if (has_next) {
      wait_dyn((D == 1 ? 0 : early) + ((COUNTED && pend) ? Cf::NST : 0));
      pend = false;
      GW_PR(2);
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      GW_PR(3);
      if (last_kt) nx = gw_decode<Cf::BN>(g, item_c + P);
      fix_tail(nxt, last_kt ? nx.kbeg : ci.kbeg + (kt_c + 1) * 64, last_kt ? nx.kend : ci.kend);
    }
    GW_SB();
    // second k-step; the NEXT stage's first fragments between its MFMAs (has_next false: the last stage is read again, unused)
#define GW_H1(r) GW_ROW(fa1, fb1, r, GW_RA0(r), GW_RB0(r), (D == 1 ? -1 : MI + (r)), -1)
    if (!has_next) nxt = cur;
    GW_H1(0) GW_H1(1) GW_H1(2) GW_H1(3)
    if constexpr (MI == 8) { GW_H1(4) GW_H1(5) GW_H1(6) GW_H1(7) }
    if constexpr (NI > MI) {
#pragma unroll
      for (int j = MI; j < NI; ++j) fb0[j] = gw_frag<BCOL>(nxt + Cf::A_BYTES, wn + j * 16, 0, lane);
      GW_SB();
    }
#undef GW_H0
#undef GW_H1
#undef GW_ROW
#undef GW_PIECE
    if (spread) load_advance();
    if (staged) lslot = lslot + 1 == Cf::NS ? 0 : lslot + 1;
    GW_PR(4);
#ifdef GW_PROBE
    pr_acc[6] += 1;
#endif
    if (++kt_c == ci.nkt) break;
   }
    if (WAVES == 4) gw_acc_settle();

    // ------------------------------------------------------------------ the item is complete: epilogue
    {
      const GwProb& p = g.p[ci.pi];
      const int mrow = ci.m0 + wm + (lane & 15);              // + 16 i
      const int ncol = ci.n0 + wn + 4 * (lane >> 4);          // + 16 j
      bool run_epilogue = true;
      if (p.splitk > 1) {
        // partial tile -> slab (z, t), fragment-major; sc1: written through, no release fence (gemm_lds.hpp, FIX)
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)p.ws, 0, 0x7ffffffc, 0x00020000);
        constexpr unsigned TILE_B = (unsigned)Cf::BM * Cf::BN * 4u;
        const unsigned base = ((unsigned)ci.z * (unsigned)p.tiles + (unsigned)ci.t) * TILE_B + (unsigned)wave * ((unsigned)MI * NI * 1024u) + (unsigned)lane * 16u;
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
          for (int j = 0; j < NI; ++j)
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4_t, acc[i][j]), rs, base + (unsigned)(i * NI + j) * 1024u, 0, 16);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        unsigned* flag = reinterpret_cast<unsigned*>(gw_smem + Cf::NS * Cf::STAGE);
        unsigned* cn = p.cnt + ci.t;
        if (tid == 0) *flag = __hip_atomic_fetch_add(cn, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        const unsigned got = *flag;
        __syncthreads();                 // (the flag word is rewritten by the next split item)
        if (got != (unsigned)(p.splitk - 1)) run_epilogue = false;
        else {
          if (tid == 0) __hip_atomic_store(cn, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // ready for the next launch
#if RC_ACQUIRE_INV
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
#endif
          const unsigned b0 = (unsigned)ci.t * TILE_B + (unsigned)wave * ((unsigned)MI * NI * 1024u) + (unsigned)lane * 16u;
          const unsigned zs = (unsigned)p.tiles * TILE_B;
#pragma unroll
          for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
          // slice order, whoever arrived last; per fragment row i: two slices' loads in flight (2 NI x 16 bytes per lane)
#pragma unroll
          for (int i = 0; i < MI; ++i) {
            for (int zz = 0; zz < p.splitk; zz += 2) {
              f32x4 pa[NI], pb[NI];
              const bool two_z = zz + 1 < p.splitk;
              const float* s0 = p.ws + ((size_t)b0 + (size_t)zz * zs) / 4 + (size_t)i * NI * 256;
              const float* s1 = two_z ? s0 + zs / 4 : s0;      // (odd count: the last slice is read twice, added once)
#pragma unroll
              for (int j = 0; j < NI; ++j) { pa[j] = gw_load16_sc1(s0 + j * 256); pb[j] = gw_load16_sc1(s1 + j * 256); }
              gw_wait_loads(pa); gw_wait_loads(pb);
#pragma unroll
              for (int j = 0; j < NI; ++j) { acc[i][j] += pa[j]; if (two_z) acc[i][j] += pb[j]; }
            }
          }
        }
      }
      if (run_epilogue) {
        f32x4 b4[NI];
#pragma unroll
        for (int j = 0; j < NI; ++j) b4[j] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (p.bias) {      // (columns beyond N re-read the last whole quad; their stores are dropped)
#pragma unroll
          for (int j = 0; j < NI; ++j) { const int n = ncol + j * 16; b4[j] = gw_load16(p.bias + (n + 4 <= p.N ? n : p.N - 4)); }
          gw_wait_loads(b4);
        }
        if (!p.c_bf16) {
          const __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc((void*)p.C, 0, (unsigned)p.M * (unsigned)p.ldc * 4u, 0x00020000);
          const bool acc_c = p.accumulate != 0;
#pragma unroll
          for (int i = 0; i < MI; ++i) {
            const int m = mrow + i * 16;
            f32x4 old[NI];
            if (acc_c) {      // (block-uniform)
              const int mc = m < p.M ? m : p.M - 1;
#pragma unroll
              for (int j = 0; j < NI; ++j) { const int n = ncol + j * 16; old[j] = gw_load16(p.C + (size_t)mc * p.ldc + (n + 4 <= p.N ? n : p.N - 4)); }
              gw_wait_loads(old);
            }
#pragma unroll
            for (int j = 0; j < NI; ++j) {
              const int n = ncol + j * 16;
              const unsigned off = (m < p.M && n < p.N) ? ((unsigned)m * (unsigned)p.ldc + (unsigned)n) * 4u : 0xFFFFFF00u;
              f32x4 w = p.alpha * acc[i][j] + b4[j];
              if (acc_c) w += old[j];
              __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4_t, w), rc, off, 0, 0);
              if (p.C2 && m < p.M && n < p.N) {
                bf16x4 hb; hb[0] = (bf16_t)w[0]; hb[1] = (bf16_t)w[1]; hb[2] = (bf16_t)w[2]; hb[3] = (bf16_t)w[3];
                *reinterpret_cast<bf16x4*>(reinterpret_cast<bf16_t*>(p.C2) + (size_t)m * p.ldc2 + n) = hb;
              }
            }
          }
          pend = !p.C2;
        } else {
          const __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc((void*)p.C, 0, (unsigned)p.M * (unsigned)p.ldc * 2u, 0x00020000);
#pragma unroll
          for (int i = 0; i < MI; ++i) {
            const int m = mrow + i * 16;
#pragma unroll
            for (int j = 0; j < NI; ++j) {
              const int n = ncol + j * 16;
              const unsigned off = (m < p.M && n < p.N) ? ((unsigned)m * (unsigned)p.ldc + (unsigned)n) * 2u : 0xFFFFFF00u;
              const f32x4 w = p.alpha * acc[i][j] + b4[j];
              bf16x4 hb; hb[0] = (bf16_t)w[0]; hb[1] = (bf16_t)w[1]; hb[2] = (bf16_t)w[2]; hb[3] = (bf16_t)w[3];
              typedef int i32x2_t __attribute__((ext_vector_type(2)));
              __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(i32x2_t, hb), rc, off, 0, 0);
            }
          }
          pend = true;
        }
      }
    }
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (WAVES == 4) gw_acc_settle();
    GW_PR(5);
#ifdef GW_PROBE
    pr_acc[7] += 1;
#endif
    item_c += P;
    if (item_c >= g.items) break;
    ci = nx;
    kt_c = 0;
  }
#ifdef GW_PROBE
  if (lane == 0) for (int i = 0; i < 8; ++i) gw_probe_buf[(blockIdx.x * 8 + wave) * 8 + i] = pr_acc[i];
#endif
  if (g.stamp && tid == 0) __hip_atomic_fetch_max(g.stamp + 1, (unsigned long long)wall_clock64(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
