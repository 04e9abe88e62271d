// The local reconstructor's forward chain (local_reconstructor.py:37-55 inside train.py:122-123) as ONE launch:
//   for s < F:  Whr = hr_{s-1} . W_r^T ;  beta[t] = w . tanh(Whr + Ud[t] + b) ;  x = drop((1/T) sum_t beta[t] h_t) ;
//               gates = x . W_ih^T + hr_{s-1} . W_hh^T + bias ;  (hr_s, cr_s) = cell
// The per-launch path runs four kernels per step (attention GEMM, attention, gate GEMM, cell: 33 us per step at
// B = 100, R = 1536) and re-streams [W_ih | W_hh] (25 MB) every step.  Here the weights are read once and a step is two
// hand-overs between two kinds of resident workgroups (one per CU, rec_chain.hpp's exchange discipline):
//   U (unit owner): 16 hidden units x one row part.  Its 64 gate rows of W_hh stay in MFMA B-operand registers, its
//      rows of W_ih as B fragments in LDS, its 16 columns of W_r in registers.  Per step: (x_s arrived) x_s . W_ih^T on top
//      of the recurrent part, 4-wave K reduction, cell pointwise, publishes hr_s (k-group-major panel) AND its rank-16
//      contribution hr_s[:, own units] . W_r[:, own units]^T to the next step's attention projection (so that no
//      workgroup has to wait for all of hr_s and run a second GEMM before the attention can start); then, off the
//      critical path, the recurrent product hr_s . W_hh^T for the next step.
//   C (caption owner): two captions.  Their decoder states h_t (fp32) and projections Ud[t] stay in registers for all F
//      steps.  Per step: sums the NG rank-16 contributions to Whr, scores beta over the T decoder steps (wavefront
//      reductions), x_s = dropout(mean_t beta h_t), publishes x_s (panel) and the tensors the backward needs.
//   One extra workgroup relays the two barriers per step (all U arrived -> C may read; all C arrived -> U may read).
// Limits (host-checked): bf16 path, B <= 112, R % 32 == 0, R <= 2048, H % 32 == 0, H <= 512, A <= 128, T <= 32,
// R/16 * row parts + ceil(B/2) + 1 workgroups resident at once (<= CU count, <= 256 flags).
#pragma once
#include "common.hpp"
#include "rec_chain.hpp"

struct LocChainArgs {
  int F, T, B, R, H, A, gru;
  int Bs;                             // rows per step of the saved [F][.][.] / [T][.][.] tensors (= batch size; B = rows of THIS launch, a row group; pointers pre-offset; Pw / panels are private to the launch)
  int NU, NG, MS, NC;                 // unit-owner workgroups = NG unit groups x MS row parts; caption workgroups
  int relay;                          // 1: workgroup NU + NC relays the barriers; 0 (no CU left for it, NU + NC = CU count): the last caption workgroup does (lc_wait_or_relay)
  const bf16_t* W; int ldw;           // [4R][ldw] packed [W_ih (H) | W_hh (R) | 0], gate-major rows
  const bf16_t* Wr; int ldwr;         // [A][ldwr] W_r
  const bf16_t* Wst;                  // hybrid form: the streamed k-steps of W_hh as MFMA B fragments in consumption order,
                                      // [NG][4 waves][STEPS - SR][4][64 lanes][8] (lc_pack_stream_kernel): every wave load is 1 KB contiguous
  const float* bias;                  // [4R] b_ih + b_hh in the 4-block layout
  const float* Hs; const float* Ud;   // [T][B][H] decoder states, [T][B][A] their projections U_r h_t
  const float* ab; const float* w;    // [A], [A]
  float* Hr; float* Cr; float* acts;  // [F][B][R], [F][B][R], [F][B][4R]
  bf16_t* Hlp; int ld_hlp;            // [F][B][ld_hlp] row-major operand copy of hr_s (zero padded)
  bf16_t* Xcat; int ld_xcat;          // [F][B][ld_xcat]: x_s -> columns [0, H)
  float* beta; float* Whr;            // [F][B][T], [F][B][A]
  bf16_t* PanH; bf16_t* PanX;         // exchange: [F][rc_pan_elems(R)], [F][rc_pan_elems(H)]
  _Float16* Pw;                       // exchange: [F][B][NG][A] rank-16 contributions to Whr of step s (written at s-1), fp16:
                                      // |contribution| < 0.5, rounding 2^-11 per term — a quarter of the bf16 operand rounding
                                      // already inside each product — for half the bytes of the step's largest store burst
  unsigned* bar; unsigned* epoch; float* poison;
  DropDesc dd;
  unsigned long long* ts;             // probe builds only (LC_PROBE): [role][step][8] wall-clock stamps of one workgroup per role
};

#ifdef LC_PROBE
#define LC_TS(role, step, i) do { if (tid == 0) p.ts[((size_t)(role) * 64 + (step)) * 8 + (i)] = wall_clock64(); } while (0)
#else
#define LC_TS(role, step, i) do { } while (0)
#endif
#define LC_CPW 2              // captions per C workgroup
// hybrid forward chain: k-steps per wave resident in registers for STEPS k-steps in all — as many as compile without scratch
// next to the streaming ring (28: 14; 32: 12; 16 resident spill 132 bytes per lane)
#ifndef LC_HYB_SR28
#define LC_HYB_SR28 14
#endif
#define LC_HYB_SR(STEPS) ((STEPS) == 28 ? LC_HYB_SR28 : 12)
#define LC_MAX_PHASE 128       // barrier words are (launch epoch << 7) + phase: every phase number of a launch stays below this
__device__ __forceinline__ void lc_poll(const unsigned* flags, int n, unsigned target, unsigned* bar, unsigned& spin) {
  // wave 0 of the relay workgroup: all n <= 256 flags have reached `target`
  const int l = threadIdx.x;
  const unsigned* f0 = flags + (l < n ? l : n - 1);
  const unsigned* f1 = flags + (l + 64 < n ? l + 64 : n - 1);
  const unsigned* f2 = flags + (l + 128 < n ? l + 128 : n - 1);
  const unsigned* f3 = flags + (l + 192 < n ? l + 192 : n - 1);
  for (;;) {
    const unsigned a0 = __hip_atomic_load(f0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned a1 = __hip_atomic_load(f1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned a2 = __hip_atomic_load(f2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned a3 = __hip_atomic_load(f3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const bool ok = (int)(a0 - target) >= 0 && (int)(a1 - target) >= 0 && (int)(a2 - target) >= 0 && (int)(a3 - target) >= 0;
    if (__all(ok)) break;
    if (rc_give_up(bar, spin)) break;
  }
}
__device__ __forceinline__ void lc_release(unsigned* rel, unsigned v) {
  if (threadIdx.x < 8) __hip_atomic_store(rel + threadIdx.x * 32, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// workers: wave 0 polls this workgroup's copy of a release word (one 128-byte line per group of workgroups)
__device__ __forceinline__ void lc_wait(const unsigned* rel, unsigned target, unsigned* bar) {
  if (threadIdx.x < 64) {
    const unsigned* r = rel + (blockIdx.x & 7) * 32;
    unsigned spin = 0;
    while ((int)(__hip_atomic_load(r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target) < 0) { if (rc_give_up(bar, spin)) break; }
    if (RC_ACQUIRE_INV) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
  __syncthreads();
}
// A launch without a CU for the relay workgroup (NU + NC = CU count) folds the relay into the LAST caption workgroup, in
// the time it would spend waiting anyway: where the others wait for a release word it polls the arrival flags itself and
// then writes the release words (`me` = this workgroup is that relay).  One poller, as with the dedicated relay — with every
// waiter polling the 224 flags the U -> C hand-over took 4.3 us instead of ~2.
__device__ __forceinline__ void lc_wait_or_relay(bool me, unsigned* rel, const unsigned* flags, int n, unsigned target, unsigned* bar) {
  if (!me) { lc_wait(rel, target, bar); return; }
  if (threadIdx.x < 64) {
    unsigned spin = 0;
    lc_poll(flags, n, target, bar, spin);
    lc_release(rel, target);
    if (RC_ACQUIRE_INV) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
  __syncthreads();
}
__device__ __forceinline__ void lc_arrive(unsigned* flag, unsigned v) {
  if (threadIdx.x == 0) __hip_atomic_store(flag, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// (16-byte write-through stores: rc_store16f / rc_store16, rec_chain.hpp)
__device__ __forceinline__ void lc_store16f(float* dst, f32x4 v) { rc_store16f(dst, v); }
__device__ __forceinline__ void lc_store16(bf16_t* dst, const bf16_t* src) { rc_store16(dst, src); }
// A pointer that went through an empty asm (to keep its address arithmetic out of loop-invariant hoisting) has lost its address
// space: loads through it are FLAT loads, which may return out of order with global loads, so every wait next to them becomes
// vmcnt(0) and a register ring of loads drains at each use.  These put the address space back.
#define LC_GLOBAL __attribute__((address_space(1)))
typedef const LC_GLOBAL bf16x8* lc_gbf16x8;
typedef const LC_GLOBAL f32x4* lc_gf32x4;
typedef const LC_GLOBAL float* lc_gf32;
template <typename T> __device__ __forceinline__ const LC_GLOBAL T* lc_launder_v(const T* ptr) { asm volatile("" : "+v"(ptr)); return (const LC_GLOBAL T*)ptr; }
template <typename T> __device__ __forceinline__ const LC_GLOBAL T* lc_launder_s(const T* ptr) { asm volatile("" : "+s"(ptr)); return (const LC_GLOBAL T*)ptr; }
typedef short lc_s4 __attribute__((ext_vector_type(4)));
typedef _Float16 lc_h4 __attribute__((ext_vector_type(4)));

// release words: relU (word 0 of each line) = "every U workgroup has finished step .", relC (word 16) = "every C ...".
// Flag / release values are fb + phase, fb = launch epoch << 7 (rec_chain.hpp).  Phases: U arrives with s + 1 after
// publishing hr_s, C arrives with s + 1 after publishing x_s; both arrive with F + 1 when they are done, after which the
// relay bumps the launch epoch (every workgroup has read it by then).
// SR < STEPS (R above 2048, BASELINE configs[4]: R = 3584): the hybrid form.  64 gate rows x (H + R) of [W_ih | W_hh] are
// 512 KB per unit-owner workgroup at R = 3584 — more than a CU's register file.  The fragments of the first SR k-steps of
// every wave stay resident in registers as before (SR = 16: 256 registers per lane, 57 % of W_hh at R = 3584), the other
// STEPS - SR k-steps are STREAMED from memory every step through a register ring, inside the recurrent product that runs
// off the critical path (their addresses do not depend on the chain: the first ring slots are requested before the wait).
// Per step the chip then streams 43 MB instead of the 117 MB the per-step GEMM reads.  NQ = groups of four unit groups the
// caption workgroups sum (NG <= 4 NQ).
template <int STEPS, int PF, int RB, int SR = STEPS, int NQ = 32>
__global__ __launch_bounds__(256) void loc_chain_kernel(const LocChainArgs p) {
  constexpr int CG = 4, UW = 16, ROWS = RB * 16, NP = STEPS / 2, KG = UW / 8, SX = 4, NPR = SR / 2;
  static_assert(SR % 2 == 0 && SR <= STEPS && SR >= 2 * PF, "resident k-steps: whole pairs, at least the prefetch distance");
  extern __shared__ __attribute__((aligned(16))) float lc_smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wg = blockIdx.x, R = p.R, B = p.B, Bs = p.Bs, H = p.H, A = p.A, F = p.F, T = p.T;
  const unsigned ep = rc_epoch_read(p.epoch), fb = ep << 7;
  unsigned* relU = p.bar + 256; unsigned* relC = p.bar + 256 + 16;
  const size_t pan_h = rc_pan_elems(R), pan_x = rc_pan_elems(H);

  // ================================================================================== relay workgroup
  if (p.relay && wg == p.NU + p.NC) {
    if (tid < 64) {
      unsigned spin = 0;
      for (int s = 0; s < F; ++s) {
        if (s >= 1) { lc_poll(p.bar, p.NU, fb + (unsigned)s, p.bar, spin); lc_release(relU, fb + (unsigned)s); }
        LC_TS(2, s, 0);
        lc_poll(p.bar + p.NU, p.NC, fb + (unsigned)(s + 1), p.bar, spin); lc_release(relC, fb + (unsigned)(s + 1));
        LC_TS(2, s, 1);
      }
      lc_poll(p.bar, p.NU + p.NC, fb + (unsigned)(F + 1), p.bar, spin);      // everybody is done (and has read the epoch)
      if (tid == 0) {
        __hip_atomic_store(p.epoch, ep + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); rc_stamp_slot(p.epoch)[1] = wall_clock64();
        if (__hip_atomic_load(p.bar + 257, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) *p.poison = __builtin_nanf("");
      }
    }
    return;
  }

  // ================================================================================== caption workgroups
  if (wg >= p.NU) {
    float* swh = lc_smem;                                   // [LC_CPW][128]
    float* sbeta = swh + LC_CPW * 128;                      // [LC_CPW][32]
    bf16_t* xl = reinterpret_cast<bf16_t*>(sbeta + LC_CPW * 32);   // [LC_CPW][512]
    float* spw = sbeta + LC_CPW * 32 + LC_CPW * 256;               // [LC_CPW][4][128] partial sums of the Whr contributions
    float* swab = spw + LC_CPW * 4 * 128;                          // [128][2] (w_k, b_k)
    float* spb = swab + 256;                                       // [LC_CPW][4][32] partial scores
    const int ci = wg - p.NU, c = tid >> 7, j = tid & 127;
    const int b = ci * LC_CPW + c;
    const bool bok = b < B;
    const int bb = bok ? b : 0;
    // residents: this thread's four columns of h_t (t < T), this wave's Ud rows (t = wv, wv + 2, ..; k = lane, lane + 64)
    float hv[4][32];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int t = 0; t < 32; ++t) {
        const int h = j + 128 * q;
        hv[q][t] = (bok && t < T && h < H) ? p.Hs[((size_t)t * Bs + bb) * H + h] : 0.f;
      }
    // scores: thread = (decoder step tt = j % 32, attention columns [32 kq4, 32 kq4 + 32)): the sum over k is mostly inside a
    // thread (32 independent tanh), the four column quarters are added through LDS — no wavefront reductions
    const int tt = j & 31, kq4 = j >> 5;
    float udr[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) {
      const int k = 32 * kq4 + i;
      udr[i] = (bok && tt < T && k < A) ? p.Ud[((size_t)tt * Bs + bb) * A + k] : 0.f;
    }
    if (tid < 128) { swab[2 * tid] = tid < A ? p.w[tid] : 0.f; swab[2 * tid + 1] = tid < A ? p.ab[tid] : 0.f; }
    const uint32_t key = drop_key(p.dd);
    const float invT = 1.0f / (float)T;
    const bool fold = !p.relay && ci == p.NC - 1;          // this workgroup relays the barriers (no dedicated relay workgroup)
    for (int s = 0; s < F; ++s) {
      // ---- Whr_s[b][j] = sum over the unit groups' rank-16 contributions (fixed order; zero at s = 0: hr_{-1} = 0)
      float whr = 0.f;
      if (s >= 1) {
        lc_wait_or_relay(fold, relU, p.bar, p.NU, fb + (unsigned)s, p.bar);
        if (ci == 0) LC_TS(1, s, 0);
        // thread = (attention columns 4 aq .. 4 aq + 3, unit groups gg, gg + 4, ..): every load of the step is issued before
        // the first use (one memory round trip; the blocks were written by other XCDs a moment ago and come from memory)
        const int aq = j & 31, gg = j >> 5;
        lc_h4 v[NQ];
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
          const int g = gg + 4 * q;
          v[q] = (bok && 4 * aq < A && g < p.NG) ? *reinterpret_cast<const lc_h4*>(p.Pw + (((size_t)s * B + b) * p.NG + g) * A + 4 * aq)
                                               : lc_h4{0, 0, 0, 0};
        }
        f32x4 a4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int q = 0; q < NQ; q += 4) {
#pragma unroll
          for (int e = 0; e < 4; ++e) a4[e] += ((float)v[q][e] + (float)v[q + 1][e]) + ((float)v[q + 2][e] + (float)v[q + 3][e]);
        }
        *reinterpret_cast<f32x4*>(spw + ((c * 4 + gg) * 128 + 4 * aq)) = a4;
        __syncthreads();
        whr = (spw[(c * 4 + 0) * 128 + j] + spw[(c * 4 + 1) * 128 + j]) + (spw[(c * 4 + 2) * 128 + j] + spw[(c * 4 + 3) * 128 + j]);
      }
      swh[c * 128 + j] = whr;
      if (bok && j < A) p.Whr[((size_t)s * Bs + b) * A + j] = whr;
      __syncthreads();
      if (ci == 0) LC_TS(1, s, 1);
      {
        float sc0 = 0.f, sc1 = 0.f;
#pragma unroll
        for (int i = 0; i < 32; i += 2) {
          const int k = 32 * kq4 + i;
          const f32x4 wa = *reinterpret_cast<const f32x4*>(swab + 2 * k);          // (w_k, b_k, w_k+1, b_k+1)
          const float h0 = swh[c * 128 + k], h1 = swh[c * 128 + k + 1];
          sc0 += wa[0] * rn_tanh(h0 + wa[1] + udr[i]);
          sc1 += wa[2] * rn_tanh(h1 + wa[3] + udr[i + 1]);
        }
        spb[(c * 4 + kq4) * 32 + tt] = sc0 + sc1;
      }
      __syncthreads();
      if (kq4 == 0) {
        const float sc = (spb[(c * 4) * 32 + tt] + spb[(c * 4 + 1) * 32 + tt]) + (spb[(c * 4 + 2) * 32 + tt] + spb[(c * 4 + 3) * 32 + tt]);
        sbeta[c * 32 + tt] = sc;
        if (bok && tt < T) p.beta[((size_t)s * Bs + b) * T + tt] = sc;
      }
      __syncthreads();
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int h = j + 128 * q;
        float a0 = 0.f, a1 = 0.f;
#pragma unroll
        for (int t = 0; t < 32; t += 2) {       // (h_t is held as zero for t >= T)
          a0 += sbeta[c * 32 + t] * hv[q][t];
          a1 += sbeta[c * 32 + t + 1] * hv[q][t + 1];
        }
        if (h < H) xl[c * 512 + h] = (bf16_t)((a0 + a1) * invT * drop_at(p.dd, key, s, bb, H, h));
      }
      __syncthreads();
      if (ci == 0) LC_TS(1, s, 2);
      // publish x_s[b]: 16 bytes per k-group, written through; then the row-major copy for the deferred dW_ih GEMM
      const int pc = tid >> 6, kg = tid & 63, pb = ci * LC_CPW + pc;
      const bool pon = tid < LC_CPW * 64 && pb < B && kg < (H >> 3);
      if (pon) lc_store16(p.PanX + (size_t)s * pan_x + ((size_t)kg * RC_PAN_ROWS + pb) * 8, xl + pc * 512 + kg * 8);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      lc_arrive(p.bar + p.NU + ci, fb + (unsigned)(s + 1));
      if (ci == 0) LC_TS(1, s, 3);
      if (pon) *reinterpret_cast<bf16x8*>(p.Xcat + ((size_t)s * Bs + pb) * p.ld_xcat + kg * 8) = *reinterpret_cast<const bf16x8*>(xl + pc * 512 + kg * 8);
      if (fold && tid < 64) {      // "every caption workgroup has published x_s": release the unit owners
        unsigned spin = 0;
        lc_poll(p.bar + p.NU, p.NC, fb + (unsigned)(s + 1), p.bar, spin);
        lc_release(relC, fb + (unsigned)(s + 1));
      }
    }
    __syncthreads();
    lc_arrive(p.bar + p.NU + ci, fb + (unsigned)(F + 1));
    if (!p.relay && ci == 0 && tid < 64) {      // no relay workgroup: this one closes the launch (everybody is done and has read the epoch)
      unsigned spin = 0;
      lc_poll(p.bar, p.NU + p.NC, fb + (unsigned)(F + 1), p.bar, spin);
      if (tid == 0) {
        __hip_atomic_store(p.epoch, ep + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); rc_stamp_slot(p.epoch)[1] = wall_clock64();
        if (__hip_atomic_load(p.bar + 257, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) *p.poison = __builtin_nanf("");
      }
    }
    return;
  }

  // ================================================================================== unit-owner workgroups
  // K partials of the four waves COLUMN-major, a thread's cells = consecutive rows of one unit (round 6, as rec_chain_kernel): an
  // accumulator fragment is one ds_write_b128, a thread's partials one 16-byte (RB = 4) or 8-byte (RB = 2) read per gate and wave
  constexpr int RLD = ROWS + 4, RED_W = CG * 16 * RLD;
  float* red = lc_smem;                                                       // [4 waves][64 columns][RLD]
  bf16_t* hl = reinterpret_cast<bf16_t*>(lc_smem + 4 * RED_W);               // [ROWS][UW]
  bf16x8* wih = reinterpret_cast<bf16x8*>(lc_smem + 4 * RED_W + ROWS * UW / 2);   // [4 waves][SX][CG][64] B fragments of W_ih
  // XCD-aware roles (rec_chain.hpp: rc_role): workgroup i runs on XCD i % 8; with two row parts, part = (i % 8) / 4 — the four XCDs of a
  // part fetch its rows of the hr / x panels only (with the unit group as the fast index every XCD pulled both halves every step)
  const bool xmap = p.MS == 2 && (p.NG & 3) == 0;
  const int ug = xmap ? (wg >> 3) * 4 + (wg & 3) : wg % p.NG, part = xmap ? (wg & 7) >> 2 : wg / p.NG;
  const int u0 = ug * UW;
  const int own = RC_PAN_ROWS / p.MS, own_lo = part * own;
  const int r0 = own_lo < RC_PAN_ROWS - ROWS ? own_lo : RC_PAN_ROWS - ROWS;
  const int kw0 = wave * (STEPS * 32);
  const int kq = (lane >> 4) * 8;
  const int rot = ug % NP;
  auto k_of = [&](int pr, int hh) { int prr = pr + rot; prr = prr >= NP ? prr - NP : prr; return kw0 + (prr * 2 + hh) * 32; };

  // ---- residents: W_hh (registers), W_ih (LDS, each lane keeps its own fragments), W_r columns of the own units
  bf16x8 wb[SR][CG];
  // streamed k-steps (SR < STEPS): this wave's fragments in the pre-packed image, this lane's 16 bytes of each
  const bf16_t* wst = (SR < STEPS) ? p.Wst + ((size_t)(ug * 4 + wave) * (STEPS - SR) * CG * 64 + lane) * 8 : nullptr;
#pragma unroll
  for (int g = 0; g < CG; ++g) {
    const int col = g * 16 + (lane & 15), gate = col / UW, ul = col % UW;
    const bf16_t* wrow = p.W + (size_t)(gate * R + u0 + ul) * p.ldw;
#pragma unroll
    for (int pr = 0; pr < NPR; ++pr)
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) {
        const int k = k_of(pr, hh);
        wb[pr * 2 + hh][g] = (k + kq < R) ? *reinterpret_cast<const bf16x8*>(wrow + H + k + kq) : bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
      }
#pragma unroll
    for (int ks = 0; ks < SX; ++ks) {
      const int k = (wave * SX + ks) * 32;
      wih[((wave * SX + ks) * CG + g) * 64 + lane] = (k + kq < H) ? *reinterpret_cast<const bf16x8*>(wrow + k + kq) : bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
    }
  }
  // W_r[a][u0 + (lane / 16) * 4 .. + 4] for a = (2 wave + q) * 16 + lane % 16: B operand of the 16x16x16 MFMA
  bf16x4 wr[2];
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int a = (2 * wave + q) * 16 + (lane & 15);
    wr[q] = a < A ? *reinterpret_cast<const bf16x4*>(p.Wr + (size_t)a * p.ldwr + u0 + (lane >> 4) * 4) : bf16x4{0, 0, 0, 0};
  }
  constexpr int CPT = (ROWS * UW + 255) / 256;
  static_assert(CPT == 4 || CPT == 2, "RB = 4 or 2");
#define LCU_CELL(c) (((tid / UW) * CPT + (c)) * UW + tid % UW)
  float xb[CPT][4], cpv[CPT];
  bool mine[CPT];
#pragma unroll
  for (int c = 0; c < CPT; ++c) {
    const int cell = LCU_CELL(c), rg = r0 + cell / UW;
    mine[c] = cell < ROWS * UW && rg >= own_lo && rg < own_lo + own && rg < B;
    cpv[c] = 0.f;
#pragma unroll
    for (int q = 0; q < 4; ++q) xb[c][q] = p.bias[q * R + u0 + cell % UW];
  }
  const int lane_off = ((lane >> 4) * RC_PAN_ROWS + r0 + (lane & 15)) * 8;
  f32x4 acc[RB][CG];
#pragma unroll
  for (int i = 0; i < RB; ++i)
#pragma unroll
    for (int g = 0; g < CG; ++g) acc[i][g] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int it_j = tid / own, it_rg = own_lo + tid % own;              // publish item = (k-group of this workgroup, owned row)
  const bool it_on = tid < KG * own && it_rg < B && it_rg - r0 < ROWS;
  const bf16_t* it_src = hl + (it_on ? it_rg - r0 : 0) * UW + (it_on ? it_j : 0) * 8;

  for (int s = 0; s < F; ++s) {
    // ---- x_s . W_ih^T on top of the recurrent part (this wave's K slice of both)
    lc_wait(relC, fb + (unsigned)(s + 1), p.bar);
    if (wg == 0) LC_TS(0, s, 0);
    {
      const bf16_t* Ax = p.PanX + (size_t)s * pan_x + lane_off;
      bf16x8 fx[SX][RB];
#pragma unroll
      for (int ks = 0; ks < SX; ++ks) {
        const int k = (wave * SX + ks) * 32;
#pragma unroll
        for (int i = 0; i < RB; ++i) fx[ks][i] = *reinterpret_cast<const bf16x8*>(Ax + ((k < H ? (k >> 3) : 0) * RC_PAN_ROWS + i * 16) * 8);
      }
#pragma unroll
      for (int ks = 0; ks < SX; ++ks)
#pragma unroll
        for (int g = 0; g < CG; ++g) {
          const bf16x8 bw = wih[((wave * SX + ks) * CG + g) * 64 + lane];
#pragma unroll
          for (int i = 0; i < RB; ++i) acc[i][g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fx[ks][i], bw, acc[i][g], 0, 0, 0);
        }
    }
    {
      float* prt = red + wave * RED_W;
      const int rr = (lane >> 4) * 4, cc = lane & 15;
#pragma unroll
      for (int i = 0; i < RB; ++i)
#pragma unroll
        for (int g = 0; g < CG; ++g) {
          *reinterpret_cast<f32x4*>(prt + (g * 16 + cc) * RLD + i * 16 + rr) = acc[i][g];
          acc[i][g] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }
    __syncthreads();
    if (wg == 0) LC_TS(0, s, 1);
    // ---- cell pointwise for UW units x owned rows
    float gsum[4][CPT];
    {
      const float* rp = red + (tid % UW) * RLD + (tid / UW) * CPT;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
#pragma unroll
        for (int c = 0; c < CPT; ++c) gsum[q][c] = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
          if constexpr (CPT == 4) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(rp + w * RED_W + q * UW * RLD);
            gsum[q][0] += v[0]; gsum[q][1] += v[1]; gsum[q][2] += v[2]; gsum[q][3] += v[3];
          } else {
            const float2 v = *reinterpret_cast<const float2*>(rp + w * RED_W + q * UW * RLD);
            gsum[q][0] += v.x; gsum[q][1] += v.y;
          }
        }
      }
    }
    float hv[CPT], av[CPT][4];
#pragma unroll
    for (int c = 0; c < CPT; ++c) {
      const int cell = LCU_CELL(c);
      float g4[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) g4[q] = xb[c][q] + gsum[q][c];
      if (p.gru) {
        const GruOut r = gru_point(g4[0], g4[1], g4[2], g4[3], cpv[c]);
        hv[c] = r.h; av[c][0] = r.r; av[c][1] = r.z; av[c][2] = r.n; av[c][3] = r.hn; cpv[c] = r.h;
      } else {
        const LstmOut r = lstm_point(g4[0], g4[1], g4[2], g4[3], cpv[c]);
        hv[c] = r.h; av[c][0] = r.i; av[c][1] = r.f; av[c][2] = r.g; av[c][3] = r.o; cpv[c] = r.c;
      }
      if (cell < ROWS * UW) hl[cell] = (bf16_t)hv[c];
    }
    __syncthreads();
    if (wg == 0) LC_TS(0, s, 2);
    const bool more = s + 1 < F;
    if (more) {
      // publish hr_s (what the recurrent product of the next step reads)
      if (it_on) lc_store16(p.PanH + (size_t)s * pan_h + ((size_t)(ug * KG + it_j) * RC_PAN_ROWS + it_rg) * 8, it_src);
      // this workgroup's rank-16 contribution to Whr_{s+1}, transposed: W_r[:, own units] . [ROWS x 16 units]^T as 16x16x16
      // MFMAs (wave -> attention columns [32 wave, 32 wave + 32)): a lane then holds four consecutive attention columns of
      // one caption - one 8-byte fp16 store straight from the accumulator, no staging through LDS
      {
        _Float16* Pn = p.Pw + ((size_t)(s + 1) * B * p.NG + ug) * A;
#pragma unroll
        for (int i = 0; i < RB; ++i) {
          const bf16x4 fa = *reinterpret_cast<const bf16x4*>(hl + (i * 16 + (lane & 15)) * UW + (lane >> 4) * 4);
          const int rg = r0 + i * 16 + (lane & 15);
          const bool ron = rg >= own_lo && rg < own_lo + own && rg < B;
#pragma unroll
          for (int q = 0; q < 2; ++q) {
            const int a0 = (2 * wave + q) * 16 + (lane >> 4) * 4;
            if ((2 * wave + q) * 16 < A) {
              const f32x4 d = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(lc_s4, wr[q]), __builtin_bit_cast(lc_s4, fa),
                                                                        f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
              if (ron && a0 < A) {
                union { lc_h4 h; uint64_t u; } pk;
                pk.h[0] = (_Float16)d[0]; pk.h[1] = (_Float16)d[1]; pk.h[2] = (_Float16)d[2]; pk.h[3] = (_Float16)d[3];
                __hip_atomic_store(reinterpret_cast<uint64_t*>(Pn + (size_t)rg * p.NG * A + a0), pk.u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              }
            }
          }
        }
      }
      if (wg == 0) LC_TS(0, s, 3);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      lc_arrive(p.bar + wg, fb + (unsigned)(s + 1));
      if (wg == 0) LC_TS(0, s, 4);
    }
    // ---- off the critical path: what the backward and the output layer read
    float* Ht = p.Hr + (size_t)s * Bs * R;
    float* Ct = p.Cr + (size_t)s * Bs * R;
    float* At = p.acts + (size_t)s * Bs * 4 * R;
    bf16_t* Lt = p.Hlp + (size_t)s * Bs * p.ld_hlp;
    if (it_on) *reinterpret_cast<bf16x8*>(Lt + (size_t)it_rg * p.ld_hlp + u0 + it_j * 8) = *reinterpret_cast<const bf16x8*>(it_src);
#pragma unroll
    for (int c = 0; c < CPT; ++c) {
      const int cell = LCU_CELL(c);
      if (mine[c]) {
        const int row = r0 + cell / UW, u = u0 + cell % UW;
        const size_t o = (size_t)row * R + u;
        Ht[o] = hv[c];
        if (!p.gru) Ct[o] = cpv[c];
        float* a = At + (size_t)row * 4 * R + u;
        a[0] = av[c][0]; a[R] = av[c][1]; a[2 * R] = av[c][2]; a[3 * R] = av[c][3];
      }
    }
    if (wg == 0 && p.ld_hlp > R)
      for (int jj = tid; jj < B * (p.ld_hlp - R); jj += 256) Lt[(size_t)(jj / (p.ld_hlp - R)) * p.ld_hlp + R + jj % (p.ld_hlp - R)] = (bf16_t)0.f;
    if (more) {
      // ---- the recurrent product of the NEXT step, hr_s . W_hh^T, as soon as every workgroup has published hr_s; the
      // attention of step s + 1 runs in the caption workgroups meanwhile
      // streamed weight fragments (SR < STEPS): ring slot = pair mod PF, like the activations.  Pairs are consumed in the order
      // streamed first, resident last, so that the resident MFMAs cover the tail of the stream; the first PF streamed pairs
      // are requested BEFORE the wait (their addresses do not depend on the chain).
      bf16x8 fw[PF][2][CG];
      // (the row pointers are laundered through an empty asm every step: as loop invariants the compiler materialised all
      // 4 (STEPS - SR) 64-bit addresses ahead of the time loop — 128 registers — and spilled them)
      const LC_GLOBAL bf16_t* wsl = (const LC_GLOBAL bf16_t*)wst;
      if constexpr (SR < STEPS) wsl = lc_launder_v(wst);
      auto issue_w = [&](int slot, int i) {        // i = streamed pair, in consumption order (= the image's order)
#pragma unroll
        for (int hh = 0; hh < 2; ++hh)
#pragma unroll
          for (int g = 0; g < CG; ++g) fw[slot][hh][g] = *(lc_gbf16x8)(wsl + (size_t)(((i * 2 + hh) * CG + g) * 512));
      };
      constexpr int NPS = NP - NPR;                       // streamed pairs; iteration i < NPS -> pair NPR + i, then the resident pairs
      auto pair_of = [&](int i) { return i < NPS ? NPR + i : i - NPS; };
      if constexpr (NPS > 0) {
#pragma unroll
        for (int i = 0; i < PF; ++i)
          if (i < NPS) issue_w(i % PF, i);
      }
      lc_wait(relU, fb + (unsigned)(s + 1), p.bar);
      if (wg == 0) LC_TS(0, s, 5);
      const bf16_t* Ah = p.PanH + (size_t)s * pan_h + lane_off;
      bf16x8 fa[PF][2][RB];
      auto issue_pair = [&](int slot, int pr) {
#pragma unroll
        for (int hh = 0; hh < 2; ++hh)
#pragma unroll
          for (int i = 0; i < RB; ++i) {
            const int k = k_of(pr, hh);
            const int kg = k < R ? (k >> 3) : 0;      // (k-steps beyond R meet zero weights; wave-uniform)
            fa[slot][hh][i] = *reinterpret_cast<const bf16x8*>(Ah + (kg * RC_PAN_ROWS + i * 16) * 8);
          }
      };
#pragma unroll
      for (int i = 0; i < PF; ++i)
        if (i < NP) issue_pair(i, pair_of(i));
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int it = 0; it < NP; ++it) {
        const int slot = it % PF;
        const int pr = pair_of(it);
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
#pragma unroll
          for (int i = 0; i < RB; ++i)
#pragma unroll
            for (int g = 0; g < CG; ++g) {
              const bf16x8 w = it < NPS ? fw[slot][hh][g] : wb[(it < NPS ? 0 : pr * 2 + hh)][g];
              acc[i][g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[slot][hh][i], w, acc[i][g], 0, 0, 0);
            }
        }
        if (it + PF < NP) {
          __builtin_amdgcn_sched_barrier(0);
          issue_pair(slot, pair_of(it + PF));
          if (it + PF < NPS) issue_w(slot, it + PF);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      if (wg == 0) LC_TS(0, s, 6);
    }
  }
  __syncthreads();
  lc_arrive(p.bar + wg, fb + (unsigned)(F + 1));
}
// The streamed k-steps of the hybrid forward chain as MFMA B fragments, in the order loc_chain_kernel<STEPS, ., ., SR> consumes
// them (pair NPR + i of its rotated sequence, i = 0 ..): dst[ug][wave][js][g][lane][8].  Run after every update of W_hh.
#undef LCU_CELL
__global__ __launch_bounds__(256) void lc_pack_stream_kernel(const bf16_t* __restrict__ W, int ldw, int R, int H, int STEPS, int SR, bf16_t* __restrict__ dst, size_t n_frag) {
  const size_t f = (size_t)blockIdx.x * 256 + threadIdx.x;       // one 16-byte fragment piece per thread
  if (f >= n_frag) return;
  const int NS = STEPS - SR, NP = STEPS / 2, NPR = SR / 2;
  const int lane = (int)(f & 63), g = (int)((f >> 6) & 3);
  const size_t r = f >> 8;
  const int js = (int)(r % NS), wave = (int)((r / NS) & 3), ug = (int)(r / NS / 4);
  const int col = g * 16 + (lane & 15), gate = col / 16, ul = col % 16, kq = (lane >> 4) * 8;
  int prr = NPR + js / 2 + ug % NP; prr = prr >= NP ? prr - NP : prr;
  const int k = wave * (STEPS * 32) + (prr * 2 + (js & 1)) * 32;
  const bf16_t* src = W + (size_t)(gate * R + ug * 16 + ul) * ldw + H + k + kq;
  bf16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
  if (k + kq < R) v = *reinterpret_cast<const bf16x8*>(src);
  *reinterpret_cast<bf16x8*>(dst + f * 8) = v;
}
template <int RB> constexpr size_t lc_smem_bytes() {
  return (size_t)4 * 64 * (RB * 16 + 4) * 4 + (size_t)RB * 16 * 16 * 2 + (size_t)4 * 4 * 4 * 64 * 16;
}

// =============================================================================================
// The backward chain of the local reconstructor, one launch (the mirror of loc_chain_kernel; chain step q <-> s = F-1-q):
//   dhr_s = dHr[s] + dG_{s+1} . W_hh + dWhr_{s+1} . W_r ;  (dG_s, dc) = cell backward
//   dx_s  = dropmask(s) * (dG_s . W_ih) ;  attention backward of step s: dbeta, dHs +=, dUd +=, dw +=, dWhr_s
// Three kinds of resident workgroups and three relayed hand-overs per step:
//   U' (16 output units of dhr, all rows or one row part): its rows of [W_ih | W_hh]^T (K = 4R contiguous, image WT) in
//      registers; per step the recurrent product over the gate-gradient panel of the step before (off the critical path:
//      it runs while X' and C' work), then the small product with dWhr (K = A), cell backward, publishes dG_s;
//   X' (16 columns of dx x one row part): the same product with the W_ih rows of WT; publishes dx_s (fp32, masked);
//   C' (two captions): h_t, Ud[t] and the dHs / dUd / dw accumulators live in registers for all F steps; publishes dWhr_s.
// The panels are indexed by chain step (fresh addresses every step, rec_chain.hpp).
struct LocChainBwdArgs {
  int F, T, B, R, H, A, gru;
  int Bs;                            // see LocChainArgs (Dx / panels are private to the launch)
  int NGU, MSU, NGX, MSX, NC;        // U' = NGU unit groups x MSU row parts, X' = NGX column groups x MSX row parts, C'
  int KSX;                           // X' parts of the contraction (1: the whole of K = 4R per workgroup; > 1: lcb_xsplit_role)
  const bf16_t* WT; int ldwt;        // [H + R][ldwt]: rows [0,H) = W_ih^T (x columns), rows [H, H+R) = W_hh^T; K = 4R contiguous
  const bf16_t* Wr; int ldwr;        // [A][ldwr]
  const float* dHr;                  // [F][B][R] d loss / d hr_s through the output layer
  const float* acts; const float* Cr; const float* Hr;     // saved by the forward
  const float* Hs; const float* Ud; const float* ab; const float* w;
  const float* Whr; const float* beta;                       // [F][B][A], [F][B][T]
  bf16_t* dG; int ld_dg;             // [F][B][ld_dg] row-major gate gradients (deferred weight-gradient GEMMs)
  bf16_t* dWhrs; int ld_dwhr;        // [F][B][ld_dwhr] row-major dWhr_s (deferred attn_W / attn_b gradients)
  float* dHs;                        // [T][B][H] out: d loss / d decoder states (attention path)
  float* dUd; bf16_t* dUd_lp; int ld_dUd;   // [T][B][A] (+ operand copy, zero padded)
  float* dwacc;                      // [nch][B][A]: chunk 0 = sum_t dbeta tanh(.), other chunks zero
  int nch;
  bf16_t* PanG; bf16_t* PanW; float* Dx;    // exchange: [F][rc_pan_elems(4R)], [F][rc_pan_elems(A)], [F][KSX][B][H]
  unsigned* bar; unsigned* epoch; float* poison;
  DropDesc dd;
  int defer_big;                     // 1: U' starts its recurrent product after X' has finished with the same panel (L2 bandwidth)
  unsigned long long* ts;            // probe builds only
};

// acc[i] += rows [r0 + 16 i, +16) of the panel (K-group-major, all of K) . this lane's resident weight fragments
template <int STEPS, int PF, int RB>
__device__ __forceinline__ void lcb_product(f32x4 (&acc)[RB], const bf16x8 (&wb)[STEPS], const bf16_t* A, int K, int kw0, int rot) {
  constexpr int NP = STEPS / 2;
  auto k_of = [&](int pr, int hh) { int prr = pr + rot; prr = prr >= NP ? prr - NP : prr; return kw0 + (prr * 2 + hh) * 32; };
  bf16x8 fa[PF][2][RB];
  auto issue_pair = [&](int slot, int pr) {
#pragma unroll
    for (int hh = 0; hh < 2; ++hh)
#pragma unroll
      for (int i = 0; i < RB; ++i) {
        const int k = k_of(pr, hh);
        fa[slot][hh][i] = *reinterpret_cast<const bf16x8*>(A + ((k < K ? (k >> 3) : 0) * RC_PAN_ROWS + i * 16) * 8);
      }
  };
#pragma unroll
  for (int pr = 0; pr < PF; ++pr)
    if (pr < NP) issue_pair(pr, pr);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int pr = 0; pr < NP; ++pr) {
    const int slot = pr % PF;
#pragma unroll
    for (int hh = 0; hh < 2; ++hh)
#pragma unroll
      for (int i = 0; i < RB; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[slot][hh][i], wb[pr * 2 + hh], acc[i], 0, 0, 0);
    if (pr + PF < NP) {
      __builtin_amdgcn_sched_barrier(0);
      issue_pair(slot, pr + PF);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
}
template <int STEPS>
__device__ __forceinline__ void lcb_load_weights(bf16x8 (&wb)[STEPS], const bf16_t* wrow, int K, int kw0, int rot, int kq) {
  constexpr int NP = STEPS / 2;
#pragma unroll
  for (int pr = 0; pr < NP; ++pr)
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
      int prr = pr + rot; prr = prr >= NP ? prr - NP : prr;
      const int k = kw0 + (prr * 2 + hh) * 32;
      wb[pr * 2 + hh] = (k + kq < K) ? *reinterpret_cast<const bf16x8*>(wrow + k + kq) : bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
    }
}

// X' with the contraction split over workgroups (XS k-steps of 32 per wave, XS * 128 of K per workgroup): a workgroup owns
// 64 columns of dx x 32 rows x one K part, its 64 x XS * 128 weights in registers, and reads 32 rows x XS * 128 of the
// gate-gradient panel per step instead of 64 rows x 4R - the panel goes through the CU's 64 B/clk L1 fill path, which is
// what bounds the whole-K form.  The KSX partial dx (each masked: the dropout mask is a per-element factor) are summed by
// the caption workgroups when they read them.
template <int XS, int PF>
__device__ __forceinline__ void lcb_xsplit_role(const LocChainBwdArgs& p, float* red, int xi, int wg, unsigned fb, const unsigned* relG) {
  constexpr int CG = 4, RB = 2, UWX = 16 * CG, ROWS = RB * 16, RED_LD = UWX + 1, NP = XS / 2, OWN = ROWS;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int R = p.R, B = p.B, H = p.H, F = p.F, K = 4 * R;
  // XCD-aware roles: X' workgroup xi runs on XCD (NU + xi) % 8.  With four row parts a part lives on two XCDs, which share its
  // NGX KSX tiles: each then fetches a quarter of the rows and about half of K of the gate-gradient panel (1.4 MB at R = 1536)
  // instead of all of it, every step
  const int NUx = p.NGU * p.MSU, per2 = p.NGX * p.KSX;
  const bool xmap = p.MSX == 4 && (NUx & 7) == 0 && (per2 & 1) == 0 && p.NGX * p.MSX * p.KSX == 4 * per2;
  const int xidx = xmap ? (xi & 1) * (per2 >> 1) + (xi >> 3) : 0;
  const int xg = xmap ? xidx % p.NGX : xi % p.NGX, rest = xi / p.NGX, part = xmap ? (xi & 7) >> 1 : rest % p.MSX, kp = xmap ? xidx / p.NGX : rest / p.MSX;
  const int j0 = xg * UWX, own_lo = part * OWN;
  const int r0 = own_lo < RC_PAN_ROWS - ROWS ? own_lo : RC_PAN_ROWS - ROWS;
  const int kq = (lane >> 4) * 8, kw0 = kp * (XS * 128) + wave * (XS * 32);
  const int rot = xg % NP;
  auto k_of = [&](int pr, int hh) { int prr = pr + rot; prr = prr >= NP ? prr - NP : prr; return kw0 + (prr * 2 + hh) * 32; };
  bf16x8 wb[XS][CG];
#pragma unroll
  for (int g = 0; g < CG; ++g) {
    const int jr = j0 + g * 16 + (lane & 15);
    const bf16_t* wrow = p.WT + (size_t)(jr < H ? jr : 0) * p.ldwt + kq;
    const int Kg = jr < H ? K : 0;
#pragma unroll
    for (int pr = 0; pr < NP; ++pr)
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) {
        const int k = k_of(pr, hh);
        wb[pr * 2 + hh][g] = (k + kq < Kg) ? *reinterpret_cast<const bf16x8*>(wrow + k) : bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
      }
  }
  const int lane_off = ((lane >> 4) * RC_PAN_ROWS + r0 + (lane & 15)) * 8;
  const size_t pan_g = rc_pan_elems(K);
  const uint32_t key = drop_key(p.dd);
  for (int q = 0; q < F; ++q) {
    const int s = F - 1 - q;
    lc_wait(relG, fb + (unsigned)(q + 1), p.bar);
    if (xi == 0) LC_TS(5, q, 0);
    const bf16_t* A = p.PanG + (size_t)q * pan_g + lane_off;
    bf16x8 fa[PF][2][RB];
    auto issue_pair = [&](int slot, int pr) {
#pragma unroll
      for (int hh = 0; hh < 2; ++hh)
#pragma unroll
        for (int i = 0; i < RB; ++i) {
          const int k = k_of(pr, hh);
          fa[slot][hh][i] = *reinterpret_cast<const bf16x8*>(A + ((k < K ? (k >> 3) : 0) * RC_PAN_ROWS + i * 16) * 8);
        }
    };
#pragma unroll
    for (int pr = 0; pr < PF; ++pr)
      if (pr < NP) issue_pair(pr, pr);
    __builtin_amdgcn_sched_barrier(0);
    f32x4 acc[RB][CG];
#pragma unroll
    for (int i = 0; i < RB; ++i)
#pragma unroll
      for (int g = 0; g < CG; ++g) acc[i][g] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int pr = 0; pr < NP; ++pr) {
      const int slot = pr % PF;
#pragma unroll
      for (int hh = 0; hh < 2; ++hh)
#pragma unroll
        for (int i = 0; i < RB; ++i)
#pragma unroll
          for (int g = 0; g < CG; ++g) acc[i][g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[slot][hh][i], wb[pr * 2 + hh][g], acc[i][g], 0, 0, 0);
      if (pr + PF < NP) {
        __builtin_amdgcn_sched_barrier(0);
        issue_pair(slot, pr + PF);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    float* prt = red + wave * (ROWS * RED_LD);
#pragma unroll
    for (int i = 0; i < RB; ++i)
#pragma unroll
      for (int g = 0; g < CG; ++g)
#pragma unroll
        for (int r = 0; r < 4; ++r) prt[(i * 16 + (lane >> 4) * 4 + r) * RED_LD + g * 16 + (lane & 15)] = acc[i][g][r];
    __syncthreads();
    if (xi == 0) LC_TS(5, q, 1);
    float* Dq = p.Dx + ((size_t)q * p.KSX + kp) * B * H;
    // four columns per thread, one 16-byte write-through store (round 3 shipped two columns / 8-byte stores after a 4-column form
    // aborted the forced-split small-shape tests; round 4 rebuilt it with every index checked — see LCB_CHECK below and DESIGN.md)
    for (int idx = tid; idx < OWN * (UWX / 4); idx += 256) {
      const int rg = own_lo + idx / (UWX / 4), pc = (idx % (UWX / 4)) * 4, rl = rg - r0;
      if (rg < B && j0 + pc < H) {
        float* dst = Dq + (size_t)rg * H + j0 + pc;
        // LCB_CHECK: row of the reduction buffer inside the tile, the four columns inside H, a 16-byte-aligned address inside this
        // step's [KSX][B][H] block — a violation raises the chain's sticky word (the step reports NaN) instead of writing
        const bool ok = rl >= 0 && rl < ROWS && j0 + pc + 3 < H && (((uintptr_t)dst) & 15) == 0 &&
                        ((size_t)q * p.KSX + kp) * B * H + (size_t)rg * H + j0 + pc + 3 < (size_t)F * p.KSX * B * H;
        if (!ok) { __hip_atomic_store(p.bar + 257, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); continue; }
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int w = 0; w < 4; ++w)
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] += red[w * (ROWS * RED_LD) + rl * RED_LD + pc + e];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] *= drop_at(p.dd, key, s, rg, H, j0 + pc + e);
        rc_store16f(dst, v);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    lc_arrive(p.bar + wg, fb + (unsigned)(q + 1));
    if (xi == 0) LC_TS(5, q, 2);
  }
  __syncthreads();
  lc_arrive(p.bar + wg, fb + (unsigned)(F + 1));
}

template <int STEPS, int PF, int RBU, int RBX, int XS = 0>
__global__ __launch_bounds__(256) void loc_chain_bwd_kernel(const LocChainBwdArgs p) {
  constexpr int UW = 16, RED_LD = UW + 1, KG = UW / 8;
  extern __shared__ __attribute__((aligned(16))) float lc_smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wg = blockIdx.x, R = p.R, B = p.B, Bs = p.Bs, H = p.H, A = p.A, F = p.F, T = p.T, K = 4 * R;
  const int NU = p.NGU * p.MSU, NX = p.NGX * p.MSX * p.KSX;
  const unsigned ep = rc_epoch_read(p.epoch), fb = ep << 7;
  unsigned* relG = p.bar + 256; unsigned* relX = p.bar + 256 + 8; unsigned* relW = p.bar + 256 + 16;
  const size_t pan_g = rc_pan_elems(K), pan_w = rc_pan_elems(A);
  const int kq = (lane >> 4) * 8, kw0 = wave * (STEPS * 32);

  // ================================================================================== relay workgroup
  if (wg == NU + NX + p.NC) {
    if (tid < 64) {
      unsigned spin = 0;
      for (int q = 0; q < F; ++q) {
        lc_poll(p.bar, NU, fb + (unsigned)(q + 1), p.bar, spin); lc_release(relG, fb + (unsigned)(q + 1));
        LC_TS(3, q, 0);
        lc_poll(p.bar + NU, NX, fb + (unsigned)(q + 1), p.bar, spin); lc_release(relX, fb + (unsigned)(q + 1));
        LC_TS(3, q, 1);
        lc_poll(p.bar + NU + NX, p.NC, fb + (unsigned)(q + 1), p.bar, spin); lc_release(relW, fb + (unsigned)(q + 1));
        LC_TS(3, q, 2);
      }
      lc_poll(p.bar, NU + NX + p.NC, fb + (unsigned)(F + 1), p.bar, spin);
      if (tid == 0) {
        __hip_atomic_store(p.epoch, ep + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); rc_stamp_slot(p.epoch)[1] = wall_clock64();
        if (__hip_atomic_load(p.bar + 257, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) *p.poison = __builtin_nanf("");
      }
    }
    return;
  }

  // ================================================================================== caption workgroups
  if (wg >= NU + NX) {
    float* sdx = lc_smem;                                   // [LC_CPW][512] dx_s
    float* spd = sdx + LC_CPW * 512;                        // [LC_CPW][4][32] partial dbeta
    float* sdbt = spd + LC_CPW * 128;                       // [LC_CPW][32] dbeta
    bf16_t* swl = reinterpret_cast<bf16_t*>(sdbt + LC_CPW * 32);    // [LC_CPW][128] dWhr_s (bf16)
    const int ci = wg - NU - NX, c = tid >> 7, j = tid & 127;
    const int b = ci * LC_CPW + c;
    const bool bok = b < B;
    const int bb = bok ? b : 0;
    const bool kon = bok && j < A;
    // thread = (decoder step tt, quarter hq of the hidden columns): its 128 values of h_tt and of the dHs accumulator;
    // thread = attention column k = j for the (t, k) plane: Ud[t][k] and the dUd accumulator for every t
    const int tt = j & 31, hq = j >> 5;
    float hv[128], dhs[128], ud[32], dud[32];
#pragma unroll
    for (int i = 0; i < 128; ++i) {
      const int h = 128 * hq + i;
      hv[i] = (bok && tt < T && h < H) ? p.Hs[((size_t)tt * Bs + bb) * H + h] : 0.f;
      dhs[i] = 0.f;
    }
#pragma unroll
    for (int t = 0; t < 32; ++t) {
      ud[t] = (kon && t < T) ? p.Ud[((size_t)t * Bs + bb) * A + j] : 0.f;
      dud[t] = 0.f;
    }
    const float wk = kon ? p.w[j] : 0.f, abk = kon ? p.ab[j] : 0.f;
    float dwa = 0.f;
    const float invT = 1.0f / (float)T;
    for (int q = 0; q < F; ++q) {
      const int s = F - 1 - q;
      // saved tensors of step s: independent of the chain, requested before waiting
      const float whk = (kon ? p.Whr[((size_t)s * Bs + b) * A + j] : 0.f) + abk;
      const float bt = (bok && tt < T) ? p.beta[((size_t)s * Bs + b) * T + tt] * invT : 0.f;
      // tanh(W hr_s + U h_t + b) of the (t, k) plane: saved operands only, so it is formed while this workgroup waits for dx_s
      float tzr[32];
#pragma unroll
      for (int t = 0; t < 32; ++t) tzr[t] = rn_tanh(whk + ud[t]);
      lc_wait(relX, fb + (unsigned)(q + 1), p.bar);
      if (ci == 0) LC_TS(6, q, 0);
      {   // the KSX <= 4 partial dx of the X' workgroups: every load in flight before the first add
        float pv[4][4];
#pragma unroll
        for (int qq = 0; qq < 4; ++qq)
#pragma unroll
          for (int kp = 0; kp < 4; ++kp) {
            const int h = j + 128 * qq;
            pv[qq][kp] = (bok && h < H && kp < p.KSX) ? p.Dx[(((size_t)q * p.KSX + kp) * B + b) * H + h] : 0.f;
          }
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) sdx[c * 512 + j + 128 * qq] = (pv[qq][0] + pv[qq][1]) + (pv[qq][2] + pv[qq][3]);
      }
      __syncthreads();
      // dbeta[tt] = (1/T) dx . h_tt (this thread's quarter) and dHs[tt] += (beta_s[tt] / T) dx
      {
        float v0 = 0.f, v1 = 0.f;
#pragma unroll
        for (int i = 0; i < 128; i += 4) {
          const f32x4 d4 = *reinterpret_cast<const f32x4*>(sdx + c * 512 + 128 * hq + i);
          v0 += d4[0] * hv[i] + d4[2] * hv[i + 2];
          v1 += d4[1] * hv[i + 1] + d4[3] * hv[i + 3];
          dhs[i] += bt * d4[0]; dhs[i + 1] += bt * d4[1]; dhs[i + 2] += bt * d4[2]; dhs[i + 3] += bt * d4[3];
        }
        spd[(c * 4 + hq) * 32 + tt] = v0 + v1;
      }
      __syncthreads();
      if (hq == 0) sdbt[c * 32 + tt] = ((spd[(c * 4) * 32 + tt] + spd[(c * 4 + 1) * 32 + tt]) + (spd[(c * 4 + 2) * 32 + tt] + spd[(c * 4 + 3) * 32 + tt])) * invT;
      __syncthreads();
      // (t, k) plane, thread = k: dz = dbeta[t] w_k (1 - tanh^2), dUd[t] += dz, dWhr_s = sum_t dz, dw += sum_t dbeta[t] tanh
      float dwh = 0.f;
#pragma unroll
      for (int t = 0; t < 32; ++t) {
        if (t < T) {
          const float tz = tzr[t];
          const float db = sdbt[c * 32 + t];
          const float dz = db * wk * (1.f - tz * tz);
          dwa += db * tz;
          dwh += dz;
          dud[t] += dz;
        }
      }
      swl[c * 128 + j] = (bf16_t)(kon ? dwh : 0.f);
      __syncthreads();
      if (ci == 0) LC_TS(6, q, 1);
      // publish dWhr_s[b]: 16 bytes per k-group, written through; then the row-major copy (zero padded)
      const int pc = tid >> 6, kg = tid & 63, pb = ci * LC_CPW + pc;
      const bool pon = tid < LC_CPW * 64 && pb < B;
      // (k-groups up to the next multiple of 32 columns: the consumer's k-step reads them; swl holds zeros beyond A)
      if (pon && kg < (((A + 31) >> 5) << 2)) lc_store16(p.PanW + (size_t)q * pan_w + ((size_t)kg * RC_PAN_ROWS + pb) * 8, swl + pc * 128 + kg * 8);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      lc_arrive(p.bar + NU + NX + ci, fb + (unsigned)(q + 1));
      if (ci == 0) LC_TS(6, q, 2);
      if (pon && kg < (p.ld_dwhr >> 3) && kg < 16)
        *reinterpret_cast<bf16x8*>(p.dWhrs + ((size_t)s * Bs + pb) * p.ld_dwhr + kg * 8) = *reinterpret_cast<const bf16x8*>(swl + pc * 128 + kg * 8);
    }
    // ---- the accumulators
    if (bok) {
      if (tt < T) {
#pragma unroll
        for (int i = 0; i < 128; i += 4) {
          const int h = 128 * hq + i;
          if (h < H) *reinterpret_cast<f32x4*>(p.dHs + ((size_t)tt * Bs + b) * H + h) = f32x4{dhs[i], dhs[i + 1], dhs[i + 2], dhs[i + 3]};
        }
      }
#pragma unroll
      for (int t = 0; t < 32; ++t) {
        if (t < T) {
          if (j < A) { p.dUd[((size_t)t * Bs + b) * A + j] = dud[t]; p.dUd_lp[((size_t)t * Bs + b) * p.ld_dUd + j] = (bf16_t)dud[t]; }
          else if (j < p.ld_dUd) p.dUd_lp[((size_t)t * Bs + b) * p.ld_dUd + j] = (bf16_t)0.f;
        }
      }
      if (j < A) {
        p.dwacc[(size_t)b * A + j] = dwa;
        for (int ch = 1; ch < p.nch; ++ch) p.dwacc[((size_t)ch * Bs + b) * A + j] = 0.f;
      }
    }
    __syncthreads();
    lc_arrive(p.bar + NU + NX + ci, fb + (unsigned)(F + 1));
    return;
  }

  // ================================================================================== X': dx_s = mask * dG_s . W_ih
  if (wg >= NU) {
    if constexpr (XS > 0) { lcb_xsplit_role<XS, 3>(p, lc_smem, wg - NU, wg, fb, relG); return; }
    constexpr int ROWS = RBX * 16;
    float* red = lc_smem;                                   // [4 waves][ROWS][RED_LD]
    const int xi = wg - NU, xg = xi % p.NGX, part = xi / p.NGX;
    const int j0 = xg * UW;
    const int own = RC_PAN_ROWS / p.MSX, own_lo = part * own;
    const int r0 = own_lo < RC_PAN_ROWS - ROWS ? own_lo : RC_PAN_ROWS - ROWS;
    const int rot = xg % (STEPS / 2);
    bf16x8 wb[STEPS];
    {
      const int jr = j0 + (lane & 15);
      lcb_load_weights<STEPS>(wb, p.WT + (size_t)(jr < H ? jr : 0) * p.ldwt, jr < H ? K : 0, kw0, rot, kq);
    }
    const int lane_off = ((lane >> 4) * RC_PAN_ROWS + r0 + (lane & 15)) * 8;
    const uint32_t key = drop_key(p.dd);
    for (int q = 0; q < F; ++q) {
      const int s = F - 1 - q;
      lc_wait(relG, fb + (unsigned)(q + 1), p.bar);
      if (xi == 0) LC_TS(5, q, 0);
      f32x4 acc[RBX];
#pragma unroll
      for (int i = 0; i < RBX; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
      lcb_product<STEPS, PF, RBX>(acc, wb, p.PanG + (size_t)q * pan_g + lane_off, K, kw0, rot);
      float* prt = red + wave * (ROWS * RED_LD);
#pragma unroll
      for (int i = 0; i < RBX; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) prt[(i * 16 + (lane >> 4) * 4 + r) * RED_LD + (lane & 15)] = acc[i][r];
      __syncthreads();
      if (xi == 0) LC_TS(5, q, 1);
      for (int idx = tid; idx < own * (UW / 2); idx += 256) {
        const int rg = own_lo + idx / (UW / 2), pc = (idx % (UW / 2)) * 2, rl = rg - r0;
        if (rg < B && rl < ROWS && j0 + pc < H) {
          float v0 = 0.f, v1 = 0.f;
#pragma unroll
          for (int w = 0; w < 4; ++w) { v0 += red[w * (ROWS * RED_LD) + rl * RED_LD + pc]; v1 += red[w * (ROWS * RED_LD) + rl * RED_LD + pc + 1]; }
          union { float f[2]; uint64_t u; } pk;
          pk.f[0] = v0 * drop_at(p.dd, key, s, rg, H, j0 + pc); pk.f[1] = v1 * drop_at(p.dd, key, s, rg, H, j0 + pc + 1);
          __hip_atomic_store(reinterpret_cast<uint64_t*>(p.Dx + ((size_t)q * B + rg) * H + j0 + pc), pk.u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      lc_arrive(p.bar + wg, fb + (unsigned)(q + 1));
      if (xi == 0) LC_TS(5, q, 2);
    }
    __syncthreads();
    lc_arrive(p.bar + wg, fb + (unsigned)(F + 1));
    return;
  }

  // ================================================================================== U': dhr, cell backward, dG_s
  constexpr int ROWS = RBU * 16;
  float* red = lc_smem;                                                           // [4 waves][ROWS][RED_LD]
  bf16_t* hl = reinterpret_cast<bf16_t*>(lc_smem + ((4 * ROWS * RED_LD + 3) / 4) * 4);   // [ROWS][4][UW]
  const int ug = wg % p.NGU, part = wg / p.NGU;
  const int u0 = ug * UW;
  const int own = RC_PAN_ROWS / p.MSU, own_lo = part * own;
  const int r0 = own_lo < RC_PAN_ROWS - ROWS ? own_lo : RC_PAN_ROWS - ROWS;
  const int rot = ug % (STEPS / 2);
  bf16x8 wb[STEPS];
  lcb_load_weights<STEPS>(wb, p.WT + (size_t)(H + u0 + (lane & 15)) * p.ldwt, K, kw0, rot, kq);
  // W_r^T fragment of the small product (K = A <= 128: one k-step per wave): B[k = a][n = unit]
  bf16x8 wrt;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int a = wave * 32 + kq + e;
    wrt[e] = a < A ? p.Wr[(size_t)a * p.ldwr + u0 + (lane & 15)] : (bf16_t)0.f;
  }
  constexpr int CPT = (ROWS * UW + 255) / 256;
  bool mine[CPT];
  float direct[CPT], carry[CPT], av[CPT][4], cc[CPT], cp[CPT];
#pragma unroll
  for (int c = 0; c < CPT; ++c) {
    const int cell = tid + c * 256, rg = r0 + cell / UW;
    mine[c] = cell < ROWS * UW && rg >= own_lo && rg < own_lo + own && rg < B;
    carry[c] = 0.f;
  }
  auto prefetch = [&](int s) {                            // saved activations, states and the direct gradient of step s
#pragma unroll
    for (int c = 0; c < CPT; ++c) {
      const int cell = tid + c * 256;
      const size_t row = mine[c] ? r0 + cell / UW : 0;
      const int u = u0 + cell % UW;
      const float* a = p.acts + ((size_t)s * Bs + row) * 4 * R + u;
#pragma unroll
      for (int qq = 0; qq < 4; ++qq) av[c][qq] = a[(size_t)qq * R];
      cc[c] = p.gru ? 0.f : p.Cr[((size_t)s * Bs + row) * R + u];
      cp[c] = s > 0 ? (p.gru ? p.Hr : p.Cr)[((size_t)(s - 1) * Bs + row) * R + u] : 0.f;
      direct[c] = mine[c] ? p.dHr[((size_t)s * Bs + row) * R + u] : 0.f;
    }
  };
  prefetch(F - 1);
  const int lane_off = ((lane >> 4) * RC_PAN_ROWS + r0 + (lane & 15)) * 8;
  constexpr int IPT = (4 * KG * RC_PAN_ROWS + 255) / 256;            // upper bound: own <= 112
  const bf16_t* it_src[IPT]; int it_rg[IPT], it_col[IPT]; bool it_on[IPT];
#pragma unroll
  for (int jj = 0; jj < IPT; ++jj) {
    const int idx = tid + jj * 256;
    const int gq = idx / (KG * own), rem = idx - gq * (KG * own), kgi = rem / own, rg = own_lo + rem % own;
    it_on[jj] = idx < 4 * KG * own && rg < B && rg - r0 < ROWS;
    it_rg[jj] = rg; it_col[jj] = gq * R + u0 + kgi * 8;
    it_src[jj] = hl + ((size_t)(it_on[jj] ? rg - r0 : 0) * 4 + (it_on[jj] ? gq : 0)) * UW + kgi * 8;
  }

  for (int q = 0; q < F; ++q) {
    const int s = F - 1 - q;
    if (q > 0) {
      f32x4 acc[RBU];
#pragma unroll
      for (int i = 0; i < RBU; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
      // dG_{s+1} is complete.  X' (on the critical path) streams the same 1.4 MB panel out of the L2s at the same time: with
      // defer_big U' waits until X' has published dx (its own product is only needed after C' has finished as well)
      if (p.defer_big) lc_wait(relX, fb + (unsigned)q, p.bar); else lc_wait(relG, fb + (unsigned)q, p.bar);
      if (wg == 0) LC_TS(4, q, 0);
      lcb_product<STEPS, PF, RBU>(acc, wb, p.PanG + (size_t)(q - 1) * pan_g + lane_off, K, kw0, rot);
      if (wg == 0) LC_TS(4, q, 1);
      lc_wait(relW, fb + (unsigned)q, p.bar);                        // dWhr_{s+1} is complete
      if (wg == 0) LC_TS(4, q, 2);
      {
        const int k = wave * 32;
        const bf16_t* Aw = p.PanW + (size_t)(q - 1) * pan_w + lane_off;
        if (k < A) {
#pragma unroll
          for (int i = 0; i < RBU; ++i) {
            const bf16x8 fw = *reinterpret_cast<const bf16x8*>(Aw + ((k >> 3) * RC_PAN_ROWS + i * 16) * 8);
            acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw, wrt, acc[i], 0, 0, 0);
          }
        }
      }
      float* prt = red + wave * (ROWS * RED_LD);
#pragma unroll
      for (int i = 0; i < RBU; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) prt[(i * 16 + (lane >> 4) * 4 + r) * RED_LD + (lane & 15)] = acc[i][r];
      __syncthreads();
      if (wg == 0) LC_TS(4, q, 5);
    }
#pragma unroll
    for (int c = 0; c < CPT; ++c) {
      const int cell = tid + c * 256;
      const int row = cell < ROWS * UW ? cell / UW : 0, ul = cell % UW;
      float dh = direct[c];
      if (q > 0) {
#pragma unroll
        for (int w = 0; w < 4; ++w) dh += red[w * (ROWS * RED_LD) + row * RED_LD + ul];
      }
      const LstmGrad g = p.gru ? gru_point_bwd(dh + carry[c], av[c][0], av[c][1], av[c][2], av[c][3], cp[c])
                               : lstm_point_bwd(dh, carry[c], av[c][0], av[c][1], av[c][2], av[c][3], cc[c], cp[c]);
      carry[c] = g.dc_prev;
      if (cell < ROWS * UW) {
        bf16_t* d = hl + (size_t)row * 4 * UW + ul;
        d[0] = (bf16_t)g.di; d[UW] = (bf16_t)g.df; d[2 * UW] = (bf16_t)g.dg; d[3 * UW] = (bf16_t)g.d_o;
      }
    }
    __syncthreads();
    if (wg == 0) LC_TS(4, q, 3);
#pragma unroll
    for (int jj = 0; jj < IPT; ++jj)
      if (it_on[jj]) lc_store16(p.PanG + (size_t)q * pan_g + ((size_t)(it_col[jj] >> 3) * RC_PAN_ROWS + it_rg[jj]) * 8, it_src[jj]);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    lc_arrive(p.bar + wg, fb + (unsigned)(q + 1));
    if (wg == 0) LC_TS(4, q, 4);
    // ---- off the critical path: the row-major copy for the deferred weight-gradient GEMMs
    bf16_t* Gt = p.dG + (size_t)s * Bs * p.ld_dg;
#pragma unroll
    for (int jj = 0; jj < IPT; ++jj)
      if (it_on[jj]) *reinterpret_cast<bf16x8*>(Gt + (size_t)it_rg[jj] * p.ld_dg + it_col[jj]) = *reinterpret_cast<const bf16x8*>(it_src[jj]);
    if (wg == 0 && p.ld_dg > K)
      for (int jj = tid; jj < B * (p.ld_dg - K); jj += 256) Gt[(size_t)(jj / (p.ld_dg - K)) * p.ld_dg + K + jj % (p.ld_dg - K)] = (bf16_t)0.f;
    if (q + 1 < F) prefetch(s - 1);
  }
  __syncthreads();
  lc_arrive(p.bar + wg, fb + (unsigned)(F + 1));
}
template <int RBU, int RBX, int XS = 0> constexpr size_t lcb_smem_bytes() {
  constexpr int RB = RBU > RBX ? RBU : RBX;
  constexpr size_t a = ((size_t)4 * RB * 16 * 17 + 3) / 4 * 4 * 4 + (size_t)RB * 16 * 4 * 16 * 2 + 1024, x = XS ? (size_t)4 * 32 * 65 * 4 : 0;
  return a > x ? a : x;
}
