// The local reconstructor's BACKWARD chain for R above 2048 (BASELINE configs[4]: R = 3584, 64 captions per GPU) as ONE launch —
// the mirror of local_reconstructor.py:37-55 inside train.py:122-123, chain step q <-> s = F-1-q:
//   dhr_s = dHr[s] + dG_{s+1} . W_hh + dWhr_{s+1} . W_r ;  (dG_s, dc) = cell backward
//   dx_s  = dropmask(s) * (dG_s . W_ih) ;  attention backward of step s: dbeta_s, dWhr_s
// loc_chain_bwd_kernel (loc_chain.hpp) keeps [W_ih | W_hh]^T register-resident in workgroups that own 16 output columns over the
// whole contraction; at K = 4R = 14336 such a workgroup would read the whole 1.8 MB gate-gradient panel every step and hold
// 458 KB of weights.  Here the product [dx | dhr] = dG . [W_ih | W_hh] is cut like the forward's hybrid chain, transposed:
//   P  (every workgroup): 64 output columns x one K quarter (R gate rows).  Per wave STEPS k-steps of 32: SR resident in
//      registers, the rest streamed every step from a fragment-major image (1 KB per wave load), through a register ring.
//      Reads its 64-row slice of the panel (458 KB at R = 3584), writes a [64 x 64] fp32 partial (16 KB, written through).
//   C  (workgroup b < B): caption b.  Stateless: sums the 4 K partials of dx_s, applies the dropout mask, dbeta_s = (1/T) dx . h_t
//      (h_t re-read from L2: 63 KB), the (t, k) plane for dWhr_s.  What loc_chain_bwd_kernel accumulates in registers over the F
//      steps (dHs, dUd, dw) is rebuilt AFTER the chain from the per-step dbeta_s and masked dx_s (lcbig_dhs_kernel,
//      lcbig_dud_kernel): the sums over s commute with everything else.
//   L  (workgroup j < R / 16): 16 hidden units x all rows.  dhr_{s} from the direct gradient, the 4 K partials of the step
//      before and dWhr . W_r (K = A <= 128, MFMA); cell backward with the dc carry in registers; publishes dG_s.
// The three stages of a step are strictly sequential, so they are PHASES of the same NWG = (H + R) / 64 * 4 workgroups (256 at
// R = 3584, H = 512: every CU) between two grid barriers (L -> P, C -> L; the last workgroup relays them, lc_wait_or_relay) and
// one partial hand-over (P's dx tiles -> C) per step.
// Limits (host-checked): bf16 path, B <= 64, R % 128 == 0 with R / 128 in {24, 28, 32} (even), H % 64 == 0, H <= 512, A <= 128,
// T <= 32, F <= 40, NWG <= CU count.
#pragma once
#include "common.hpp"
#include "rec_chain.hpp"
#include "loc_chain.hpp"

// resident k-steps per wave for STEPS k-steps in all.  Measured at R = 3584 (C5, ms per train step): 14 resident 3.49, 12: 3.42,
// 10: 3.365, 8: 3.39-3.48, 6: 3.47 — the compiler allocates MFMA operands to the VGPR half of the register file only, above 10
// k-steps (160 registers beside the 128 of the two rings) it spills, and every reload from scratch is a wait for ALL loads in
// flight: an honestly streamed fragment is cheaper than a "resident" one that lives in scratch.  Holding 10-12 k-steps in the
// accumulation half through an inline-asm MFMA (B operand constraint "a") is correct and was slower (3.58-3.68).
#define LB_SR(STEPS) 10
#ifndef LB_PF
#define LB_PF 2                // ring depth of the product's panel / streamed-fragment loads, in pairs of k-steps
#endif
#define LB_NL 6                // further k-steps per wave resident in LDS (96 KB per workgroup)

struct LocBigBwdArgs {
  int F, T, B, Bs, R, H, A, gru;
  int NCB, NWG;                       // column blocks of 64 over [x | hr]; workgroups = 4 NCB (four K quarters)
  const bf16_t* WT; int ldwt;         // [H + R][ldwt]: ([W_ih | W_hh])^T, K = 4R (gate-major) contiguous
  const bf16_t* WstT;                 // streamed k-steps as MFMA B fragments in consumption order, [NWG][4 waves][STEPS - SR][4][64][8]
  const bf16_t* Wr; int ldwr;         // [A][ldwr]
  const float* dHr;                   // [F][B][R] d loss / d hr_s through the output layer
  const float* acts; const float* Cr; const float* Hr;     // saved by the forward
  const float* Hs; const float* Ud; const float* ab; const float* w; const float* Whr;
  bf16_t* dG; int ld_dg;              // [F][B][ld_dg] row-major gate gradients (deferred weight-gradient GEMMs)
  bf16_t* dWhrs; int ld_dwhr;         // [F][B][ld_dwhr] row-major dWhr_s
  float* dxm; float* dbeta;           // [F][B][H] masked dx_s, [F][B][T] dbeta_s (by step s): what the post-chain kernels sum
  bf16_t* PanG; bf16_t* PanW; float* Part;   // exchange, by chain step q: [F][rc_pan_elems(4R)], [F][rc_pan_elems(A)], [F][4][64][H + R]
  unsigned* bar; unsigned* epoch; float* poison;
  DropDesc dd;
  unsigned long long* ts;             // probe builds only (LC_PROBE): roles 3 (workgroup 0: L + P + C) and 4 (a P-only workgroup)
};
#ifdef LC_PROBE
#define LB_WTS(q, i) do { if ((q) == 10 && tid == 0) p.ts[5120 + (i) * 256 + wg] = wall_clock64(); } while (0)
#else
#define LB_WTS(q, i) do { } while (0)
#endif
#define LB_TS(q, i) do { if (wg == 0) LC_TS(3, q, i); else if (wg == p.NWG - 2) LC_TS(4, q, i); } while (0)

// NL further k-steps per wave are resident in LDS (loaded once; the CU's 160 KB hold the reduction buffer, these and the dG
// staging): of the STEPS k-steps per wave SR + NL never move again, STEPS - SR - NL are streamed every step.
template <int STEPS, int SR, int PF, int NL>
__global__ __launch_bounds__(256) void lcbig_bwd_kernel(const LocBigBwdArgs p) {
  constexpr int RB = 4, CG = 4, ROWS = 64, RED_LD = 65, RED_ROWS = 32, LRED_LD = 17, NP = STEPS / 2, NPR = SR / 2, NPS = NP - NPR, UW = 16;
  constexpr int NPL = NL / 2, NPG = NPS - NPL;             // streamed-image pairs: the first NPG through the register ring every step, the last NPL resident in LDS
  static_assert(NL % 2 == 0 && NPG >= 0, "LDS-resident k-steps: whole pairs, no more than the image holds");
  static_assert(SR % 2 == 0 && SR <= STEPS && SR >= 2 * PF, "resident k-steps: whole pairs, at least the prefetch distance");
  extern __shared__ __attribute__((aligned(16))) float lb_smem[];
  float* red = lb_smem;                                                      // P: [4 waves][RED_ROWS][RED_LD], two passes of 32 rows; L: [4 waves][ROWS][LRED_LD]; phase C aliases it
  char* wl = reinterpret_cast<char*>(lb_smem + 4 * RED_ROWS * RED_LD);       // LDS-resident fragments [4 waves][NL][CG][64 lanes][16 bytes]
  bf16_t* hl = reinterpret_cast<bf16_t*>(wl + (size_t)4 * NL * CG * 1024);   // [ROWS][4 gates][UW]: dG_s of this workgroup's units
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wg = blockIdx.x, R = p.R, B = p.B, Bs = p.Bs, H = p.H, A = p.A, F = p.F, T = p.T, K = 4 * R, NT = H + R;
  const unsigned ep = rc_epoch_read(p.epoch), fb = ep << 7;
  unsigned* rel = p.bar + 256;
  const size_t pan_g = rc_pan_elems(K), pan_w = rc_pan_elems(A);
  const bool relay = wg == p.NWG - 1;
  const int kq = (lane >> 4) * 8;
  const uint32_t key = drop_key(p.dd);
  const float invT = 1.0f / (float)T;
  // every workgroup arrives at every barrier (idle ones at once); the write-through stores of the phase are acknowledged first
  // (split in two so that the loads the NEXT phase can already request are issued between the arrival and the wait: in front
  // of the arrival they would sit in the same vmcnt as the stores being acknowledged and delay everybody's barrier)
  // Arrival flags: the 4 NXB workgroups whose tiles are columns of dx come first — the caption phase needs only those, so the
  // hand-over P -> C is a PARTIAL barrier (the caption workgroups poll these flags themselves) and the product's stragglers
  // among the other 4 (NCB - NXB) workgroups finish under the caption phase; barriers 1 and 3 of a step are full and relayed.
  const int cb = wg % p.NCB, kqi = wg / p.NCB, NXB = H >> 6;
  unsigned* myflag = p.bar + (cb < NXB ? kqi * NXB + cb : 4 * NXB + kqi * (p.NCB - NXB) + (cb - NXB));
  auto bar_arrive = [&](unsigned phase) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    lc_arrive(myflag, fb + phase);
  };
  auto bar_wait = [&](unsigned phase) { lc_wait_or_relay(relay, rel, p.bar, p.NWG, fb + phase, p.bar); };
  // wave-level wait for n <= 64 consecutive arrival flags (lane i polls flag i)
  auto flags_wait = [&](const unsigned* f, int n, unsigned target) {
    const unsigned* fl = f + (lane < n ? lane : 0);
    unsigned spin = 0;
    for (;;) {
      const bool ok = (int)(__hip_atomic_load(fl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target) >= 0;
      if (__all(ok)) break;
      if (rc_give_up(p.bar, spin)) break;
    }
    if (RC_ACQUIRE_INV) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  };

  // ---------------------------------------------------------------- P: residents
  const int kw0 = kqi * R + wave * (STEPS * 32);            // this wave's K range inside the quarter (K quarter = R gate rows)
  const int rot = wg % NP;
  auto k_of = [&](int pr, int hh) { int prr = pr + rot; prr = prr >= NP ? prr - NP : prr; return kw0 + (prr * 2 + hh) * 32; };
  bf16x8 wb[SR][CG];
#pragma unroll
  for (int g = 0; g < CG; ++g) {
    const bf16_t* wrow = p.WT + (size_t)(cb * 64 + g * 16 + (lane & 15)) * p.ldwt + kq;
#pragma unroll
    for (int pr = 0; pr < NPR; ++pr)
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) wb[pr * 2 + hh][g] = *reinterpret_cast<const bf16x8*>(wrow + k_of(pr, hh));
  }
  const bf16_t* wst = p.WstT + ((size_t)(wg * 4 + wave) * (STEPS - SR) * CG * 64 + lane) * 8;
  const int lane_off = ((lane >> 4) * RC_PAN_ROWS + (lane & 15)) * 8;          // k-group (lane / 16), row lane % 16
  if constexpr (NL > 0) {     // each wave copies (DMA) and later reads only its own fragments
#pragma unroll
    for (int j = 0; j < NL; ++j)
#pragma unroll
      for (int g = 0; g < CG; ++g)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wst + (size_t)(((2 * NPG + j) * CG + g) * 512)),
                                         (__attribute__((address_space(3))) void*)(wl + (size_t)((wave * NL + j) * CG + g) * 1024), 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }

  // ---------------------------------------------------------------- L: residents (workgroups j < R / 16)
  const bool isL = wg < R / UW, isC = wg < B;
  const int u0 = wg * UW;
  bf16x8 wrt;                                               // W_r^T fragment of the small product: B[k = a][n = unit]
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int a = wave * 32 + kq + e;
    wrt[e] = (isL && a < A) ? p.Wr[(size_t)a * p.ldwr + u0 + (lane & 15)] : (bf16_t)0.f;
  }
  float carry[4] = {0.f, 0.f, 0.f, 0.f};
  // L's operands that do not depend on the chain (saved activations, cell states, the direct gradient) are requested one
  // phase ahead — behind the arrival at the barrier that ends the previous step — and the K partials of the step before as
  // soon as they are complete (behind barrier 2): behind barrier 3 only the dWhr fragments are still to come
  f32x4 av[4], cc = {0.f, 0.f, 0.f, 0.f}, cp = {0.f, 0.f, 0.f, 0.f}, dhd = {0.f, 0.f, 0.f, 0.f}, pk[4];
  // (Per-lane indices are re-derived from a laundered copy of the thread index in every step: as loop invariants every address
  // built from them — 16 + 16 + 11 of them — was computed ahead of the time loop and kept in, i.e. spilled from, registers.)
  auto l_prefetch = [&](int s, int tl) {     // (scalar bases + 32-bit lane offsets: see phase C's prefetch)
    const int lrow = tl >> 2, luq = (tl & 3) * 4, lrow_c = lrow < B ? lrow : 0;
    const unsigned lo4 = (unsigned)lrow_c * (unsigned)(4 * R) + (unsigned)luq, lo1 = (unsigned)lrow_c * (unsigned)R + (unsigned)luq;
    const lc_gf32 ab_ = lc_launder_s(p.acts + (size_t)s * Bs * 4 * R + u0);
#pragma unroll
    for (int g = 0; g < 4; ++g) av[g] = *(lc_gf32x4)(ab_ + lo4 + (unsigned)(g * R));
    if (!p.gru) cc = *(lc_gf32x4)(lc_launder_s(p.Cr + (size_t)s * Bs * R + u0) + lo1);
    if (s > 0) cp = *(lc_gf32x4)(lc_launder_s((p.gru ? p.Hr : p.Cr) + (size_t)(s - 1) * Bs * R + u0) + lo1);
    else cp = f32x4{0.f, 0.f, 0.f, 0.f};
    dhd = *(lc_gf32x4)(lc_launder_s(p.dHr + (size_t)s * Bs * R + u0) + lo1);
  };
  {
    int tl = tid;
    asm volatile("" : "+v"(tl));
    if (isL) l_prefetch(F - 1, tl);
  }

  for (int q = 0; q < F; ++q) {
    const int s = F - 1 - q;
    LB_TS(q, 0);
    int tl = tid;
    asm volatile("" : "+v"(tl));
    const int lrow = tl >> 2, luq = (tl & 3) * 4;          // this thread's cells: row lrow, units u0 + luq .. + 4
    // ============================================================ L(q): dhr_s, cell backward, dG_s
    // streamed weight fragments of P(q): their addresses do not depend on the chain — the first ring slots are requested now
    bf16x8 fw[PF][2][CG];
    const LC_GLOBAL bf16_t* wsl = lc_launder_v(wst);     // (laundered every step: as loop invariants the addresses were materialised ahead of the loop and spilled)
    auto issue_w = [&](int slot, int i) {
#pragma unroll
      for (int hh = 0; hh < 2; ++hh)
#pragma unroll
        for (int g = 0; g < CG; ++g) fw[slot][hh][g] = *(lc_gbf16x8)(wsl + (size_t)(((i * 2 + hh) * CG + g) * 512));
    };
    if (isL) {
      if (q > 0) {
        // dWhr_{s+1} . W_r : K = A <= 128, one k-step per wave
        f32x4 a1[RB];
#pragma unroll
        for (int i = 0; i < RB; ++i) a1[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        const int k = wave * 32;
        if (k < A) {
          const bf16_t* Aw = p.PanW + (size_t)(q - 1) * pan_w + lane_off;
          bf16x8 fr[RB];
#pragma unroll
          for (int i = 0; i < RB; ++i) fr[i] = *reinterpret_cast<const bf16x8*>(Aw + ((k >> 3) * RC_PAN_ROWS + i * 16) * 8);
#pragma unroll
          for (int i = 0; i < RB; ++i) a1[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fr[i], wrt, a1[i], 0, 0, 0);
        }
        float* prt = red + wave * (ROWS * LRED_LD);
#pragma unroll
        for (int i = 0; i < RB; ++i)
#pragma unroll
          for (int r = 0; r < 4; ++r) prt[(i * 16 + (lane >> 4) * 4 + r) * LRED_LD + (lane & 15)] = a1[i][r];
      }
      f32x4 dh = dhd;
      __syncthreads();
      if (q > 0) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float v = (pk[0][e] + pk[1][e]) + (pk[2][e] + pk[3][e]);
#pragma unroll
          for (int w = 0; w < 4; ++w) v += red[w * (ROWS * LRED_LD) + lrow * LRED_LD + luq + e];
          dh[e] += v;
        }
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const LstmGrad g = p.gru ? gru_point_bwd(dh[e] + carry[e], av[0][e], av[1][e], av[2][e], av[3][e], cp[e])
                                 : lstm_point_bwd(dh[e], carry[e], av[0][e], av[1][e], av[2][e], av[3][e], cc[e], cp[e]);
        carry[e] = g.dc_prev;
        bf16_t* d = hl + (size_t)lrow * 4 * UW + luq + e;
        d[0] = (bf16_t)g.di; d[UW] = (bf16_t)g.df; d[2 * UW] = (bf16_t)g.dg; d[3 * UW] = (bf16_t)g.d_o;
      }
      __syncthreads();
      // publish dG_s: 4 gates x 2 k-groups x 64 rows items of 16 bytes (written through), and the row-major copy
      bf16_t* Gt = p.dG + (size_t)s * Bs * p.ld_dg;
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) {
        const int idx = tid + jj * 256, gq = idx >> 7, kgi = (idx >> 6) & 1, rg = idx & 63;
        const int col = gq * R + u0 + kgi * 8;
        const bf16_t* src = hl + ((size_t)rg * 4 + gq) * UW + kgi * 8;
        if (rg < B) {
          lc_store16(p.PanG + (size_t)q * pan_g + ((size_t)(col >> 3) * RC_PAN_ROWS + rg) * 8, src);
          *reinterpret_cast<bf16x8*>(Gt + (size_t)rg * p.ld_dg + col) = *reinterpret_cast<const bf16x8*>(src);
        }
      }
      if (wg == 0 && p.ld_dg > K)
        for (int jj = tid; jj < B * (p.ld_dg - K); jj += 256) Gt[(size_t)(jj / (p.ld_dg - K)) * p.ld_dg + K + jj % (p.ld_dg - K)] = (bf16_t)0.f;
    }
    LB_TS(q, 1);
    bar_arrive(3u * (unsigned)q + 1u);
    if constexpr (NPG > 0) {
#pragma unroll
      for (int i = 0; i < PF; ++i)
        if (i < NPG) issue_w(i % PF, i);
    }
    bar_wait(3u * (unsigned)q + 1u);
    LB_TS(q, 2);
    LB_WTS(q, 0);

    // ============================================================ P(q): partial [dx | dhr] of this tile
    {
      auto pair_of = [&](int i) { return i < NPS ? NPR + i : i - NPS; };      // streamed pairs first, the resident ones cover their tail
      const bf16_t* Ag = p.PanG + (size_t)q * pan_g + lane_off;
      bf16x8 fa[PF][2][RB];
      auto issue_pair = [&](int slot, int pr) {
#pragma unroll
        for (int hh = 0; hh < 2; ++hh)
#pragma unroll
          for (int i = 0; i < RB; ++i) fa[slot][hh][i] = *reinterpret_cast<const bf16x8*>(Ag + ((size_t)(k_of(pr, hh) >> 3) * RC_PAN_ROWS + i * 16) * 8);
      };
      f32x4 acc[RB][CG];
#pragma unroll
      for (int i = 0; i < RB; ++i)
#pragma unroll
        for (int g = 0; g < CG; ++g) acc[i][g] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int i = 0; i < PF; ++i)
        if (i < NP) issue_pair(i, pair_of(i));
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int it = 0; it < NP; ++it) {
        const int slot = it % PF;
        const int pr = pair_of(it);
#pragma unroll
        for (int hh = 0; hh < 2; ++hh)
#pragma unroll
          for (int i = 0; i < RB; ++i)
#pragma unroll
            for (int g = 0; g < CG; ++g) {
              bf16x8 w;
              if (it < NPG) w = fw[slot][hh][g];
              else if (it < NPS) w = *reinterpret_cast<const bf16x8*>(wl + ((size_t)((wave * NL + (it - NPG) * 2 + hh) * CG + g) * 64 + lane) * 16);
              else w = wb[(it < NPS ? 0 : pr * 2 + hh)][g];
              acc[i][g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[slot][hh][i], w, acc[i][g], 0, 0, 0);
            }
        if (it + PF < NP) {
          __builtin_amdgcn_sched_barrier(0);
          issue_pair(slot, pair_of(it + PF));
          if (it + PF < NPG) issue_w(slot, it + PF);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      LB_TS(q, 3);
      LB_WTS(q, 1);
      // the four waves' K parts are summed through LDS in two passes of 32 rows (a 64-row buffer would take the room of two
      // more LDS-resident k-steps per wave).  thread = (row 16 rr + tid / 16, 4 columns); a wave's store covers 4 rows x 256
      // contiguous bytes, written through (as 8-byte stores 64 bytes apart the 16 KB took 6 us to be acknowledged)
      float* prt = red + wave * (RED_ROWS * RED_LD);
      const int prow = tid >> 4, pc0 = (tid & 15) * 4;
      float* dst = p.Part + (((size_t)q * 4 + kqi) * ROWS + prow) * NT + cb * 64 + pc0;
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        if (half) __syncthreads();
#pragma unroll
        for (int i2 = 0; i2 < 2; ++i2)
#pragma unroll
          for (int g = 0; g < CG; ++g)
#pragma unroll
            for (int r = 0; r < 4; ++r) prt[(i2 * 16 + (lane >> 4) * 4 + r) * RED_LD + g * 16 + (lane & 15)] = acc[half * 2 + i2][g][r];
        __syncthreads();
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {
          f32x4 v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int w = 0; w < 4; ++w)
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] += red[w * (RED_ROWS * RED_LD) + (rr * 16 + prow) * RED_LD + pc0 + e];
          lc_store16f(dst + (size_t)(half * 32 + rr * 16) * NT, v);
        }
      }
    }
    LB_TS(q, 4);
    bar_arrive(3u * (unsigned)q + 2u);
    LB_WTS(q, 2);
    // phase C's saved operands do not depend on the chain: requested while the barrier completes
    int tc = tid;
    asm volatile("" : "+v"(tc));
    const int tt = tc & 31, h8 = tc >> 5;                     // (decoder step, eighth of the hidden columns)
    const int ck = tc & 127, cth = wave >> 1;                 // (attention column, half of the decoder steps: wave-uniform)
    float hv[64], udv[16], whk = 0.f, wk = 0.f;
    if (isC) {
      // Scalar bases + 32-bit lane offsets, every load unconditional (clamped addresses: what lies beyond T / H / A meets zeros
      // or a guard below).  As per-lane 64-bit pointers these addresses were loop invariants or induction variables, were
      // spilled, and every reload from scratch waited for ALL loads in flight: the "prefetch" took 4.5 us.
      const int ckc = ck < A ? ck : 0;
      const lc_gf32 whb = lc_launder_s(p.Whr + ((size_t)s * Bs + wg) * A);
      const lc_gf32 abb = lc_launder_s(p.ab);
      const lc_gf32 wwb = lc_launder_s(p.w);
      const lc_gf32 udb = lc_launder_s(p.Ud + (size_t)wg * A);
      const lc_gf32 hsb = lc_launder_s(p.Hs + (size_t)wg * H);
      whk = whb[ckc] + abb[ckc];
      wk = wwb[ckc];
#pragma unroll
      for (int i = 0; i < 16; ++i) udv[i] = (udb + (size_t)(cth * 16 + i < T ? cth * 16 + i : 0) * (size_t)(Bs * A))[ckc];     // (scalar row address + lane column)
      const unsigned hoff = (unsigned)(tt < T ? tt : 0) * (unsigned)(Bs * H) + (unsigned)(64 * h8 < H ? 64 * h8 : 0);
#pragma unroll
      for (int i = 0; i < 64; i += 4) {
        const f32x4 v = *(lc_gf32x4)(hsb + hoff + i);
        hv[i] = v[0]; hv[i + 1] = v[1]; hv[i + 2] = v[2]; hv[i + 3] = v[3];
      }
      if (ck >= A) wk = 0.f;
    }
    LB_WTS(q, 3);
    if (isC) {
      if (tid < 64) flags_wait(p.bar, 4 * NXB, fb + 3u * (unsigned)q + 2u);
      LB_WTS(q, 4);
      __syncthreads();
    }
    LB_TS(q, 5);

    // ============================================================ C(q): attention backward of step s for caption wg
    if (isC) {
      float* sdx = red;                    // [512] masked dx_s
      float* spd = sdx + 512;              // [8][32] partial dbeta
      float* sdbt = spd + 256;             // [32] dbeta
      float* sdw = sdbt + 32;              // [2][128] partial dWhr
      bf16_t* swl = reinterpret_cast<bf16_t*>(sdw + 256);   // [128] dWhr_s (bf16)
      const int b = wg;
#pragma unroll
      for (int qq = 0; qq < 2; ++qq) {
        const int h = tid + 256 * qq;
        if (h < H) {
          float v[4];
#pragma unroll
          for (int z = 0; z < 4; ++z) v[z] = p.Part[(((size_t)q * 4 + z) * ROWS + b) * NT + h];
          const float d = ((v[0] + v[1]) + (v[2] + v[3])) * drop_at(p.dd, key, s, b, H, h);
          sdx[h] = d;
          p.dxm[((size_t)s * Bs + b) * H + h] = d;
        } else if (h < 512) {
          sdx[h] = 0.f;
        }
      }
      __syncthreads();
      {
        float v0 = 0.f, v1 = 0.f;
#pragma unroll
        for (int i = 0; i < 64; i += 4) {
          const f32x4 d4 = *reinterpret_cast<const f32x4*>(sdx + 64 * h8 + i);
          v0 += d4[0] * hv[i] + d4[2] * hv[i + 2];
          v1 += d4[1] * hv[i + 1] + d4[3] * hv[i + 3];
        }
        spd[h8 * 32 + tt] = v0 + v1;
      }
      __syncthreads();
      if (tid < 32) {
        float v = 0.f;
#pragma unroll
        for (int z = 0; z < 8; ++z) v += spd[z * 32 + tid];
        v *= invT;
        sdbt[tid] = v;
        if (tid < T) p.dbeta[((size_t)s * Bs + b) * T + tid] = v;
      }
      __syncthreads();
      {
        float dwh = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int t = cth * 16 + i;
          if (t < T) {
            const float tz = rn_tanh(whk + udv[i]);
            dwh += sdbt[t] * wk * (1.f - tz * tz);
          }
        }
        sdw[cth * 128 + ck] = dwh;
      }
      __syncthreads();
      if (tid < 128) swl[tid] = (bf16_t)(tid < A ? sdw[tid] + sdw[128 + tid] : 0.f);
      __syncthreads();
      // publish dWhr_s[b] (k-groups up to the next multiple of 32 columns; swl holds zeros beyond A) + the row-major copy
      if (tid < (((A + 31) >> 5) << 2)) lc_store16(p.PanW + (size_t)q * pan_w + ((size_t)tid * RC_PAN_ROWS + b) * 8, swl + tid * 8);
      if (tid < (p.ld_dwhr >> 3) && tid < 16)
        *reinterpret_cast<bf16x8*>(p.dWhrs + ((size_t)s * Bs + b) * p.ld_dwhr + tid * 8) = *reinterpret_cast<const bf16x8*>(swl + tid * 8);
    }
    LB_TS(q, 6);
    bar_arrive(3u * (unsigned)q + 3u);
    if (isL && q + 1 < F) {
      l_prefetch(s - 1, tl);
      // the K partials of this workgroup's columns of dhr: their four producers finished P(q) a caption phase ago
      {
        const unsigned* f = p.bar + 4 * NXB + ((H + u0) >> 6) - NXB + (lane & 3) * (p.NCB - NXB);
        unsigned spin = 0;
        for (;;) {
          const bool ok = (int)(__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - (fb + 3u * (unsigned)q + 2u)) >= 0;
          if (__all(ok)) break;
          if (rc_give_up(p.bar, spin)) break;
        }
        if (RC_ACQUIRE_INV) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      }
#pragma unroll
      for (int z = 0; z < 4; ++z)
        pk[z] = *(lc_gf32x4)(lc_launder_s(p.Part + ((size_t)q * 4 + z) * ROWS * NT + H + u0) + (unsigned)lrow * (unsigned)NT + (unsigned)luq);
    } else {
      // (defined on every path: assigned under the condition only, the values of the step before stay live through the P and
      // C phases of every step — 44 registers the product's ring was spilled for)
      const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int z = 0; z < 4; ++z) { av[z] = z4; pk[z] = z4; }
      cc = z4; cp = z4; dhd = z4;
    }
    bar_wait(3u * (unsigned)q + 3u);
    LB_TS(q, 7);
  }
  // everybody has passed the last barrier (and read the epoch long ago)
  if (relay && tid == 0) {
    __hip_atomic_store(p.epoch, ep + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); rc_stamp_slot(p.epoch)[1] = wall_clock64();
    if (__hip_atomic_load(p.bar + 257, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) *p.poison = __builtin_nanf("");
  }
}
template <int NL> constexpr size_t lcbig_smem_bytes() { return (size_t)4 * 32 * 65 * 4 + (size_t)4 * NL * 4 * 1024 + (size_t)64 * 4 * 16 * 2; }

// The streamed k-steps of lcbig_bwd_kernel<STEPS, SR, .> as MFMA B fragments in the order it consumes them:
// dst[wg][wave][js][g][lane][8] from WT = ([W_ih | W_hh])^T [H + R][ldwt].  Run after every update of the weights.
__global__ __launch_bounds__(256) void lcbig_pack_stream_kernel(const bf16_t* __restrict__ WT, int ldwt, int R, int NCB, int STEPS, int SR,
                                                                bf16_t* __restrict__ dst, size_t n_frag) {
  const size_t f = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (f >= n_frag) return;
  const int NS = STEPS - SR, NP = STEPS / 2, NPR = SR / 2;
  const int lane = (int)(f & 63), g = (int)((f >> 6) & 3);
  const size_t r = f >> 8;
  const int js = (int)(r % NS), wave = (int)((r / NS) & 3), wg = (int)(r / NS / 4);
  const int cb = wg % NCB, kqi = wg / NCB;
  int prr = NPR + js / 2 + wg % NP; prr = prr >= NP ? prr - NP : prr;
  const int k = kqi * R + wave * (STEPS * 32) + (prr * 2 + (js & 1)) * 32 + (lane >> 4) * 8;
  *reinterpret_cast<bf16x8*>(dst + f * 8) = *reinterpret_cast<const bf16x8*>(WT + (size_t)(cb * 64 + g * 16 + (lane & 15)) * ldwt + k);
}

// After the chain: what loc_chain_bwd_kernel's caption workgroups accumulate over the F steps.
//   dHs[t][b][h] = sum_s (beta_s[b][t] / T) dxm_s[b][h]          (attention path into the decoder's hidden states)
__global__ __launch_bounds__(256) void lcbig_dhs_kernel(const float* __restrict__ beta, const float* __restrict__ dxm, float* __restrict__ dHs,
                                                        int F, int T, int B, int H) {
  // grid (B, H / 256, 2): blockIdx.z = half of the decoder steps (16 accumulators per thread)
  __shared__ float sb[40 * 16];
  const int b = blockIdx.x, h = blockIdx.y * 256 + threadIdx.x, t0 = blockIdx.z * 16;
  for (int i = threadIdx.x; i < F * 16; i += 256) { const int s = i >> 4, t = t0 + (i & 15); sb[i] = t < T ? beta[((size_t)s * B + b) * T + t] : 0.f; }
  __syncthreads();
  if (h >= H) return;
  float acc[16];
#pragma unroll
  for (int t = 0; t < 16; ++t) acc[t] = 0.f;
  for (int s = 0; s < F; ++s) {
    const float x = dxm[((size_t)s * B + b) * H + h];
#pragma unroll
    for (int t = 0; t < 16; ++t) acc[t] += sb[s * 16 + t] * x;
  }
  const float invT = 1.0f / (float)T;
#pragma unroll
  for (int t = 0; t < 16; ++t)
    if (t0 + t < T) dHs[((size_t)(t0 + t) * B + b) * H + h] = acc[t] * invT;
}
//   dUd[t][b][k] = sum_s dbeta_s[b][t] w_k (1 - tanh^2(Whr_s[b][k] + Ud[t][b][k] + b_k)) ;  dw[b][k] = sum_{s,t} dbeta_s[b][t] tanh(.)
// grid (B, 4): blockIdx.y = a quarter of the decoder steps; its part of dw goes to chunk blockIdx.y of dwacc (nch = 4 chunks, summed
// by the column sum that follows, like the per-step path's RN_TCH chunks)
__global__ __launch_bounds__(256) void lcbig_dud_kernel(const float* __restrict__ dbeta, const float* __restrict__ Whr, const float* __restrict__ Ud,
                                                        const float* __restrict__ ab, const float* __restrict__ w, float* __restrict__ dUd,
                                                        bf16_t* __restrict__ dUd_lp, int ld_dUd, float* __restrict__ dwacc, int nch,
                                                        int F, int T, int B, int A) {
  __shared__ float sdb[40 * 8];
  __shared__ float sdw[2 * 128];
  const int b = blockIdx.x, k = threadIdx.x & 127, th = threadIdx.x >> 7, t0 = blockIdx.y * 8;
  for (int i = threadIdx.x; i < F * 8; i += 256) { const int s = i >> 3, t = t0 + (i & 7); sdb[i] = t < T ? dbeta[((size_t)s * B + b) * T + t] : 0.f; }
  __syncthreads();
  float dwa = 0.f;
  if (k < A) {
    const float wk = w[k], abk = ab[k];
    float ud[4], dud[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { const int t = t0 + th * 4 + i; ud[i] = t < T ? Ud[((size_t)t * B + b) * A + k] : 0.f; dud[i] = 0.f; }
    for (int s = 0; s < F; ++s) {
      const float whk = Whr[((size_t)s * B + b) * A + k] + abk;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int t = t0 + th * 4 + i;
        if (t < T) {
          const float tz = rn_tanh(whk + ud[i]);
          const float db = sdb[s * 8 + th * 4 + i];
          dud[i] += db * wk * (1.f - tz * tz);
          dwa += db * tz;
        }
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int t = t0 + th * 4 + i;
      if (t < T) { dUd[((size_t)t * B + b) * A + k] = dud[i]; dUd_lp[((size_t)t * B + b) * ld_dUd + k] = (bf16_t)dud[i]; }
    }
  }
  for (int i = 0; i < 4; ++i) {
    const int t = t0 + th * 4 + i;
    if (t < T) for (int j = A + k; j < ld_dUd; j += 128) dUd_lp[((size_t)t * B + b) * ld_dUd + j] = (bf16_t)0.f;
  }
  sdw[th * 128 + k] = dwa;
  __syncthreads();
  if (th == 0 && k < A && (int)blockIdx.y < nch) dwacc[((size_t)blockIdx.y * B + b) * A + k] = sdw[k] + sdw[128 + k];
}
