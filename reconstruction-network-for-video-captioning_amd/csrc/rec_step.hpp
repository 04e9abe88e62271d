// Fused recurrent step of the reconstructor's LSTM (global reconstructor, forward):
//     gates = Xg[t] + h_{t-1} . W_hh^T ;  (h_t, c_t) = LSTM pointwise          (global_reconstructor.py:43, nn.LSTM)
// in ONE launch per time step — no split-K slabs, no separate pointwise kernel.
//
// The step is a dependent link of a chain: B <= 112 rows of activations against the whole W_hh (4R x R, 19 MB bf16 at
// R = 1536), whose result the next launch needs.  A workgroup owns 8 hidden units = 32 weight rows (the four gate rows of
// each unit: the packed image `Wg` is gate-interleaved, row (u/8)*32 + gate*8 + u%8) over the FULL contraction, so the
// pre-activations of its units are complete inside the workgroup and the LSTM pointwise runs in the epilogue.
//   * the four waves split K; a wave loads its weight fragments straight into MFMA B-operand registers (all of them, up
//     front) and streams its activation fragments from L2 through a register ring (PF pairs of k-steps ahead) — no LDS
//     staging, no barrier in the main loop, plain loads whose in-order completion the compiler's wait counts track;
//   * the four partial [112 x 32] tiles are summed through LDS, then 256 threads apply the cell to the 8 units x B rows.
// Traffic per launch: W_hh once (HBM / MALL) + the activation block once per workgroup (L2 hits) + h, c, gates out.
#pragma once
#include "common.hpp"

struct RecStepArgs {
  int B, R, K;                    // rows, hidden size, contraction length (= R for the global reconstructor)
  const bf16_t* A; int lda;       // [B][lda]   operand copy of h_{t-1}
  const bf16_t* W; int ldw;       // [4R][ldw]  gate-interleaved packed W_hh
  const float* X; int ldx;        // [B][ldx]   input part of the gates + biases, gate-major columns (gate*R + u)
  const float* c_prev;            // [B][R]
  float* h_out; float* c_out;     // [B][R]
  float* acts;                    // [B][4R] post-activation gates (gate-major), for the backward
  bf16_t* h_lp; int ld_hlp;       // [B][ld_hlp] operand copy of h_t, zero padded
};

#define RS_MB 7               // 16-row blocks: B <= 112
#define RS_RED_LD 33

// STEPS = k32-steps per wave (K <= 4 * 32 * STEPS); PF = activation prefetch distance in k-steps (<= 6)
template <int STEPS, int PF>
__global__ __launch_bounds__(256) void rec_step_fused_kernel(const RecStepArgs p) {
  __shared__ float red[4 * RS_MB * 16 * RS_RED_LD];      // 4 waves x 112 rows x 33
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wrow0 = blockIdx.x * 32;                     // first packed weight row of this workgroup
  const int kw0 = wave * (STEPS * 32);                   // this wave's K range
  const int kq = (lane >> 4) * 8;

  // ---- epilogue operands (input part of the gates, c_{t-1}) of this thread's cells: requested first, used last, so
  // their HBM latency hides behind the main loop (thread -> cells tid, tid + 256, ...: row = cell / 8, unit = cell % 8)
  constexpr int CPT = (RS_MB * 16 * 8 + 255) / 256;      // cells per thread
  const int u0 = blockIdx.x * 8, R = p.R;
  float xg[CPT][4], cpv[CPT];
#pragma unroll
  for (int c = 0; c < CPT; ++c) {
    const int cell = tid + c * 256, row = cell >> 3, u = u0 + (cell & 7);
    if (cell < p.B * 8) {
#pragma unroll
      for (int q = 0; q < 4; ++q) xg[c][q] = p.X[(size_t)row * p.ldx + q * R + u];
      cpv[c] = p.c_prev ? p.c_prev[(size_t)row * R + u] : 0.f;
    } else {
      xg[c][0] = xg[c][1] = xg[c][2] = xg[c][3] = 0.f; cpv[c] = 0.f;
    }
  }
  // ---- weights: 2 column groups x STEPS fragments, all issued now
  bf16x8 wb[STEPS][2];
  const bf16_t* wrow[2];
#pragma unroll
  for (int g = 0; g < 2; ++g) wrow[g] = p.W + (size_t)(wrow0 + g * 16 + (lane & 15)) * p.ldw + kq;
  const int rot_w = blockIdx.x % (STEPS / 2);
#pragma unroll
  for (int s = 0; s < STEPS; ++s)
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      int prw = (s >> 1) + rot_w; prw = prw >= STEPS / 2 ? prw - STEPS / 2 : prw;
      const int k = kw0 + (prw * 2 + (s & 1)) * 32;
      const bf16_t* src = (k + kq < p.K) ? wrow[g] + k : wrow[g];
#ifdef RS_PROBE_SKIP_B
      wb[s][g] = bf16x8{1, 1, 1, 1, 1, 1, 1, 1}; (void)src;
#else
      wb[s][g] = *reinterpret_cast<const bf16x8*>(src);
#endif
    }
  // ---- activations: register ring of PF PAIRS of k-steps, 7 row blocks each.  A pair covers 64 consecutive k = one
  // 128-byte line per row, and its two halves are requested back to back (the second is a hit on the line the first
  // one brought in; requested a k-step apart, the line is evicted from the 16 KiB L1 in between and fetched twice)
  const bf16_t* arow[RS_MB];
#pragma unroll
  for (int i = 0; i < RS_MB; ++i) {
    int r = i * 16 + (lane & 15);
    r = r < p.B ? r : p.B - 1;
    arow[i] = p.A + (size_t)r * p.lda + kq;
  }
  constexpr int NP = STEPS / 2;                          // pairs per wave (STEPS is even)
  // every workgroup reads the same activation block: start each one at a different k so that at any moment the chip's
  // requests are spread over all L2 channels instead of the few that one 64-k column of the block maps to
  const int rot = blockIdx.x % NP;
  bf16x8 fa[PF][2][RS_MB];
  auto issue_pair = [&](int slot, int pr) {
#pragma unroll
    for (int hh = 0; hh < 2; ++hh)
#pragma unroll
      for (int i = 0; i < RS_MB; ++i) {
        int prr = pr + rot; prr = prr >= NP ? prr - NP : prr;
        const int k = kw0 + (prr * 2 + hh) * 32;
        const bf16_t* src = (k + kq < p.K) ? arow[i] + k : arow[i];
#ifdef RS_PROBE_SKIP_A
        fa[slot][hh][i] = bf16x8{1, 1, 1, 1, 1, 1, 1, 1}; (void)src;
#else
        fa[slot][hh][i] = *reinterpret_cast<const bf16x8*>(src);
#endif
      }
  };
#pragma unroll
  for (int pr = 0; pr < PF; ++pr)
    if (pr < NP) issue_pair(pr, pr);
  __builtin_amdgcn_sched_barrier(0);     // everything above is issued before the first MFMA: the scheduler must not sink it

  f32x4 acc[RS_MB][2];
#pragma unroll
  for (int i = 0; i < RS_MB; ++i) { acc[i][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[i][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }

#pragma unroll
  for (int pr = 0; pr < NP; ++pr) {
    const int slot = pr % PF;
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
      const int s = pr * 2 + hh;                         // weights are loaded in the same rotated order (below)
      int prr = pr + rot; prr = prr >= NP ? prr - NP : prr;
      const bool live = kw0 + (prr * 2 + hh) * 32 + kq < p.K;
      bf16x8 b0 = wb[s][0], b1 = wb[s][1];
      if (!live) { b0 = bf16x8{0, 0, 0, 0, 0, 0, 0, 0}; b1 = b0; }
#ifndef RS_PROBE_SKIP_MMA
#pragma unroll
      for (int i = 0; i < RS_MB; ++i) {
        acc[i][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[slot][hh][i], b0, acc[i][0], 0, 0, 0);
        acc[i][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[slot][hh][i], b1, acc[i][1], 0, 0, 0);
      }
#else
      acc[0][0] += f32x4{(float)fa[slot][hh][0][0], (float)b0[0], (float)b1[0], (float)fa[slot][hh][6][0]};
#endif
    }
    if (pr + PF < NP) {
      __builtin_amdgcn_sched_barrier(0);
      issue_pair(slot, pr + PF);         // refill the slot just consumed
      __builtin_amdgcn_sched_barrier(0);
    }
  }

#ifdef RS_PROBE_SKIP_EPI
  if (acc[0][0][0] == 12345.678f) p.h_out[tid] = acc[0][0][1] + xg[0][0] + cpv[0];
  return;
#endif
  // ---- sum the four K-partials through LDS
  {
    float* mine = red + wave * (RS_MB * 16 * RS_RED_LD);
    const int rr = (lane >> 4) * 4, cc = lane & 15;
#pragma unroll
    for (int i = 0; i < RS_MB; ++i)
#pragma unroll
      for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int r = 0; r < 4; ++r) mine[(i * 16 + rr + r) * RS_RED_LD + g * 16 + cc] = acc[i][g][r];
  }
  __syncthreads();
  // ---- LSTM pointwise for 8 units x B rows: thread -> (row, unit)
#pragma unroll
  for (int c = 0; c < CPT; ++c) {
    const int cell = tid + c * 256;
    if (cell >= p.B * 8) break;
    const int row = cell >> 3, ul = cell & 7, u = u0 + ul;
    float g4[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float v = xg[c][q];
#pragma unroll
      for (int w = 0; w < 4; ++w) v += red[w * (RS_MB * 16 * RS_RED_LD) + row * RS_RED_LD + q * 8 + ul];
      g4[q] = v;
    }
    const size_t o = (size_t)row * R + u;
    const LstmOut r = lstm_point(g4[0], g4[1], g4[2], g4[3], cpv[c]);
    p.h_out[o] = r.h;
    p.c_out[o] = r.c;
    p.h_lp[(size_t)row * p.ld_hlp + u] = (bf16_t)r.h;
    float* a = p.acts + (size_t)row * 4 * R + u;
    a[0] = r.i; a[R] = r.f; a[2 * R] = r.g; a[3 * R] = r.o;
  }
  // zero padding of the operand copy, columns [R, ld_hlp)
  if (blockIdx.x == 0)
    for (int j = tid; j < p.B * (p.ld_hlp - R); j += 256) {
      const int row = j / (p.ld_hlp - R), c = R + j % (p.ld_hlp - R);
      p.h_lp[(size_t)row * p.ld_hlp + c] = (bf16_t)0.f;
    }
}
