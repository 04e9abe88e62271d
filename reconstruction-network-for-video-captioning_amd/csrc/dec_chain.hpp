// The decoder's teacher-forced forward chain (decoder.py:45-70 inside train.py:39-52) as ONE launch.
//
// Per step the per-launch path runs two kernels: h_{t-1} . [W_hh ; attn_W]^T (split-K slabs) and the cell kernel (one
// workgroup per caption: attention scores, context = (1/F) sum_f a_f P[b,f,:], gates, pointwise), each re-reading its
// loop invariants — the packed weights (2.2 MB) and the caption's block of P = enc . W_ih[:,E:]^T (115 KB bf16 per
// caption, 11.5 MB per step) — from L2 / memory every step.  Here both stay on chip for all T steps and the step becomes
// two phases of one persistent kernel, separated by grid barriers (rec_chain.hpp):
//   phase A, workgroup a < NA = (4H + A) / 16: owns 16 columns of [W_hh ; attn_W] (in MFMA B-operand registers), reads
//            the bf16 copy of h_{t-1} of all captions (k-group-major exchange panel), writes the finished fp32
//            pre-activations of its columns for every caption (no split-K slabs);
//   phase B, workgroup b < B: owns caption b — its block of P in registers (wave = gate, lane = 8 units, 32 frames x 16
//            bytes), its rows of Uv, c_{t-1}; reads its 4H + A pre-activations, computes scores, context, gates and the
//            cell, publishes h_t (bf16) for phase A of the next step and writes everything the backward needs.
// A workgroup takes part in both phases (grid = max(NA, B) <= CU count, one workgroup per CU).
// Round 5 (template parameter RP, the default): phase A is tiled as 64 columns x one of 4 row parts (28 captions) instead of 16 columns x
// all rows, the roles are laid out XCD-aware, and the B -> A hand-over stays inside a row part (see DCF_PARTS / DCF_NA below).
// Limits: bf16 path, H % 8 == 0, H <= 512, F <= 32, A <= 128, (4H + A) % 16 == 0, B <= 112.
#pragma once
#ifndef DC_STAMP_PAIR
#define DC_STAMP_PAIR 1
#endif
#include "common.hpp"
#include "rec_chain.hpp"

struct DecChainArgs {
  int T, B, F, H, A, gru;
  int Bs;                          // rows per time step of the saved [T][.][.] tensors (= batch size; B = rows of THIS launch, a row group; pointers pre-offset; exchange buffers are private to the launch and indexed with B)
  const bf16_t* W; int ldw;        // [4H + A ..][ldw] packed [W_hh ; attn_W] (Wcomb)
  const bf16_t* P; int ldp;        // [B][F][ldp]
  const float* Uv; const float* ab; const float* w;
  const float* Xe;                 // [T][B][4H] emb . W_e^T + biases
  float* G1;                       // [T][B][4H + A] exchange: recurrent pre-activations (ll: 8-byte words {value, stamp})
  unsigned* epoch; int ll;         // ll = 1: phase A -> B hand-over through stamped words instead of a grid barrier
  int master;                      // 1: the last workgroup of the grid is the barrier master (rc_master_loop)
  int partial; unsigned* rep;      // RP kernels: phase A waits for the captions of its row part only, through per-XCD replicas of the part's arrival line (see dec_chain_bwd_kernel)
  float* poison;                   // see rc_give_up (rec_chain.hpp)
  float* mp; float mp_scale;       // optional: mp_scale * sum_t h_t [B][H] (the global reconstructor's mean-pooled input)
  bf16_t* Xcat; int ld_xcat; DropDesc xdd;   // optional (global reconstructor, ld_xcat == 2H): its LSTM input operand [T][Bs][2H] = [h_t ; drop_t(mp)]
                                   // (global_reconstructor.py:38-41) written by the chain itself instead of xcat_global_kernel behind it
  bf16_t* Pan;                     // [T][rc_pan_elems(H)] exchange: h_t, k-group-major
  float* Hs; float* Cs; float* acts;   // [T][B][H], [T][B][H], [T][B][4H]
  bf16_t* Hlp; int ld_hlp;         // [T][B][ld_hlp] row-major operand copy of h_t, zero padded
  float* Wh; float* att;           // [T][B][A], [T][B][F]
  unsigned* bar;
  int softmax;                     // 1: softmax over the frames of the attention energies (recnet_config.decoder_attn_normalize)
  unsigned long long* ts;          // probe only (DC_PROBE_TS): [T][12] timestamps of one workgroup
};

#define DC_RED_LD 17
#if defined(LC_PROBE) && !defined(DC_PROBE_TS)      // the library's probe build (make probe): stamps of workgroup 0
#define DC_PROBE_TS
#define DC_PROBE_WG 0
#endif
#ifdef DC_PROBE_TS
#define DC_TS(i) do { if (wg == DC_PROBE_WG && tid == 0) p.ts[(size_t)t * 12 + (i)] = wall_clock64(); } while (0)
#else
#define DC_TS(i) do { } while (0)
#endif

// XF: frames 32 .. 47 of the caption's P block and Uv rows are served from (dynamic) LDS — the register file holds 32
// frames; F <= 32 launches the XF = false instance with no dynamic LDS.
#define DC_XF 16              // extra frames
// LW (round 4, F <= 32): the attention projection W h_{t-1} of a caption is computed BY ITS OWN workgroup — attn_W (128 x 512
// bf16 = 128 KB) stays in LDS for the whole launch and h_{t-1}[b] never left the workgroup — so scores and context no longer
// wait for phase A: the scores of step t + 1 are formed while the grid barrier of step t completes, the context MFMAs while
// phase A's gate pre-activations travel, and only the cell waits for them.  Phase A shrinks to the 4H gate columns.
// RP (round 5): phase A tiled as 64 columns x one of 4 row parts (28 rows) instead of 16 columns x all 112 rows — the same number of
// workgroups and (almost) of MFMAs, but a workgroup pulls 32 rows x H of the h_{t-1} panel through its CU's 64 B/clk L1 path instead
// of 112 (32 KB instead of 115 KB per step: 0.2 us instead of 0.75), and its consumers and producers all lie in ITS row part, so the
// B -> A hand-over can be partial (DecChainArgs::partial, as in dec_chain_bwd_kernel).
#define DCF_PARTS 4
#define DCF_RLD 68      // (multiple of 4: an accumulator fragment of the TRANSPOSED product is four consecutive columns of one row = one ds_write_b128, round 6)
// XCD-aware roles: workgroup i runs on XCD i % 8 (rec_chain.hpp), and every XCD fetches what its workgroups read into its own L2.
// Row part = (i % 8) / 2: the two XCDs of a part read only that part's 28 panel rows (a quarter of h_{t-1}) — with the column block
// as the fast index every XCD pulled the whole panel every step, eight copies of it over the fabric.  The column blocks are dealt
// to (i % 2, i / 8); a part's arrival line is polled from its two XCDs only.
#define DCF_NA(NN) (8 * (((((NN) + 63) >> 6) + 1) >> 1))
template <bool XF, bool LW = false, bool RP = false>
__global__ __launch_bounds__(256) void dec_chain_kernel(const DecChainArgs p) {
  static_assert(!(XF && LW), "the LDS-resident attn_W and the LDS frames 32..47 do not fit together");
  // phase A: K-partials of the [112 x 16] tile (LW: two buffers, summed in two stages — the room attn_W needs); RP: of the [32 x 64] tile
  __shared__ __attribute__((aligned(16))) float red[RP ? 2 * 32 * DCF_RLD : (LW ? 2 : 4) * RC_PAN_ROWS * DC_RED_LD];
  __shared__ __attribute__((aligned(16))) float spre[4 * 512];       // phase B: gate pre-activations
  __shared__ float swh[128];
  __shared__ __attribute__((aligned(16))) float sa[32 + DC_XF];
  extern __shared__ __attribute__((aligned(16))) float dc_dyn[];     // XF: P fragments of frames 32..47 (64 KB), then [DC_XF][128] fp32 Uv rows
  float* suvx = dc_dyn + 4 * 32 * 64 * 2;
  __shared__ __attribute__((aligned(16))) bf16_t hl[512];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int H = p.H, A = p.A, F = p.F, B = p.B, Bs = p.Bs, W4 = 4 * H, N = 4 * H + A;
  const int NN = LW ? W4 : N;                            // columns phase A produces
  const int NCB = (NN + 63) >> 6, NCBH = (NCB + 1) >> 1;  // RP: 64-column blocks, and half of them (below)
  const int NA = RP ? DCF_NA(NN) : (NN >> 4);
  const int wg = blockIdx.x;
  bf16_t* wlds = reinterpret_cast<bf16_t*>(dc_dyn);      // LW: attn_W as [k / 8][128 a][8] (an MFMA B fragment = 16 bytes per lane, 256 contiguous bytes per 16 lanes)
  if (LW) {
    for (int c = tid; c < 64 * 128; c += 256) {
      const int a = c & 127, kg = c >> 7;
      bf16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
      if (a < A && kg * 8 < H) v = *reinterpret_cast<const bf16x8*>(p.W + (size_t)(W4 + a) * p.ldw + kg * 8);
      *reinterpret_cast<bf16x8*>(wlds + (size_t)c * 8) = v;
    }
    for (int j = tid; j < 512; j += 256) hl[j] = (bf16_t)0.f;
  }
  const int cb = RP ? (wg & 1) * NCBH + (wg >> 3) : 0, part = RP ? (wg & 7) >> 1 : 0;
  const bool isA = wg < NA && (!RP || cb < NCB), isB = wg < B;
  const int kq = (lane >> 4) * 8;
  const size_t pan_t = rc_pan_elems(H);

  // ---- phase A residents: 16 weight rows x K = H (4 waves x 4 k-steps of 32); RP: 4 column groups of 16 rows
  const int own = RC_PAN_ROWS / DCF_PARTS, own_lo = part * own;
  const int r0 = own_lo < RC_PAN_ROWS - 32 ? own_lo : RC_PAN_ROWS - 32;
  bf16x8 wb[4][RP ? 4 : 1];
#pragma unroll
  for (int gq = 0; gq < (RP ? 4 : 1); ++gq) {
    const int n = RP ? (isA ? cb : 0) * 64 + gq * 16 + (lane & 15) : (isA ? wg : 0) * 16 + (lane & 15);
    const bf16_t* wrow = p.W + (size_t)(n < NN ? n : 0) * p.ldw + kq;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int k = wave * 128 + s * 32;
      wb[s][gq] = (k + kq < H && n < NN) ? *reinterpret_cast<const bf16x8*>(wrow + k) : bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
    }
  }
  if ((H & 31) && wg == 0) {     // zero the k-groups that pad H to a multiple of 32 in every step's panel (see rec_chain.hpp)
    const int pad0 = H >> 3, padn = (((H + 31) >> 5) << 2) - pad0;
    for (int t = 0; t < p.T; ++t)
      for (int j = tid; j < padn * RC_PAN_ROWS * 2; j += 256)
        __hip_atomic_store(reinterpret_cast<uint64_t*>(p.Pan + (size_t)t * pan_t + (size_t)pad0 * RC_PAN_ROWS * 8) + j, (uint64_t)0,
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  // ---- phase B residents (caption b = wg): P block, Uv rows, attention vectors; wave = gate g, lane = 8 units
  const int b = isB ? wg : 0, g = wave;
  const int u = lane * 8;
  const bool live = isB && u < H;
  const int col = g * H + u;
  // The caption's P block as MFMA B fragments: for column group cg (16 columns of this wave's gate block) lane l holds
  // P[b][8 (l >> 4) .. + 7][g H + 16 cg + (l & 15)] — the context (1/F) sum_f a_f P[b,f,:] is then one 16x16x32 MFMA per column
  // group against an A operand whose rows 0 / 1 are the attention weights split into a bf16 high and low part (31 VALU
  // multiply-adds per column pair before).  Frames 32 .. 47 (XF): 16x16x16 fragments in LDS, each lane its own 8 bytes.
  bf16x8 pb[32];
  typedef short dc_s4 __attribute__((ext_vector_type(4)));
  bf16x4* plf = reinterpret_cast<bf16x4*>(dc_dyn);                     // XF: [4 waves][32 cg][64 lanes] x 8 bytes
  // scores: thread = (frame tid % 32, attention columns [16 ko, 16 ko + 16), ko = tid / 32): 16 independent tanh per thread,
  // the eight partial sums of a frame are added through LDS — no wavefront reductions
  __shared__ float swab[256], sps[8 * 32 + 16 * DC_XF];
  const int sf = tid & 31, ko = tid >> 5;
  float uvq[16], cpre[2] = {0.f, 0.f}, hs_sum[2] = {0.f, 0.f};
  {
#pragma unroll
    for (int cg = 0; cg < 32; ++cg) {
      const int hc = cg * 16 + (lane & 15);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int f = (lane >> 4) * 8 + j;
        pb[cg][j] = (isB && hc < H && f < F) ? p.P[((size_t)b * F + f) * p.ldp + g * H + hc] : (bf16_t)0.f;
      }
      if (XF) {
        bf16x4 r;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int f = 32 + (lane >> 4) * 4 + j;
          r[j] = (isB && hc < H && f < F) ? p.P[((size_t)b * F + f) * p.ldp + g * H + hc] : (bf16_t)0.f;
        }
        plf[(g * 32 + cg) * 64 + lane] = r;
      }
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int k = 16 * ko + i;
      uvq[i] = (isB && sf < F && k < A) ? p.Uv[((size_t)b * F + sf) * A + k] : 0.f;
    }
    if (XF) {
      for (int idx = tid; idx < DC_XF * 128; idx += 256) {
        const int f = 32 + (idx >> 7), k = idx & 127;
        suvx[idx] = (isB && f < F && k < A) ? p.Uv[((size_t)b * F + f) * A + k] : 0.f;
      }
    }
    if (tid < 128) { swab[2 * tid] = tid < A ? p.w[tid] : 0.f; swab[2 * tid + 1] = tid < A ? p.ab[tid] : 0.f; }
  }
  const int lane_off = ((lane >> 4) * RC_PAN_ROWS + (lane & 15)) * 8;
  unsigned ph = 0;
  // Stamped hand-over (ll): a word is {fp32 value, stamp = launch epoch << 6 | t}, written by one 8-byte store, so the
  // consumer can poll the data itself: no acknowledgement wait, no flag, no barrier between phase A and phase B.  The
  // epoch (one more per launch, kept in device memory) makes the words of earlier launches stale.
  const unsigned ep0 = rc_epoch_read(p.epoch), ep = ep0 << 6, fb = ep0 << 7;
  if (p.master && wg == (int)gridDim.x - 1) {
    rc_master_loop(p.bar, p.bar + 256, (int)gridDim.x - 1, fb, (p.ll ? 1 : 2) * (p.T - 1));
    return;
  }
  // RP + partial: the B -> A hand-over inside a row part (the comment at dec_chain_bwd_kernel's wait_part has the measurements)
  const bool partial = RP && p.partial;
  auto wait_part = [&](unsigned target) {
    if (isA && tid < 64) {
      const int n = B - own_lo < own ? B - own_lo : own;
      if (n > 0) {
        const unsigned* f = p.rep + ((wg & 7) * DCF_PARTS + part) * 32 + (tid < n ? tid : n - 1);
        unsigned spin = 0;
        while (!__all((int)(__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target) >= 0)) { if (rc_give_up(p.bar, spin)) break; }
        if (RC_ACQUIRE_INV) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      }
    }
    __syncthreads();
  };
  auto arrive_part = [&](unsigned v) {      // the replicas of the part's two XCDs
    if (isB && tid < 2) __hip_atomic_store(p.rep + ((2 * (b / own) + tid) * DCF_PARTS + b / own) * 32 + b % own, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  };

  // LW: attention projection and scores of step tt, from this caption's own h_{tt-1} (hl, bf16) — no other workgroup involved
  auto lw_scores = [&](const int tt) {
    if (tt > 0) {
      // Wh[a] = sum_k attn_W[a][k] h[k]: 16x16x32 MFMAs with h in every row of the A operand (row 0 of the result is read);
      // wave = two 16-column groups of the 128 attention columns
      f32x4 w0 = {0.f, 0.f, 0.f, 0.f}, w1 = w0;
      const int cg0 = 2 * wave;
      // (every k-step unconditionally: attn_W in LDS and hl are zero beyond H — the run-time test per k-step made the compiler branch
      // around every pair of MFMAs and shuffle the accumulators between the register files, see dec_chain_bwd_kernel's da product)
#pragma unroll
      for (int k4 = 0; k4 < 16; k4 += 4) {
        bf16x8 fa[4], f0[4], f1[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int ks = k4 + j;
          fa[j] = *reinterpret_cast<const bf16x8*>(hl + ks * 32 + (lane >> 4) * 8);
          const bf16_t* wp = wlds + ((size_t)(ks * 4 + (lane >> 4)) * 128 + (lane & 15)) * 8;
          f0[j] = *reinterpret_cast<const bf16x8*>(wp + (cg0 * 16) * 8);
          f1[j] = *reinterpret_cast<const bf16x8*>(wp + (cg0 * 16 + 16) * 8);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          w0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[j], f0[j], w0, 0, 0, 0);
          w1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[j], f1[j], w1, 0, 0, 0);
        }
      }
      if (lane < 16) { swh[cg0 * 16 + lane] = w0[0]; swh[cg0 * 16 + 16 + lane] = w1[0]; }
    } else if (tid < 128) swh[tid] = 0.f;
    __syncthreads();
    if (tid < A) p.Wh[((size_t)tt * Bs + b) * A + tid] = swh[tid];
    {
      float s0 = 0.f, s1 = 0.f;
#pragma unroll
      for (int i = 0; i < 16; i += 2) {
        const int k = 16 * ko + i;
        const f32x4 wa = *reinterpret_cast<const f32x4*>(swab + 2 * k);          // (w_k, b_k, w_k+1, b_k+1)
        s0 += wa[0] * rn_tanh(swh[k] + wa[1] + uvq[i]);
        s1 += wa[2] * rn_tanh(swh[k + 1] + wa[3] + uvq[i + 1]);
      }
      sps[ko * 32 + sf] = s0 + s1;
    }
    __syncthreads();
    if (tid < 32) {
      const float sc = ((sps[tid] + sps[32 + tid]) + (sps[64 + tid] + sps[96 + tid])) + ((sps[128 + tid] + sps[160 + tid]) + (sps[192 + tid] + sps[224 + tid]));
      sa[tid] = tid < F ? sc : 0.f;
      if (tid < F && !p.softmax) p.att[((size_t)tt * Bs + b) * F + tid] = sc;
    }
    __syncthreads();
    if (p.softmax) { attn_softmax_lds(sa, F, p.att + ((size_t)tt * Bs + b) * F); __syncthreads(); }
  };
  if (LW && isB) { __syncthreads(); lw_scores(0); }
  for (int t = 0; t < p.T; ++t) {
    // input part of the gates of this step: independent of the chain, requested before any waiting
    f32x4 x0 = {0.f, 0.f, 0.f, 0.f}, x1 = x0;
    if (live) {
      const float* xe = p.Xe + ((size_t)t * Bs + b) * W4 + col;
      x0 = *reinterpret_cast<const f32x4*>(xe); x1 = *reinterpret_cast<const f32x4*>(xe + 4);
    }
    if (t > 0) {
      // ================= phase A: G1[t][:, 16 columns] = h_{t-1} . W^T
      DC_TS(0);
      if (isA) {
       if constexpr (RP) {
        const bf16_t* Ap = p.Pan + (size_t)(t - 1) * pan_t + lane_off + r0 * 8;
        bf16x8 fa[4][2];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          const int k = wave * 128 + s * 32;
#pragma unroll
          for (int i = 0; i < 2; ++i)
            fa[s][i] = *reinterpret_cast<const bf16x8*>(Ap + ((k < H ? (k >> 3) : 0) * RC_PAN_ROWS + i * 16) * 8);
        }
        __builtin_amdgcn_sched_barrier(0);
        f32x4 acc[2][4];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int gq = 0; gq < 4; ++gq) acc[i][gq] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) acc[i][gq] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb[s][gq], fa[s][i], acc[i][gq], 0, 0, 0);
        // (transposed product, round 6: the lane holds columns 4 (lane / 16) .. + 3 of row lane % 16 — one 16-byte LDS access per
        //  fragment where the row-major fragment took four scalar ones; same sums)
        const int rw = lane & 15, cq = (lane >> 4) * 4;
        // the four K quarters in two stages through two [32 x 64] buffers: waves 2, 3 store, waves 0, 1 add
        float* part_ = red + (wave & 1) * (32 * DCF_RLD);
        if (wave >= 2) {
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) *reinterpret_cast<f32x4*>(part_ + (i * 16 + rw) * DCF_RLD + gq * 16 + cq) = acc[i][gq];
        }
        __syncthreads();
        if (wave < 2) {
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) *reinterpret_cast<f32x4*>(part_ + (i * 16 + rw) * DCF_RLD + gq * 16 + cq) += acc[i][gq];
        }
        __syncthreads();
        DC_TS(1);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int idx = tid + j * 256, row = own_lo + (idx >> 5), pc = (idx & 31) * 2, rl = row - r0;
          if (idx < own * 32 && row < B && cb * 64 + pc < NN) {
            const float v0 = red[rl * DCF_RLD + pc] + red[32 * DCF_RLD + rl * DCF_RLD + pc];
            const float v1 = red[rl * DCF_RLD + pc + 1] + red[32 * DCF_RLD + rl * DCF_RLD + pc + 1];
            if (p.ll) {
              uint64_t* L = reinterpret_cast<uint64_t*>(p.G1) + ((size_t)t * B + row) * N + cb * 64 + pc;
              const unsigned st = ep | (unsigned)t;
              rc_store16f(reinterpret_cast<float*>(L), f32x4{v0, __builtin_bit_cast(float, st), v1, __builtin_bit_cast(float, st)});
            } else {
              union { float f[2]; uint64_t q; } pk; pk.f[0] = v0; pk.f[1] = v1;
              __hip_atomic_store(reinterpret_cast<uint64_t*>(p.G1 + ((size_t)t * B + row) * N + cb * 64 + pc), pk.q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
          }
        }
       } else {
        const bf16_t* Ap = p.Pan + (size_t)(t - 1) * pan_t + lane_off;
        bf16x8 fa[4][RC_MB];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          const int k = wave * 128 + s * 32;
#pragma unroll
          for (int i = 0; i < RC_MB; ++i)
            fa[s][i] = *reinterpret_cast<const bf16x8*>(Ap + ((k < H ? (k >> 3) : 0) * RC_PAN_ROWS + i * 16) * 8);
        }
        __builtin_amdgcn_sched_barrier(0);
        f32x4 acc[RC_MB];
#pragma unroll
        for (int i = 0; i < RC_MB; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
          for (int i = 0; i < RC_MB; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[s][i], wb[s][0], acc[i], 0, 0, 0);
        const int rr = (lane >> 4) * 4, cl = lane & 15;
        if (LW) {
          // waves 2, 3 hand their partials to waves 0, 1 through the two buffers, which then hold the two half sums
          float* part = red + (wave & 1) * (RC_PAN_ROWS * DC_RED_LD);
          if (wave >= 2) {
#pragma unroll
            for (int i = 0; i < RC_MB; ++i)
#pragma unroll
              for (int r = 0; r < 4; ++r) part[(i * 16 + rr + r) * DC_RED_LD + cl] = acc[i][r];
          }
          __syncthreads();
          if (wave < 2) {
#pragma unroll
            for (int i = 0; i < RC_MB; ++i)
#pragma unroll
              for (int r = 0; r < 4; ++r) part[(i * 16 + rr + r) * DC_RED_LD + cl] += acc[i][r];
          }
        } else {
          float* part = red + wave * (RC_PAN_ROWS * DC_RED_LD);
#pragma unroll
          for (int i = 0; i < RC_MB; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) part[(i * 16 + rr + r) * DC_RED_LD + cl] = acc[i][r];
        }
        __syncthreads();
        DC_TS(1);
        float* Gt = p.G1 + (size_t)t * B * N + wg * 16;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int idx = tid + j * 256, row = idx >> 3, pc = (idx & 7) * 2;
          if (idx < RC_PAN_ROWS * 8 && row < B) {
            float v0 = 0.f, v1 = 0.f;
#pragma unroll
            for (int w = 0; w < (LW ? 2 : 4); ++w) {
              v0 += red[w * (RC_PAN_ROWS * DC_RED_LD) + row * DC_RED_LD + pc];
              v1 += red[w * (RC_PAN_ROWS * DC_RED_LD) + row * DC_RED_LD + pc + 1];
            }
            if (p.ll) {
              uint64_t* L = reinterpret_cast<uint64_t*>(p.G1) + ((size_t)t * B + row) * N + wg * 16 + pc;
              const uint64_t st = (uint64_t)(ep | (unsigned)t) << 32;
              // two stamped words = 16 contiguous, 16-byte-aligned bytes: one store (each 8-byte word lies inside one 32-byte
              // sector, which is what its reader's 8-byte load observes as a unit — DC_STAMP_PAIR=0 restores two 8-byte atomics)
              if (DC_STAMP_PAIR) {
                rc_store16f(reinterpret_cast<float*>(L), f32x4{v0, __builtin_bit_cast(float, (unsigned)(st >> 32)), v1, __builtin_bit_cast(float, (unsigned)(st >> 32))});
              } else {
                __hip_atomic_store(L, st | __builtin_bit_cast(unsigned, v0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(L + 1, st | __builtin_bit_cast(unsigned, v1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              }
            } else {
              union { float f[2]; uint64_t q; } pk; pk.f[0] = v0; pk.f[1] = v1;
              __hip_atomic_store(reinterpret_cast<uint64_t*>(Gt + (size_t)row * N + pc), pk.q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
          }
        }
       }
        if (!p.ll) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      if (!p.ll) {
        __syncthreads();
        DC_TS(2);
        ++ph;
        rc_arrive(p.bar, fb + ph);
        { if (p.master) rc_wait_release(p.bar + 256, fb + ph); else rc_wait(p.bar, fb + ph); }
      }
      DC_TS(3);
    }
    // ================= phase B: caption b
    if (isB) {
      float pre[8];
      pre[0] = x0[0]; pre[1] = x0[1]; pre[2] = x0[2]; pre[3] = x0[3]; pre[4] = x1[0]; pre[5] = x1[1]; pre[6] = x1[2]; pre[7] = x1[3];
      float whv = 0.f;
      // (LW: the scores of this step are already in `sa` — lw_scores ran before the barrier wait of the step before; the context
      // MFMAs below come first and the gate pre-activations of phase A are polled after them)
      auto poll_pre = [&]() {
      if (t > 0 && p.ll) {
        // poll this caption's words until every stamp is this step's
        const uint64_t* L = reinterpret_cast<const uint64_t*>(p.G1) + ((size_t)t * B + b) * N;
        const uint64_t* lc = L + (live ? col : 0);
        const uint64_t* lw = LW ? lc : L + W4 + (tid < A ? tid : 0);      // (LW: phase A does not produce the attention columns)
        const unsigned want = ep | (unsigned)t;
        uint64_t wv[8], ww;
        unsigned spin = 0;
        for (;;) {
#pragma unroll
          for (int j = 0; j < 8; ++j) wv[j] = __hip_atomic_load(lc + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          ww = __hip_atomic_load(lw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          bool ok = (unsigned)(ww >> 32) == want;
#pragma unroll
          for (int j = 0; j < 8; ++j) ok = ok && (unsigned)(wv[j] >> 32) == want;
          if (__all(ok)) break;
          if (rc_give_up(p.bar, spin)) break;
        }
        if (live) {
#pragma unroll
          for (int j = 0; j < 8; ++j) pre[j] += __builtin_bit_cast(float, (unsigned)wv[j]);
        }
        if (tid < A) whv = __builtin_bit_cast(float, (unsigned)ww);
      } else if (t > 0) {
        const float* gr = p.G1 + ((size_t)t * B + b) * N;
        f32x4 g0 = {0.f, 0.f, 0.f, 0.f}, g1 = g0;
        if (live) { g0 = *reinterpret_cast<const f32x4*>(gr + col); g1 = *reinterpret_cast<const f32x4*>(gr + col + 4); }
        if (tid < A) whv = gr[W4 + tid];
        pre[0] += g0[0]; pre[1] += g0[1]; pre[2] += g0[2]; pre[3] += g0[3];
        pre[4] += g1[0]; pre[5] += g1[1]; pre[6] += g1[2]; pre[7] += g1[3];
      }
      };
      if (!LW) {
      poll_pre();
      // all 128 entries: the score threads read swh[k] for every k of their column range (w_k = 0 beyond A, but 0 x garbage
      // from uninitialised LDS could be NaN)
      if (tid < 128) swh[tid] = tid < A ? whv : 0.f;
      if (tid < A) p.Wh[((size_t)t * Bs + b) * A + tid] = whv;
      __syncthreads();
      DC_TS(4);
      {
        float s0 = 0.f, s1 = 0.f;
#pragma unroll
        for (int i = 0; i < 16; i += 2) {
          const int k = 16 * ko + i;
          const f32x4 wa = *reinterpret_cast<const f32x4*>(swab + 2 * k);          // (w_k, b_k, w_k+1, b_k+1)
          s0 += wa[0] * rn_tanh(swh[k] + wa[1] + uvq[i]);
          s1 += wa[2] * rn_tanh(swh[k + 1] + wa[3] + uvq[i + 1]);
        }
        sps[ko * 32 + sf] = s0 + s1;
        if (XF) {      // frames 32 .. 47: thread = (frame 32 + tid % 16, columns [8 kp, 8 kp + 8), kp = tid / 16)
          const int f2 = tid & 15, kp = tid >> 4;
          float e0 = 0.f;
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            const int k = 8 * kp + i;
            e0 += swab[2 * k] * rn_tanh(swh[k] + swab[2 * k + 1] + suvx[f2 * 128 + k]);
          }
          sps[256 + kp * 16 + f2] = e0;
        }
      }
      __syncthreads();
      if (tid < 32) {
        const float sc = ((sps[tid] + sps[32 + tid]) + (sps[64 + tid] + sps[96 + tid])) + ((sps[128 + tid] + sps[160 + tid]) + (sps[192 + tid] + sps[224 + tid]));
        sa[tid] = tid < F ? sc : 0.f;
        if (tid < F && !p.softmax) p.att[((size_t)t * Bs + b) * F + tid] = sc;
      } else if (XF && tid < 32 + DC_XF) {
        const int f2 = tid - 32;
        float sc = 0.f;
#pragma unroll
        for (int kp = 0; kp < 16; ++kp) sc += sps[256 + kp * 16 + f2];
        sa[tid] = tid < F ? sc : 0.f;
        if (tid < F && !p.softmax) p.att[((size_t)t * Bs + b) * F + tid] = sc;
      }
      __syncthreads();
      if (p.softmax) { attn_softmax_lds(sa, F, p.att + ((size_t)t * Bs + b) * F); __syncthreads(); }
      }      // !LW
      DC_TS(8);
      {
        // context of this wave's gate block: ctx[n] = sum_f a_f P[b, f, n] as MFMAs.  Rows 4i / 4i + 1 of the A operand are the
        // attention weights split into a bf16 high / low part (the other rows zero), so EVERY 16-lane group of the result holds
        // ctx of the 16 columns in d[0] + d[1]; four column groups per round, lane group q keeps group 4 r + q, and one
        // full-wave LDS store writes 64 consecutive columns (all 32 groups are computed: fragments beyond H are zero)
        bf16x8 av;
        const int q = lane >> 4, m4 = lane & 3;
        // branch-free: hm * hi + lm * (a - hi) is exactly hi (rows 4i), the low part (rows 4i + 1) or zero
        const float hm = m4 == 0 ? 1.f : 0.f, lm = m4 == 1 ? 1.f : 0.f;
        {
          const f32x4 a0 = *reinterpret_cast<const f32x4*>(sa + 8 * q), a1 = *reinterpret_cast<const f32x4*>(sa + 8 * q + 4);
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const float a = j < 4 ? a0[j] : a1[j - 4];
            const float hif = (float)(bf16_t)a;
            av[j] = (bf16_t)(hm * hif + lm * (a - hif));
          }
        }
        bf16x4 ax = {0, 0, 0, 0};
        if (XF) {
          const f32x4 a0 = *reinterpret_cast<const f32x4*>(sa + 32 + 4 * q);
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float hif = (float)(bf16_t)a0[j];
            ax[j] = (bf16_t)(hm * hif + lm * (a0[j] - hif));
          }
        }
        const float k0 = q == 0 ? 1.f : 0.f, k1 = q == 1 ? 1.f : 0.f, k2 = q == 2 ? 1.f : 0.f, k3 = q == 3 ? 1.f : 0.f;
        const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < 8; ++r) {
          f32x4 d0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, pb[4 * r], z4, 0, 0, 0);
          f32x4 d1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, pb[4 * r + 1], z4, 0, 0, 0);
          f32x4 d2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, pb[4 * r + 2], z4, 0, 0, 0);
          f32x4 d3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, pb[4 * r + 3], z4, 0, 0, 0);
          if (XF) {
            const dc_s4 axs = __builtin_bit_cast(dc_s4, ax);
            d0 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(axs, __builtin_bit_cast(dc_s4, plf[(g * 32 + 4 * r) * 64 + lane]), d0, 0, 0, 0);
            d1 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(axs, __builtin_bit_cast(dc_s4, plf[(g * 32 + 4 * r + 1) * 64 + lane]), d1, 0, 0, 0);
            d2 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(axs, __builtin_bit_cast(dc_s4, plf[(g * 32 + 4 * r + 2) * 64 + lane]), d2, 0, 0, 0);
            d3 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(axs, __builtin_bit_cast(dc_s4, plf[(g * 32 + 4 * r + 3) * 64 + lane]), d3, 0, 0, 0);
          }
          const float c0 = d0[0] + d0[1], c1 = d1[0] + d1[1], c2 = d2[0] + d2[1], c3 = d3[0] + d3[1];
          spre[g * 512 + r * 64 + lane] = (k0 * c0 + k1 * c1) + (k2 * c2 + k3 * c3);
        }
      }
      if (LW) poll_pre();      // the gate pre-activations of phase A: they travelled while the context was formed
      if (live) {        // (the wave reads back what it wrote itself: LDS accesses of a wave are in order)
        const float invF = 1.0f / (float)F;
        float* dst = spre + g * 512 + lane * 8;
        const f32x4 c0 = *reinterpret_cast<const f32x4*>(dst), c1 = *reinterpret_cast<const f32x4*>(dst + 4);
        *reinterpret_cast<f32x4*>(dst) = f32x4{pre[0] + c0[0] * invF, pre[1] + c0[1] * invF, pre[2] + c0[2] * invF, pre[3] + c0[3] * invF};
        *reinterpret_cast<f32x4*>(dst + 4) = f32x4{pre[4] + c1[0] * invF, pre[5] + c1[1] * invF, pre[6] + c1[2] * invF, pre[7] + c1[3] * invF};
      }
      __syncthreads();
      DC_TS(9);
      float hv[2], av[2][4], cn[2];
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) {
        const int uu = tid + 256 * jj;
        hv[jj] = 0.f; cn[jj] = 0.f; av[jj][0] = av[jj][1] = av[jj][2] = av[jj][3] = 0.f;
        if (uu < H) {
          if (p.gru) {
            const GruOut r = gru_point(spre[uu], spre[512 + uu], spre[1024 + uu], spre[1536 + uu], cpre[jj]);
            hv[jj] = r.h; av[jj][0] = r.r; av[jj][1] = r.z; av[jj][2] = r.n; av[jj][3] = r.hn; cpre[jj] = r.h;
          } else {
            const LstmOut r = lstm_point(spre[uu], spre[512 + uu], spre[1024 + uu], spre[1536 + uu], cpre[jj]);
            hv[jj] = r.h; av[jj][0] = r.i; av[jj][1] = r.f; av[jj][2] = r.g; av[jj][3] = r.o; cpre[jj] = r.c; cn[jj] = r.c;
          }
          hl[uu] = (bf16_t)hv[jj];
          hs_sum[jj] += hv[jj];
        }
      }
      __syncthreads();
      DC_TS(5);
      // publish h_t[b]: 16 bytes per k-group, written through
      if (tid < (H >> 3)) {
        rc_store16(p.Pan + (size_t)t * pan_t + ((size_t)tid * RC_PAN_ROWS + b) * 8, hl + tid * 8);
      }
      if (t + 1 < p.T) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      // (the stores below are issued after the arrive, see the end of the loop body)
      if (t + 1 < p.T) { __syncthreads(); DC_TS(6); ++ph; if (partial) arrive_part(fb + ph); else rc_arrive(p.bar, fb + ph); }
      bf16_t* Lt = p.Hlp + ((size_t)t * Bs + b) * p.ld_hlp;
      if (tid < (H >> 3)) *reinterpret_cast<bf16x8*>(Lt + tid * 8) = *reinterpret_cast<const bf16x8*>(hl + tid * 8);
      if (p.Xcat && tid < (H >> 3)) *reinterpret_cast<bf16x8*>(p.Xcat + ((size_t)t * Bs + b) * p.ld_xcat + tid * 8) = *reinterpret_cast<const bf16x8*>(hl + tid * 8);
      for (int j = H + tid; j < p.ld_hlp; j += 256) Lt[j] = (bf16_t)0.f;
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) {
        const int uu = tid + 256 * jj;
        if (uu < H) {
          const size_t o = ((size_t)t * Bs + b) * H + uu;
          p.Hs[o] = hv[jj];
          if (!p.gru) p.Cs[o] = cn[jj];
          float* a = p.acts + ((size_t)t * Bs + b) * W4 + uu;
          a[0] = av[jj][0]; a[H] = av[jj][1]; a[2 * H] = av[jj][2]; a[3 * H] = av[jj][3];
        }
      }
      if (LW && t + 1 < p.T) lw_scores(t + 1);      // from hl = h_t, while the barrier completes
      if (t + 1 < p.T) { if (partial) wait_part(fb + ph); else if (p.master) rc_wait_release(p.bar + 256, fb + ph); else rc_wait(p.bar, fb + ph); }
      DC_TS(7);
    } else if (t + 1 < p.T) {
      __syncthreads();
      ++ph;
      if (!partial) rc_arrive(p.bar, fb + ph);
      { if (partial) wait_part(fb + ph); else if (p.master) rc_wait_release(p.bar + 256, fb + ph); else rc_wait(p.bar, fb + ph); }
    }
  }
  if (partial) {      // the launch epoch may only move once every workgroup has read it: one full barrier at the end
    __syncthreads();
    rc_arrive(p.bar, fb + (unsigned)p.T);
    if (wg == 0) rc_wait(p.bar, fb + (unsigned)p.T);
  }
  if (p.mp && isB) {
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) { const int uu = tid + 256 * jj; if (uu < H) p.mp[(size_t)b * H + uu] = hs_sum[jj] * p.mp_scale; }
    if (p.Xcat) {       // the mean-pooled half of the reconstructor's input, one dropout mask per step
      const uint32_t key = drop_key(p.xdd);
      for (int t = 0; t < p.T; ++t) {
        bf16_t* x = p.Xcat + ((size_t)t * Bs + b) * p.ld_xcat + H;
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
          const int uu = tid + 256 * jj;
          if (uu < H) x[uu] = (bf16_t)(hs_sum[jj] * p.mp_scale * drop_at(p.xdd, key, t, b, H, uu));
        }
      }
    }
  }
  rc_epoch_bump(p.epoch, ep0);
  rc_poison(p.bar, p.poison);
}

// =============================================================================================
// The decoder's BPTT chain as one launch.  Per step (t = T-1 .. 0):
//   dh_t = dHs[t] + dhid[t] + [dgates_{t+1} | dWh_{t+1}] . [W_hh ; attn_W]        (phase A', unit-owner workgroups)
//   cell backward -> dgates_t ; da = (1/F) P_b . dgates_t ; attention backward -> dWh_t, dUv += , dw +=     (phase B')
// Phase A': workgroup (16 output units, one of 4 row parts) keeps its 16 rows of [W_hh ; attn_W]^T (K = 4H + A
// contiguous, from WcombT) in registers and reads the bf16 rows [dgates | dWh] of the step before from the exchange panel.
// Phase B': workgroup b owns caption b.  Its block of P sits in registers as MFMA A-operand fragments (frames x gate
// columns), so da is 2 x H/32 MFMAs per wave against the bf16 dgates broadcast over the 16 B-columns; Uv, the dUv and dw
// accumulators and the dc carry stay in registers for all T steps (the per-launch path re-read and re-wrote dUv every
// step and split each caption over four workgroups whose dWh partials the next GEMM had to sum).
// Output rows keep the layout of the per-launch path, [dgates (4H) | dWh (A) | 0 (3A) | pad], so the deferred
// weight-gradient GEMMs are unchanged.
struct DecChainBwdArgs {
  int T, B, F, H, A, gru;
  int Bs;                          // see DecChainArgs
  const bf16_t* Wt; int ldwt;      // [H][ldwt] WcombT: column n of [W_hh ; attn_W ..], K contiguous
  const bf16_t* P; int ldp;        // [B][F][ldp]
  const float* Uv; const float* ab; const float* w;
  const float* dHs; const float* dHs2;                      // [T][B][H]; dHs2 may be null
  const float* acts; const float* Cs; const float* Hs;      // [T][B][4H], [T][B][H], [T][B][H]
  const float* Wh;                 // [T][B][A]
  const float* att; int softmax;   // softmax mode: the saved attention weights [T][B][F]
  float* G2;                       // [T][B][DCB_KS][H] exchange (by chain step): the K parts of the recurrent part of dh (ll: stamped 8-byte words)
  unsigned* epoch; int ll; int master; float* poison;
  int partial;                     // 1 (needs ll, excludes master): phase A' waits for the captions of ITS row part only (below)
  unsigned* rep;                   // partial: [8 XCDs][DCB_PARTS][32] arrival words, a 128-byte line per (XCD, row part)
  bf16_t* Pan;                     // [T][rc_pan_elems(4H + A)] exchange (by chain step): rows [dgates | dWh]
  bf16_t* dGx; int ld_dgx;         // [T][B][ld_dgx]
  float* dUv; bf16_t* dUv_lp; int ld_dUv;                   // [B][F][A], [B F][ld_dUv]
  float* dwacc;                    // [RN_FCH][B][A]
  unsigned* bar;
  unsigned long long* ts;          // probe builds only: [T][12] stamps of workgroup 0
};

#ifdef DC_PROBE_TS
#define DCB_TS(i) do { if (wg == DC_PROBE_WG && tid == 0) p.ts[(size_t)s * 12 + (i)] = wall_clock64(); } while (0)
#else
#define DCB_TS(i) do { } while (0)
#endif
#define DCB_RB 2              // 32 rows per workgroup in phase A', 4 row parts
#define DCB_PARTS 4
// Phase A' tiling (round 5): workgroup = (64 output units, one of 4 row parts, one of DCB_KS parts of K = 4H + A) instead of
// (16 units, row part, all of K) — the same number of workgroups (128 at H = 512) and 40 instead of 34 MFMAs per wave, but a
// workgroup pulls 32 rows x K / 4 of the panel through its CU's 64 B/clk L1 path per step instead of 32 rows x K (35 KB instead of
// 139 KB: the forward chain's phase A went from 2.16 to 1.22 us per step with the same change of ratio).  The DCB_KS partial sums of
// a (row, unit) are separate stamped words; phase B' polls all of them and adds them in K order.
#define DCB_KS 4
#define DCB_WS 5              // k32-steps per wave: 4 waves x 5 >= ceil(68 / DCB_KS) = 17 k-steps per K part at H = 512, A = 128
#define DCB_RLD 68      // (as DCF_RLD)
#define DCB_NA(H) ((((H) + 63) >> 6) * DCB_PARTS * DCB_KS)

// XF: frames 32 .. 47 — the third 16-frame block of P as MFMA A fragments read from (dynamic) LDS (rows padded by 8
// elements: the 16 lanes of a fragment load then fall into different banks), their Uv rows and dUv accumulators in LDS.
#define DCB_PLD (4 * 512 + 8)
template <bool XF>
__global__ __launch_bounds__(256) void dec_chain_bwd_kernel(const DecChainBwdArgs p) {
  __shared__ __attribute__((aligned(16))) float red[2 * 32 * DCB_RLD];      // phase A': the K quarters of the workgroup's [32 x 64] tile, summed in two stages
  __shared__ __attribute__((aligned(16))) bf16_t srow[4 * 512 + 128 + 64];   // [dgates | dWh] of this step (+ zero tail)
  __shared__ float spartf[4 * (32 + DC_XF)], sda[32 + DC_XF], spart[256];
  extern __shared__ __attribute__((aligned(16))) float dc_dyn[];     // XF: [DC_XF][DCB_PLD] bf16, [DC_XF][128] Uv, [DC_XF][128] dUv
  bf16_t* plx = reinterpret_cast<bf16_t*>(dc_dyn);
  float* suvx = dc_dyn + DC_XF * DCB_PLD / 2;
  float* sdux = suvx + DC_XF * 128;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int H = p.H, A = p.A, F = p.F, B = p.B, Bs = p.Bs, W4 = 4 * H, KA = 4 * H + A;
  const int NUB = (H + 63) >> 6, NA = DCB_NA(H);
  const int wg = blockIdx.x;
  const bool isA = wg < NA, isB = wg < B;
  const int kq = (lane >> 4) * 8;
  const size_t pan_t = rc_pan_elems(KA);

  // ---- phase A' residents
  // XCD-aware roles (see DCF_NA): row part = (wg % 8) / 2, so the two XCDs of a part fetch 2 / 16 of the panel (its rows, half of K each)
  // instead of all of it; K part = 2 (wg % 2) + (wg / 8) / NUB, unit block = (wg / 8) % NUB
  const int ub = isA ? (wg >> 3) % NUB : 0, part = isA ? (wg & 7) >> 1 : 0, kp = isA ? ((wg & 1) << 1) | ((wg >> 3) / NUB) : 0;
  const int own = RC_PAN_ROWS / DCB_PARTS, own_lo = part * own;
  const int r0 = own_lo < RC_PAN_ROWS - DCB_RB * 16 ? own_lo : RC_PAN_ROWS - DCB_RB * 16;
  const int KT = (KA + 31) >> 5, KPP = (KT + DCB_KS - 1) / DCB_KS;      // k32-steps in all / per K part
  // slot j of this wave = k-step kp KPP + wave + 4 j of the K part (-1: none)
  auto kstep = [&](int j) { const int o = wave + 4 * j; return (o < KPP && kp * KPP + o < KT) ? kp * KPP + o : -1; };
  bf16x8 wb[DCB_WS][4];
#pragma unroll
  for (int gq = 0; gq < 4; ++gq) {
    const int n = ub * 64 + gq * 16 + (lane & 15);
    const bf16_t* wrow = p.Wt + (size_t)(n < H ? n : 0) * p.ldwt + kq;
#pragma unroll
    for (int j = 0; j < DCB_WS; ++j) {
      const int ks = kstep(j), k = ks * 32;
      wb[j][gq] = (ks >= 0 && n < H && k + kq < KA) ? *reinterpret_cast<const bf16x8*>(wrow + k) : bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
    }
  }
  if ((KA & 31) && wg == 0) {    // zero the k-groups that pad KA to a multiple of 32 in every step's panel
    const int pad0 = KA >> 3, padn = (((KA + 31) >> 5) << 2) - pad0;
    for (int t = 0; t < p.T; ++t)
      for (int j = tid; j < padn * RC_PAN_ROWS * 2; j += 256)
        __hip_atomic_store(reinterpret_cast<uint64_t*>(p.Pan + (size_t)t * pan_t + (size_t)pad0 * RC_PAN_ROWS * 8) + j, (uint64_t)0,
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  // ---- phase B' residents (caption b = wg)
  const int b = isB ? wg : 0, g = wave;
  // P as MFMA A-operand fragments: rows = frames (2 blocks of 16), K = this wave's gate block [g H, (g+1) H)
  bf16x8 pa[2][16];
#pragma unroll
  for (int fb = 0; fb < 2; ++fb)
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {
      const int f = fb * 16 + (lane & 15), ku = ks * 32 + kq;
      pa[fb][ks] = (isB && f < F && ku < H) ? *reinterpret_cast<const bf16x8*>(p.P + ((size_t)b * F + f) * p.ldp + g * H + ku)
                                            : bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
    }
  // (f, k) plane: k = tid % A, frame group gi = tid / A, G = 256 / A groups (A <= 128 -> G >= 2)
  const int G = 256 / A, kk = tid % A, gi = tid / A;
  const bool fk_on = isB && gi < G;
  float uvr[16], duv[16], dwa = 0.f;
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    const int f = gi + q * G;
    uvr[q] = (fk_on && f < F) ? p.Uv[((size_t)b * F + f) * A + kk] : 0.f;
    duv[q] = 0.f;
  }
  const float abk = fk_on ? p.ab[kk] : 0.f, wk = fk_on ? p.w[kk] : 0.f;
  if (XF) {
    for (int idx = tid; idx < DC_XF * (W4 >> 3); idx += 256) {          // 16-byte chunks of the caption's rows 32 .. 47
      const int fx = idx / (W4 >> 3), c8 = (idx % (W4 >> 3)) * 8;
      *reinterpret_cast<bf16x8*>(plx + (size_t)fx * DCB_PLD + c8) =
          (isB && 32 + fx < F) ? *reinterpret_cast<const bf16x8*>(p.P + ((size_t)b * F + 32 + fx) * p.ldp + c8) : bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
    }
    for (int idx = tid; idx < DC_XF * 128; idx += 256) {
      const int fx = idx >> 7, k = idx & 127;
      suvx[idx] = (isB && 32 + fx < F && k < A) ? p.Uv[((size_t)b * F + 32 + fx) * A + k] : 0.f;
      sdux[idx] = 0.f;
    }
    __syncthreads();
  }
  float carry[2] = {0.f, 0.f};
  const int lane_off = ((lane >> 4) * RC_PAN_ROWS + r0 + (lane & 15)) * 8;
  for (int j = tid; j < 64; j += 256) srow[W4 + 128 + j] = (bf16_t)0.f;
  unsigned ph = 0;
  const unsigned ep0 = rc_epoch_read(p.epoch), ep = ep0 << 6, fb = ep0 << 7;   // see rec_chain.hpp
  if (p.master && wg == (int)gridDim.x - 1) {
    rc_master_loop(p.bar, p.bar + 256, (int)gridDim.x - 1, fb, (p.ll ? 1 : 2) * (p.T - 1));
    return;
  }
  // Partial hand-over B' -> A' (round 5).  A phase-A' workgroup reads only the `own` = 28 panel rows of its row part, i.e. the rows
  // [dgates | dWh] of 28 captions, and phase B' of a caption polls the stamped words of the 32 unit groups of ITS part: the
  // dependencies close inside a row part, so nothing needs the whole grid.  Arrive -> master sees all 128 -> release word -> waiter
  // becomes arrive -> waiter: one memory round trip instead of two, and a part does not wait for the stragglers of the other three.
  // A caption writes its arrival into one replica of its part's flag line per XCD that hosts the part's phase-A' workgroups (p.rep:
  // [8][DCB_PARTS][32] words, a 128-byte line per (XCD, part)); a waiter polls the replica of its own XCD (blockIdx % 8) with one wave load.
  // With ONE copy polled by all 128 workgroups the step got 0.9 us LONGER (measured: 315 against 288 us per launch) — a line that is
  // written from eight XCDs and read from eight is the worst case for the L2s; in this form every line has its readers in one XCD,
  // like the arrival flags (read by the master only) and the release words (one line per XCD) of the relayed barrier.
  // Workgroups without a phase A' do not wait.
  auto wait_part = [&](unsigned target) {
    if (isA && tid < 64) {
      const int n = B - own_lo < own ? B - own_lo : own;
      if (n > 0) {
        const unsigned* f = p.rep + ((wg & 7) * DCB_PARTS + part) * 32 + (tid < n ? tid : n - 1);
        unsigned spin = 0;
        while (!__all((int)(__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target) >= 0)) { if (rc_give_up(p.bar, spin)) break; }
        if (RC_ACQUIRE_INV) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      }
    }
    __syncthreads();
  };
  auto arrive_part = [&](unsigned v) {       // caption b = wg: its part, its slot in the part's line, the replicas of the part's two XCDs
    if (isB && tid < 2) __hip_atomic_store(p.rep + ((2 * (b / own) + tid) * DCB_PARTS + b / own) * 32 + b % own, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  };

  // saved tensors of step t for this thread's two units, and Wh[t][b][kk]
  float d1[2], d2[2], av[2][4], cv[2], cpv[2], whk;
  auto prefetch = [&](int t) {
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int u = tid + 256 * q;
      const size_t o = ((size_t)t * Bs + b) * H + (u < H ? u : 0);
      d1[q] = p.dHs[o];
      d2[q] = p.dHs2 ? p.dHs2[o] : 0.f;
      const float* a = p.acts + ((size_t)t * Bs + b) * W4 + (u < H ? u : 0);
#pragma unroll
      for (int j = 0; j < 4; ++j) av[q][j] = a[(size_t)j * H];
      cv[q] = p.Cs[o];
      cpv[q] = t > 0 ? (p.gru ? p.Hs : p.Cs)[o - (size_t)Bs * H] : 0.f;
    }
    whk = p.Wh[((size_t)t * Bs + b) * A + kk];
  };
  if (isB) prefetch(p.T - 1);

  for (int s = 0; s < p.T; ++s) {
    const int t = p.T - 1 - s;
    if (s > 0) {
      // ================= phase A': G2[s][rows, 16 units] = rows_{s-1} . W
      DCB_TS(0);
      if (isA) {
        const bf16_t* Ap = p.Pan + (size_t)(s - 1) * pan_t + lane_off;
        bf16x8 fa[DCB_WS][DCB_RB];
#pragma unroll
        for (int j = 0; j < DCB_WS; ++j) {
          const int ks = kstep(j);
#pragma unroll
          for (int i = 0; i < DCB_RB; ++i)
            fa[j][i] = *reinterpret_cast<const bf16x8*>(Ap + ((ks >= 0 ? ks * 4 : 0) * RC_PAN_ROWS + i * 16) * 8);
        }
        __builtin_amdgcn_sched_barrier(0);
        f32x4 acc[DCB_RB][4];
#pragma unroll
        for (int i = 0; i < DCB_RB; ++i)
#pragma unroll
          for (int gq = 0; gq < 4; ++gq) acc[i][gq] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < DCB_WS; ++j)
#pragma unroll
          for (int i = 0; i < DCB_RB; ++i)
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) acc[i][gq] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb[j][gq], fa[j][i], acc[i][gq], 0, 0, 0);
        const int rw = lane & 15, cq = (lane >> 4) * 4;      // transposed product: four consecutive columns of one row per fragment (dec_chain_kernel)
        float* part_ = red + (wave & 1) * (32 * DCB_RLD);      // waves 2, 3 store, waves 0, 1 add
        if (wave >= 2) {
#pragma unroll
          for (int i = 0; i < DCB_RB; ++i)
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) *reinterpret_cast<f32x4*>(part_ + (i * 16 + rw) * DCB_RLD + gq * 16 + cq) = acc[i][gq];
        }
        __syncthreads();
        if (wave < 2) {
#pragma unroll
          for (int i = 0; i < DCB_RB; ++i)
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) *reinterpret_cast<f32x4*>(part_ + (i * 16 + rw) * DCB_RLD + gq * 16 + cq) += acc[i][gq];
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int idx = tid + j * 256, rg = own_lo + (idx >> 5), pc = (idx & 31) * 2, rl = rg - r0, un = ub * 64 + pc;
          if (idx < own * 32 && rg < B && un < H) {
            const float v0 = red[rl * DCB_RLD + pc] + red[32 * DCB_RLD + rl * DCB_RLD + pc];
            const float v1 = red[rl * DCB_RLD + pc + 1] + red[32 * DCB_RLD + rl * DCB_RLD + pc + 1];
            const size_t widx = (((size_t)s * B + rg) * DCB_KS + kp) * H + un;
            if (p.ll) {
              // two stamped words = 16 contiguous, 16-byte-aligned bytes: one store (each 8-byte word lies inside one 32-byte
              // sector, which is what its reader's 8-byte load observes as a unit)
              const unsigned st = ep | (unsigned)s;
              rc_store16f(reinterpret_cast<float*>(reinterpret_cast<uint64_t*>(p.G2) + widx), f32x4{v0, __builtin_bit_cast(float, st), v1, __builtin_bit_cast(float, st)});
            } else {
              union { float f[2]; uint64_t q; } pk; pk.f[0] = v0; pk.f[1] = v1;
              __hip_atomic_store(reinterpret_cast<uint64_t*>(p.G2 + widx), pk.q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
          }
        }
        if (!p.ll) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      if (!p.ll) {
        __syncthreads();
        ++ph;
        rc_arrive(p.bar, fb + ph);
        { if (p.master) rc_wait_release(p.bar + 256, fb + ph); else rc_wait(p.bar, fb + ph); }
      }
    }
    // ================= phase B': caption b
    if (isB) {
      // (1) cell backward of the thread's two units -> dgates (bf16) into the row buffer
      DCB_TS(1);
      float grec[2] = {0.f, 0.f};
      if (s > 0 && p.ll) {
        const uint64_t* L = reinterpret_cast<const uint64_t*>(p.G2) + ((size_t)s * B + b) * DCB_KS * H;
        const uint64_t* l0 = L + (tid < H ? tid : 0);
        const uint64_t* l1 = L + (tid + 256 < H ? tid + 256 : 0);
        const unsigned want = ep | (unsigned)s;
        uint64_t w0[DCB_KS], w1[DCB_KS];
        unsigned spin = 0;
        for (;;) {
          bool ok = true;
#pragma unroll
          for (int kk2 = 0; kk2 < DCB_KS; ++kk2) {
            w0[kk2] = __hip_atomic_load(l0 + (size_t)kk2 * H, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            w1[kk2] = __hip_atomic_load(l1 + (size_t)kk2 * H, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
#pragma unroll
          for (int kk2 = 0; kk2 < DCB_KS; ++kk2) ok = ok && (unsigned)(w0[kk2] >> 32) == want && (unsigned)(w1[kk2] >> 32) == want;
          if (__all(ok)) break;
          if (rc_give_up(p.bar, spin)) break;
        }
        // (K order, whoever arrived last)
        grec[0] = (__builtin_bit_cast(float, (unsigned)w0[0]) + __builtin_bit_cast(float, (unsigned)w0[1])) + (__builtin_bit_cast(float, (unsigned)w0[2]) + __builtin_bit_cast(float, (unsigned)w0[3]));
        grec[1] = (__builtin_bit_cast(float, (unsigned)w1[0]) + __builtin_bit_cast(float, (unsigned)w1[1])) + (__builtin_bit_cast(float, (unsigned)w1[2]) + __builtin_bit_cast(float, (unsigned)w1[3]));
      } else if (s > 0) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const int u = tid + 256 * q;
          if (u < H) {
            const float* gp = p.G2 + ((size_t)s * B + b) * DCB_KS * H + u;
            grec[q] = (gp[0] + gp[H]) + (gp[2 * (size_t)H] + gp[3 * (size_t)H]);
          }
        }
      }
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int u = tid + 256 * q;
        if (u < H) {
          float dh = d1[q] + d2[q] + grec[q];
          const LstmGrad gr = p.gru ? gru_point_bwd(dh + carry[q], av[q][0], av[q][1], av[q][2], av[q][3], cpv[q])
                                    : lstm_point_bwd(dh, carry[q], av[q][0], av[q][1], av[q][2], av[q][3], cv[q], cpv[q]);
          carry[q] = gr.dc_prev;
          srow[u] = (bf16_t)gr.di; srow[H + u] = (bf16_t)gr.df; srow[2 * H + u] = (bf16_t)gr.dg; srow[3 * H + u] = (bf16_t)gr.d_o;
        }
      }
      __syncthreads();
      // (2) da[f] = (1/F) sum_n P[b,f,n] dgates[n]: MFMA, this wave's gate block, dgates replicated over the 16 columns
      DCB_TS(2);
      {
        constexpr int NFB = XF ? 3 : 2, SPF = 32 + (XF ? DC_XF : 0);
        f32x4 acc[NFB];
#pragma unroll
        for (int fb = 0; fb < NFB; ++fb) acc[fb] = f32x4{0.f, 0.f, 0.f, 0.f};
        // Every k-step unconditionally (round 5): the P fragments beyond H are zeros and the dgates operand falls on the row buffer's
        // zero tail there, so a skipped step and an executed one give the same sums.  The run-time test `ks * 32 < H` around each
        // pair of MFMAs made the compiler branch per k-step and move both accumulators between the register files around EVERY
        // MFMA (s_nop 7 + 8 v_accvgpr moves, the LDS read waited for in between): 2.1 us per step for 32 MFMAs — a fifth of the BPTT
        // step.  Straight-line: the 16 LDS reads in two batches, two independent accumulation chains.
#pragma unroll
        for (int k8 = 0; k8 < 16; k8 += 8) {
          bf16x8 bv[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const int ku = (k8 + j) * 32 + kq;
            bv[j] = *reinterpret_cast<const bf16x8*>(srow + (ku < H ? g * H + ku : W4 + 128));
          }
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const int ks = k8 + j, ku = ks * 32 + kq;
            acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pa[0][ks], bv[j], acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pa[1][ks], bv[j], acc[1], 0, 0, 0);
            if (XF) {
              const bf16x8 px = ku < H ? *reinterpret_cast<const bf16x8*>(plx + (size_t)(lane & 15) * DCB_PLD + g * H + ku) : bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
              acc[NFB - 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(px, bv[j], acc[NFB - 1], 0, 0, 0);
            }
          }
        }
        if ((lane & 15) == 0) {
#pragma unroll
          for (int fb = 0; fb < NFB; ++fb)
#pragma unroll
            for (int r = 0; r < 4; ++r) spartf[g * SPF + fb * 16 + (lane >> 4) * 4 + r] = acc[fb][r];
        }
      }
      __syncthreads();
      {
        constexpr int SPF = 32 + (XF ? DC_XF : 0);
        if (tid < SPF) sda[tid] = (spartf[tid] + spartf[SPF + tid] + spartf[2 * SPF + tid] + spartf[3 * SPF + tid]) * (1.0f / (float)F);
      }
      __syncthreads();
      if (p.softmax) { attn_softmax_bwd_lds(sda, p.att + ((size_t)t * Bs + b) * F, F); __syncthreads(); }
      // (3) attention backward on the (f, k) plane
      DCB_TS(3);
      float dwh = 0.f;
      if (fk_on) {
        const float wh = whk + abk;
        float dw = 0.f;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const int f = gi + q * G;
          if (f < F) {
            const float tz = rn_tanh(wh + uvr[q]);
            const float daf = sda[f];
            const float ds = daf * wk * (1.f - tz * tz);
            dw += daf * tz;
            dwh += ds;
            duv[q] += ds;
          }
        }
        if (XF) {      // frames beyond the 16 register slots of this thread (only when G <= 2, i.e. A > 85)
          for (int q = 16; gi + q * G < F; ++q) {
            const int f = gi + q * G;
            if (f >= 32) {
              const float tz = rn_tanh(wh + suvx[(f - 32) * 128 + kk]);
              const float daf = sda[f];
              const float ds = daf * wk * (1.f - tz * tz);
              dw += daf * tz;
              dwh += ds;
              sdux[(f - 32) * 128 + kk] += ds;
            }
          }
        }
        dwa += dw;
        spart[gi * A + kk] = dwh;
      }
      __syncthreads();
      if (tid < A) {
        float a = 0.f;
        for (int j = 0; j < G; ++j) a += spart[j * A + tid];
        srow[W4 + tid] = (bf16_t)a;
      }
      __syncthreads();
      // (4) publish the row [dgates | dWh]: 16 bytes per k-group, written through
      DCB_TS(4);
      const bool more = s + 1 < p.T;
      for (int kg = tid; kg < (KA >> 3); kg += 256) {
        rc_store16(p.Pan + (size_t)s * pan_t + ((size_t)kg * RC_PAN_ROWS + b) * 8, srow + kg * 8);
      }
      if (more) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __syncthreads(); ++ph; if (p.partial) arrive_part(fb + ph); else rc_arrive(p.bar, fb + ph); }
      DCB_TS(5);
      // ---- off the critical path: the row-major copy [dgates | dWh | 0 ..] for the deferred GEMMs
      bf16_t* Gt = p.dGx + ((size_t)t * Bs + b) * p.ld_dgx;
      for (int kg = tid; kg < (p.ld_dgx >> 3); kg += 256)
        *reinterpret_cast<bf16x8*>(Gt + kg * 8) = kg < (KA >> 3) ? *reinterpret_cast<const bf16x8*>(srow + kg * 8) : bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
      if (more) { prefetch(t - 1); { if (p.partial) wait_part(fb + ph); else if (p.master) rc_wait_release(p.bar + 256, fb + ph); else rc_wait(p.bar, fb + ph); } }
      DCB_TS(6);
    } else if (s + 1 < p.T) {
      __syncthreads();
      ++ph;
      if (!p.partial) rc_arrive(p.bar, fb + ph);
      { if (p.partial) wait_part(fb + ph); else if (p.master) rc_wait_release(p.bar + 256, fb + ph); else rc_wait(p.bar, fb + ph); }
    }
  }
  if (p.partial) {
    // the launch epoch may only move once every workgroup has read it: a full barrier, once, at the end (phase T: above every step's)
    __syncthreads();
    rc_arrive(p.bar, fb + (unsigned)p.T);
    if (wg == 0) rc_wait(p.bar, fb + (unsigned)p.T);
  }
  // ---- the accumulators: dUv (+ operand copy, zero padded), dw
  if (isB) {
    if (fk_on) {
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int f = gi + q * G;
        if (f < F) {
          p.dUv[((size_t)b * F + f) * A + kk] = duv[q];
          p.dUv_lp[((size_t)b * F + f) * p.ld_dUv + kk] = (bf16_t)duv[q];
        }
      }
      if (XF) {
        for (int q = 16; gi + q * G < F; ++q) {
          const int f = gi + q * G;
          if (f >= 32) {
            const float v = sdux[(f - 32) * 128 + kk];
            p.dUv[((size_t)b * F + f) * A + kk] = v;
            p.dUv_lp[((size_t)b * F + f) * p.ld_dUv + kk] = (bf16_t)v;
          }
        }
      }
      spart[gi * A + kk] = dwa;
    }
    for (int f = wave; f < F; f += 4)
      for (int j = A + lane; j < p.ld_dUv; j += 64) p.dUv_lp[((size_t)b * F + f) * p.ld_dUv + j] = (bf16_t)0.f;
    __syncthreads();
    if (tid < A) {
      float a = 0.f;
      for (int j = 0; j < G; ++j) a += spart[j * A + tid];
      p.dwacc[(size_t)b * A + tid] = a;
#pragma unroll
      for (int ch = 1; ch < RN_FCH; ++ch) p.dwacc[((size_t)ch * Bs + b) * A + tid] = 0.f;
    }
  }
  rc_epoch_bump(p.epoch, ep0);
  rc_poison(p.bar, p.poison);
}
