// librecnet_hip.so — C ABI + host-side launch sequencing of the RecNet train step (include/recnet_hip.h).
//
// Sequence-level design: loop-invariant products are hoisted into batched MFMA GEMMs
// (Uv = enc.U^T, Xe = emb.W_e^T, logits, reconstructor input projection, every dW), only the
// h-recurrences run step by step, each step = one split-K MFMA GEMM over the packed recurrent weights
// + one fused per-caption kernel (LSTM gates + attention).  No allocation, no host synchronisation:
// every hot-path entry point only enqueues on the caller's stream.
//
// One translation unit, laid out over several files:
//   api.hip                  handle, workspace carving, create / bind / pack tables (this file)
//   host_common.inc          GEMM wrappers, split-K slabs of the chain sites, column sums, weight packing, invariants
//   host_decoder.inc         decoder forward chain + loss, backward chain + deferred weight gradients
//   host_reconstructor.inc   global / local reconstructor forward + backward, optimiser step
//   abi_search.inc           per-step decoder API, greedy / beam search
//   abi_step.inc             sequence-level entry points and the fused train step (stream orchestration)
//   abi_misc.inc             profiling hooks, bare GEMM entry points
// Device code: kernels.hpp -> kernels_{util,decoder,reconstructor,search,optim}.hpp, gemm*.hpp (GEMMs),
// rec_chain.hpp / loc_chain.hpp / dec_chain.hpp (the recurrent chains as persistent kernels), common.hpp.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <vector>

#include "../../include/recnet_hip.h"
#include "kernels.hpp"
#include "rec_chain.hpp"
#include "dec_chain.hpp"
#include "loc_chain.hpp"
#include "loc_big.hpp"

static thread_local std::string g_err;
static int fail(int code, const std::string& m) { g_err = m; return code; }
#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return fail(RECNET_EHIP, std::string(#x) + ": " + hipGetErrorString(e_)); } while (0)
#define LAUNCH_OK() do { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) return fail(RECNET_EHIP, std::string(__func__) + ": " + hipGetErrorString(e_)); } while (0)

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }
static inline int ew_blocks(size_t n) { size_t b = (n + 255) / 256; return (int)(b > 4096 ? 4096 : (b < 1 ? 1 : b)); }

struct OptGroup {
  std::vector<TensorDesc> tab;      // host copy
  std::vector<int2> chunks;
  TensorDesc* d_tab = nullptr; int2* d_chunks = nullptr; PackDesc* d_pack = nullptr;
  std::vector<PackDesc> pack;
  float* d_partial = nullptr; float* d_pnorm = nullptr; float* d_gnorm = nullptr;
  int ntens = 0, nchunks = 0;
  bool bound = false;
};

static inline int pad8(int n) { return (n + 7) & ~7; }
// index of rnn.weight_hh_l0 in the reconstructor's tensor list (rec_list): behind the four attention tensors of the local form
#define RN_REC_T_WHH(h) ((h)->kind == RECNET_REC_LOCAL ? 5 : 1)
#define RN_GCNT_WORDS 16384     // output tiles of the split members of one grouped launch
#define RN_MAX_ROW_GROUPS 4      // persistent chains up to 4 x 112 = 448 captions per GPU

struct recnet_handle {
  recnet_config c;
  int B, F, D, E, H, A, V, R, RA, Tm, kind, prec, cml;
  int dgru = 0, rgru = 0; // recurrent cell of the decoder / reconstructor: 0 LSTM, 1 GRU (4-block gate layout, kernels.hpp)
  int lp;                // 1: operand copies / packed weights are bf16 (DMA-staged GEMM); 0: fp32 (exact path)
  // leading dimensions (elements) of the operand buffers: multiples of 8
  int ldD, ldE, ldH, ldV, ldA, ld4H, ldWS, ldR, ld4R, ldRA, ldHR, ldRA4, ld2H;
  // workspace
  char* ws = nullptr; size_t ws_bytes = 0; size_t need = 0;
  uint32_t* ctrl;        // [0] seed slot, [1] step slot (int32)
  uint32_t* gbar;        // [256] grid-barrier flags of rec_chain_kernel
  float* scal;           // [0] dec_ce [1] dec_reg [2] dec_loss [3] rec_mse [4] rec_reg [5] rec_loss [6] total [7] gnorm [8] clip
  // ---- decoder: fp32 state
  float *slab2 = nullptr;   // second slab buffer (local reconstructor backward: dWhr . W_r)
  float *slab3 = nullptr;   // third: the dx part of the per-step backward product when it runs as its own branch (bwd_rec_local)
  float *bsum4 = nullptr, *bsum4r = nullptr;   // [4H], [4R] column sums of the gate gradients (source of both bias gradients)
  int prezeroed = 0;        // the step's atomic-sum targets were zeroed by one hoisted kernel (fwd_bwd)
  float *bsum_d, *Uv, *Xe, *Hs, *Cs, *acts, *Wh, *att, *logits, *rowloss, *slab, *gws, *dHs, *dHsrec, *dc_carry, *dUv,
      *dwacc, *demb, *stepw, *msep;
  // ---- decoder: operand copies (AT = bf16 | float)
  void *enc_lp, *emb_lp, *Hs_lp, *P, *dlog_lp, *dGx, *ctx_lp, *dUv_lp, *dWhs;
  // ---- decoder: packed weights (AT)
  void *U_w, *Wc_w, *We_w, *Wcomb, *Wo_w;
  void* WcombT = nullptr;   // [H][ldKW]: transpose of Wcomb, the decoder BPTT's GEMM operand with K contiguous (bf16 path)
  int ldKW = 0, use_wcomb_t = 0;
  // ---- reconstructor
  float *bsum_r, *mp, *Xg, *Hr, *Cr, *acts_r, *hrmean, *outm, *encmean, *dhrmean, *dmpd, *dmp, *dcr_carry;
  float *Ud, *beta, *Whr, *outl, *dHr, *dUd, *dwacc_r;
  void* Hr_pan = nullptr;
  void* hm_pan = nullptr;   // exchange copy of mean_t hr_t (rec_chain_kernel's output-layer epilogue)
  void* hd_pan = nullptr; void* WoT = nullptr; int dhr_done = 0;   // ... of the scaled dout; W_o^T image; dhrmean computed by the epilogue
  // backward chain for R > 2048 (loc_big.hpp): K partials, per-step masked dx / dbeta for the post-chain sums, streamed fragments of W^T
  float *lb_part = nullptr, *lb_dxm = nullptr, *lb_dbeta = nullptr; void* WstT = nullptr; int persist_big_bwd = 0, lb_steps = 0, lb_sr = 0, lb_ncb = 0;
  void* Wst = nullptr; int lc_steps = 0, lc_sr = 0;   // hybrid forward chain (R > 2048): streamed fragments image, k-steps per wave / resident
  void *lc_panh = nullptr, *lc_panx = nullptr; _Float16* lc_pw = nullptr;   // loc_chain.hpp exchange buffers
  void *lc_pang = nullptr, *lc_panw = nullptr; float* lc_dx = nullptr; void* WihhT = nullptr;   // ... of the backward chain; [W_ih | W_hh]^T
  int lcb_msx = 1, lcb_rbu = 4, lc_bwd_done = 0;
  int lcb_xs = 0;           // X' of the backward chain with K split over workgroups: k-steps per wave (0: whole K per workgroup)
  int deferred_early = 0, deferred_early_flags = -1, deferred_done = 0, join_recorded = 0;   // rec_deferred_fork (host_reconstructor.inc)
  float mse_scale = 0.f; int mse_nb = 0;   // pending MSE partials: fwd_rec finalises the loss scalars in one launch
  float dout_scale = 0.f; int dout_ready = 0;   // dout_lp already holds dout_scale * d loss / d out (written by the MSE kernel)
  unsigned long long* lc_ts = nullptr;   // probe builds (make probe): wall-clock stamps of the local chain kernels
  int lc_ms = 1, lc_rb = 4, lc_ng = 0, lc_nc = 0;
  void* dG_pan = nullptr;   // exchange copies of the gate gradients, rec_chain_bwd_kernel
  void* WhhT = nullptr;     // [R][ld4R] transpose of Whh_w (K contiguous) for rec_chain_bwd_kernel
  int persist_rec_bwd = 0;
  int persist_loc = 0, persist_loc_bwd = 0;   // loc_chain.hpp: the local reconstructor's chains as persistent launches
  int persist_dec = 0;      // dec_chain.hpp: the decoder's teacher-forced forward chain as one launch
  float* dc_G1 = nullptr; void* dc_pan = nullptr;
  int persist_dec_bwd = 0;  // ... and its BPTT chain
  int side_pending = 0, side_T = 0, side_phase = 0, side_err = 0;   // side_after_decoder_fwd (abi_step.inc)
  const int64_t* side_targets = nullptr; const float* side_stepw = nullptr; const float* side_enc = nullptr;
  int late_join = 0;
  int bgrp_loc = 0;              // ... of the local reconstructor's chains (<= 64 rows where more do not fit: recnet_create)
  int bgrp = 0;                  // rows per launch of the persistent chain kernels: B for B <= RC_PAN_ROWS, else B split evenly into ceil(B / 112) row groups
  // deferred reconstructor update (recnet_set_deferred_reconstructor_update): ctrl[2] on the device says whether an update is
  // pending; maybe_pending is the host's conservative shadow (replayed graphs do not run host code)
  int in_fused = 0, side_fork_recorded = 0;
  // Deferred refresh of the reconstructor's DERIVED weight images (round 5; opt-in with the deferred-update modes, applies where the
  // split update does not — 28 x 3584, row groups): the fused step skips the transposes / fragment packs behind its reconstructor
  // Adam step (183 us at the end of the step at 28 x 3584) and runs them at the start of the NEXT fused step, on the third stream
  // beside the decoder's forward chain; the reconstructor's chains wait for them (ev[12]).  The refresh is idempotent, so a captured
  // step is correct behind any other; images_maybe_stale is the host's shadow for the non-fused entry points (flush_pending).
  int img_defer_now = 0, images_maybe_stale = 0, s3_late = 0;
  int dp_overlap = 0, side_open = 0;   // recnet_set_dp_overlap: part 1 of the data-parallel step leaves the side stream's weight-gradient products unjoined (recnet_join_side)
  int hoist_fork_recorded = 0;   // dec_fwd_chain recorded the fork events of hoist_side_work itself, in front of the chain launch
  int rec_loss_defer = 0, rec_loss_late = 0;   // fused step: the reconstructor's loss scalars are formed beside the BPTT (rec_loss_scalars) instead of between its two chains
  int rec_norm_late = 0;   // mode 2: the norm of the pending-updated W_hh is joined in front of the loss scalars (ev[21]), not in front of the chain
  int side_tail_open = 0;  // decoder-only: dec_bwd_out recorded the BPTT's join in front of the rest of the side branch (ev[18] covers the rest)
  int total_late = 0;      // the total-loss scalar is formed on the side stream behind the BPTT's fork (fwd_bwd_impl)
  int split_ok = 0;              // the pending half of a split reconstructor update fits beside the decoder forward chain (recnet_create)
  int rec_wait_pending = 0;      // fwd_rec_global waits for ev[12] (the pending W_hh update, mode 2) in front of its recurrent chain
  int defer_rec = 0, defer_now = 0, defer_err = 0, maybe_pending = 0, def_rows = 0, defer_flags = 3; hipStream_t s3 = nullptr; float* gws3 = nullptr;
  int mp_done = 0;          // h->mp holds the mean-pooled decoder states of the last decoder forward (dec_chain_kernel)
  int ncu = 0;
  int ctx_done = 0;         // the attended features of all steps were computed early (fwd_bwd_impl)
  int hoist_pending = 0, hoist_par = 0, encmean_hoisted = 0; const float* hoist_enc = nullptr;   // see hoist_side_work (abi_step.inc)
  int xcat_done = 0;        // dec_chain_kernel wrote the global reconstructor's input operand itself (host_decoder.inc)
  float* dc_G2 = nullptr; void* dc_pan2 = nullptr;
  void *Xcat_g, *Hr_lp, *hrmean_lp, *dout_lp, *dGr, *Xcat_r, *dUd_lp, *dWhr, *dWhrs, *Wr4_w;
  void *Wih_f, *Whh_w, *Wor_w, *Ur_w, *Wr_w, *Wihh_w;
  int persist_rec = 0;     // rec_chain.hpp: the reconstructor's forward chain as one launch with W_hh resident in registers
  // inference search scratch (beam width <= 8)
  float *sr_logits, *sr_scores, *sr_h[2], *sr_c[2], *sr_hn, *sr_cn, *sr_cum[2], *sr_vals;
  int64_t *sr_tok[2], *sr_hist[2]; int32_t *sr_eos[2], *sr_idx;
  int64_t* in_tok = nullptr;   // [Tm][B] tokens fed by the last free-running forward (its backward scatters the embedding gradient by them)
  int free_fwd = 0;
  size_t gws_floats, slab_floats;
  float* gws2 = nullptr; float* gws_cur = nullptr;   // the side stream's split-K slabs / the one gemm() uses now
  unsigned* gcnt = nullptr;      // tile counters of the grouped launches' in-launch split-K sums: one block of RN_GCNT_WORDS per slab workspace
  int gg_site = 0;               // the next grouped launch stamps its start / end into this slot (1..8) of the group stamps
  // environment switches, read ONCE per handle in recnet_create (round 6: no getenv on any enqueue path; a test that flips one creates
  // a new handle).  Each selects between two tested forms of one piece of the schedule, never the arithmetic (tests/test_gpu_knobs.py,
  // tests/test_gpu_parity.py: _chain_variants).
  struct RnSw { int wait_chain = 1, mse_epi = 1, adam_epi = 1, dec_lw = 1, dec_rp = 1, dec_partial = 1, dec_xcat = 1, rec_epi = 2, rec_wide = 1, persist_ms = 0, gemm_group = 1; } sw;
  int gg_slots = 0;              // workgroup slots the next grouped launches can expect (0 = whole chip): see host_common.inc
  int gemm_single_group = 0;     // set around a single product whose K slices are to be summed inside its launch (host_decoder.inc: the embedding branch)
  hipStream_t s2 = nullptr; hipEvent_t ev[24] = {}; int overlap = 1;
  // bindings
  recnet_decoder_tensors dP{}, dGd{}, dM{}, dV{}, dVm{};
  recnet_reconstructor_tensors rP{}, rG{}, rM{}, rV{}, rVm{};
  bool dec_bound = false, rec_bound = false;
  OptGroup og[2];
  // state between forward and backward
  int T_last = 0, train_last = 0, fwd_dec_done = 0, fwd_rec_done = 0, rec_bwd_done = 0, early_opt_done = 0, norms_hoisted = 0, join_pending = 0, join_early = 0;
  // optional per-launch timing of the recurrent-step GEMM (recnet_profile_*)
  int prof_on = 0; std::vector<hipEvent_t> prof_ev; size_t prof_used = 0;
};

// ------------------------------------------------------------------------------------------------
static size_t carve(recnet_handle* h, char* base) {
  size_t off = 0;
  auto take = [&](size_t nfloats) -> float* {
    float* p = base ? (float*)(base + off) : nullptr;
    off += ((nfloats * 4 + 255) / 256) * 256;
    return p;
  };
  // operand buffers are sized as if fp32 (the bf16 path uses half of each)
  auto takev = [&](size_t n) -> void* { return (void*)take(n); };
  const size_t B = h->B, F = h->F, D = h->D, E = h->E, H = h->H, A = h->A, V = h->V, R = h->R, RA = h->RA, Tm = h->Tm;
  h->ldD = pad8(h->D); h->ldE = pad8(h->E); h->ldH = pad8(h->H); h->ldV = pad8(h->V); h->ldA = pad8(h->A);
  h->ld4H = pad8(4 * h->H); h->ldWS = pad8(4 * h->H + RN_FCH * h->A);
  h->ldR = pad8(h->R); h->ld4R = pad8(4 * h->R); h->ldRA = pad8(h->RA); h->ldHR = pad8(h->H + h->R);
  h->ldRA4 = pad8(RN_TCH * h->RA); h->ld2H = pad8(2 * h->H);
  const size_t ldD = h->ldD, ldE = h->ldE, ldH = h->ldH, ldV = h->ldV, ldA = h->ldA, ld4H = h->ld4H, ldWS = h->ldWS,
               ldR = h->ldR, ld4R = h->ld4R, ldRA = h->ldRA, ldHR = h->ldHR;
  h->ctrl = (uint32_t*)take(64);
  h->gbar = (uint32_t*)take(4096 + 64 + 2240);   // (floats: 16 KB) per chain kernel 256 arrival flags + 256 release words; epochs behind   // up to four persistent launches x 256 flags, then the launch-epoch words
  h->dc_G1 = take(2 * Tm * B * (4 * H + A));   // fp32, or 8-byte stamped words
  h->dc_pan = takev(Tm * rc_pan_elems((int)H) / 2 + 64);
  h->dc_G2 = take(2 * Tm * B * H * DCB_KS); h->dc_pan2 = takev(Tm * rc_pan_elems((int)(4 * H + A)) / 2 + 64);
  h->scal = take(64);
  h->lc_ts = (unsigned long long*)take(2 * (2 * 8 * 64 * 8));   // 8192 u64 entries (take counts floats): local chains [8 roles][64][8], decoder chains at +4096 / +4608
  h->stepw = take(Tm);
  h->msep = take(1024);
  h->bsum_d = take(4 * H);
  h->bsum4 = take(4 * H); h->bsum4r = take(4 * (R > 0 ? R : 1));
  h->Uv = take(B * F * A);
  h->Xe = take(Tm * B * 4 * H);
  h->Hs = take(Tm * B * H);
  h->Cs = take(Tm * B * H);
  h->acts = take(Tm * B * 4 * H);
  h->Wh = take(Tm * B * A);
  h->att = take(Tm * B * F);
  h->logits = take(Tm * B * V);
  h->rowloss = take(Tm * B);
  h->dHs = take(Tm * B * H);
  h->dHsrec = take(Tm * B * H);
  h->dc_carry = take(2 * B * H);
  h->dUv = take(B * F * A);
  h->dwacc = take(RN_FCH * B * A);
  h->demb = take(Tm * B * E);
  h->enc_lp = takev(B * F * ldD);
  h->emb_lp = takev(Tm * B * ldE);
  h->Hs_lp = takev(Tm * B * ldH);
  h->P = takev(B * F * ld4H);
  h->dlog_lp = takev(Tm * B * ldV);
  h->dGx = takev(Tm * B * ldWS);
  h->ctx_lp = takev(Tm * B * ldD);
  h->dUv_lp = takev(B * F * ldA);
  h->dWhs = takev(Tm * B * ldA);
  h->U_w = takev(A * ldD);
  h->Wc_w = takev(4 * H * ldD);
  h->We_w = takev(4 * H * ldE);
  h->Wcomb = takev((4 * H + RN_FCH * A) * ldH);
  h->ldKW = pad8(4 * h->H + RN_FCH * h->A);
  h->WcombT = takev(H * (size_t)h->ldKW);
  h->Wo_w = takev(V * ldH);
  size_t maxN = 4 * H + A;
  if (h->kind != RECNET_REC_NONE) {
    if (4 * R > maxN) maxN = 4 * R;
    if (H + R > maxN) maxN = H + R;
  }
  h->slab_floats = 32 * B * maxN;
  h->slab = take(h->slab_floats);
  h->gws_floats = (size_t)16 << 20;   // 64 MiB of split-K slabs for the batched GEMMs
  h->gws = take(h->gws_floats);
  h->gws2 = take(h->gws_floats);
  h->gws3 = take(h->gws_floats);
  h->gcnt = (unsigned*)take(3 * RN_GCNT_WORDS);      // zero from the workspace memset; every launch leaves its counters zeroed
  {
    const size_t W = 8;
    h->sr_logits = take(B * V); h->sr_scores = take(W * B * V);
    for (int i = 0; i < 2; ++i) {
      h->sr_h[i] = take(W * B * H); h->sr_c[i] = take(W * B * H); h->sr_cum[i] = take(W * B);
      h->sr_tok[i] = (int64_t*)take(2 * W * B); h->sr_hist[i] = (int64_t*)take(2 * W * B * Tm); h->sr_eos[i] = (int32_t*)take(W * B);
    }
    h->in_tok = (int64_t*)take(2 * Tm * B);
    h->sr_hn = take(W * B * H); h->sr_cn = take(W * B * H); h->sr_vals = take(W * B); h->sr_idx = (int32_t*)take(W * B);
  }
  if (h->kind != RECNET_REC_NONE) {
    h->bsum_r = take(4 * R);
    h->dcr_carry = take(B * R);
    h->Wor_w = takev(R * ldR);
  }
  if (h->kind == RECNET_REC_GLOBAL) {
    h->mp = take(B * H); h->Xg = take(Tm * B * 4 * R);
    h->Hr = take(Tm * B * R); h->Cr = take(Tm * B * R); h->acts_r = take(Tm * B * 4 * R);
    h->hrmean = take(B * R); h->outm = take(B * R); h->encmean = take(B * R); h->dhrmean = take(B * R);
    h->dmpd = take(Tm * B * 2 * H); h->dmp = take(B * H);   // dmpd: [T][B][2H] merged input-side product of the backward
    h->Xcat_g = takev(Tm * B * (size_t)h->ld2H); h->Hr_lp = takev(Tm * B * ldR); h->hrmean_lp = takev(B * ldR);
    h->Hr_pan = takev(Tm * rc_pan_elems((int)R) / 2 + 64);       // bf16: k-group-major copies of h_t for rec_chain_kernel
    h->hm_pan = takev(rc_pan_elems((int)R) / 2 + 64);
    h->hd_pan = takev(rc_pan_elems((int)R) / 2 + 64); h->WoT = takev(R * (size_t)h->ldR);
    h->dG_pan = takev(Tm * rc_pan_elems((int)(4 * R)) / 2 + 64);
    h->WhhT = takev(R * (size_t)h->ld4R);
    h->dout_lp = takev(B * ldR); h->dGr = takev(Tm * B * ld4R);
    h->Wih_f = takev(4 * R * (size_t)h->ld2H); h->Whh_w = takev(4 * R * ldR);
  } else if (h->kind == RECNET_REC_LOCAL) {
    h->Ud = take(Tm * B * RA);
    h->Hr = take(F * B * R); h->Cr = take(F * B * R); h->acts_r = take(F * B * 4 * R);
    h->beta = take(F * B * Tm); h->Whr = take(F * B * RA); h->outl = take(F * B * R); h->dHr = take(F * B * R);
    h->dUd = take(Tm * B * RA); h->dwacc_r = take(RN_TCH * B * RA);
    h->Xcat_r = takev(F * B * ldHR); h->Hr_lp = takev(F * B * ldR); h->dout_lp = takev(F * B * ldR);
    h->dGr = takev(F * B * ld4R); h->dUd_lp = takev(Tm * B * ldRA); h->dWhr = takev(F * B * (size_t)h->ldRA4);
    h->dWhrs = takev(F * B * ldRA); h->Wr4_w = takev(RN_TCH * RA * ldR);
    h->Ur_w = takev(RA * ldH); h->Wr_w = takev(RA * ldR); h->Wihh_w = takev(4 * R * ldHR);
    h->slab2 = take(16 * B * R); h->slab3 = take(16 * B * H);
    h->lc_panh = takev(F * rc_pan_elems((int)R) / 2 + 64);
    h->lc_panx = takev(F * rc_pan_elems((int)H) / 2 + 64);
    h->lc_pw = (_Float16*)take(F * B * ((R + 15) / 16) * RA / 2 + 64);
    {   // hybrid forward chain: 12 of the (R / 128 rounded up to even) k-steps per wave resident, the others streamed from this image
      int steps = (int)(R + 127) / 128; steps += steps & 1;
      h->lc_steps = steps > 16 ? (steps <= 28 ? 28 : 32) : 0; h->lc_sr = LC_HYB_SR(h->lc_steps);
      if (h->lc_steps) h->Wst = takev((R / 16) * 4 * (size_t)(h->lc_steps - h->lc_sr) * 4 * 512 / 2 + 64);
    }
    h->lc_pang = takev(F * rc_pan_elems((int)(4 * R)) / 2 + 64);
    h->lc_panw = takev(F * rc_pan_elems((int)RA) / 2 + 64);
    h->lc_dx = take(F * 4 * B * H);     // up to 4 K parts (lcb_xsplit_role)
    h->WihhT = takev((H + R) * (size_t)h->ld4R);
    if (R > 2048 && R % 128 == 0 && B <= 64 * 2 * RN_MAX_ROW_GROUPS) {     // the large-R backward chain (loc_big.hpp; eligibility: recnet_create; row groups of <= 64 captions)
      const int steps = (int)R / 128;
      h->lb_steps = steps; h->lb_sr = LB_SR(steps); h->lb_ncb = (int)(H + R) / 64;
      h->lb_part = take(F * 4 * 64 * (H + R));
      h->lb_dxm = take(F * B * H); h->lb_dbeta = take(F * B * Tm);
      h->WstT = takev((size_t)h->lb_ncb * 4 * 4 * (size_t)(steps - h->lb_sr) * 4 * 512 / 2 + 64);
    }
  }
  // optimiser tables (sizes are upper bounds; filled at bind time)
  for (int g = 0; g < 2; ++g) {
    OptGroup& o = h->og[g];
    size_t nparams = g == 0 ? (V * E + A * H + A * D + 2 * A + 4 * H * (E + D) + 4 * H * H + 8 * H + V * H + V)
                            : (RA * R + RA * H + 2 * RA + 4 * R * 2 * H + 4 * R * R + 8 * R + R * R + R);
    size_t maxch = nparams / RN_CHUNK + 16;
    o.d_tab = (TensorDesc*)take(16 * sizeof(TensorDesc) / 4);
    o.d_pack = (PackDesc*)take(16 * sizeof(PackDesc) / 4);
    o.d_chunks = (int2*)take(maxch * 2);
    o.d_partial = take(maxch);
    o.d_pnorm = take(16);
    o.d_gnorm = take(16);
  }
  return off;
}

static DropDesc mkdrop(const recnet_handle* h, uint32_t site, float p, int train) {
  DropDesc d;
  d.seed = h->ctrl; d.site = site;
  d.thr = train ? rn_drop_thr(p) : 0u;
  d.inv_keep = (p < 1.f) ? 1.0f / (1.0f - p) : 0.f;
  d.Bg = h->c.global_batch_size; d.boff = h->c.batch_offset;
  return d;
}

extern "C" {

int recnet_abi_version(void) { return RECNET_ABI_VERSION; }
const char* recnet_last_error(void) { return g_err.c_str(); }

int recnet_create(const recnet_config* cfg, recnet_handle** out) {
  if (!cfg || !out) return fail(RECNET_EINVAL, "null argument");
  const recnet_config& c = *cfg;
  if (c.batch_size <= 0 || c.encoder_output_len <= 0 || c.encoder_output_size <= 0 || c.embedding_size <= 0 ||
      c.decoder_hidden_size <= 0 || c.decoder_attn_size <= 0 || c.n_vocabs <= 3 || c.caption_max_len <= 0)
    return fail(RECNET_EINVAL, "non-positive dimension");
  if (c.reconstructor_type < 0 || c.reconstructor_type > 2) return fail(RECNET_EINVAL, "unknown reconstructor_type");
  if (c.precision != RECNET_PREC_F32 && c.precision != RECNET_PREC_BF16) return fail(RECNET_EINVAL, "unknown precision");
  if (c.decoder_attn_normalize != RECNET_ATTN_NONE && c.decoder_attn_normalize != RECNET_ATTN_SOFTMAX)
    return fail(RECNET_EINVAL, "unknown decoder_attn_normalize (0 = none, 1 = softmax)");
  if (c.decoder_cell < 0 || c.decoder_cell > 1 || c.reconstructor_cell < 0 || c.reconstructor_cell > 1)
    return fail(RECNET_EINVAL, "unknown recurrent cell (0 = LSTM, 1 = GRU)");
  if (c.reconstructor_type != RECNET_REC_NONE && c.reconstructor_hidden_size <= 0)
    return fail(RECNET_EINVAL, "reconstructor_hidden_size");
  if (c.reconstructor_type == RECNET_REC_LOCAL && c.reconstructor_hidden_size != c.encoder_output_size)
    return fail(RECNET_EINVAL, "local reconstructor requires reconstructor_hidden_size == encoder_output_size (train.py:128)");
  if (c.reconstructor_type == RECNET_REC_GLOBAL && c.reconstructor_hidden_size != c.encoder_output_size)
    return fail(RECNET_EINVAL, "global reconstructor requires reconstructor_hidden_size == encoder_output_size (train.py:101)");
  if (c.reconstructor_type == RECNET_REC_LOCAL && c.reconstructor_attn_size <= 0)
    return fail(RECNET_EINVAL, "reconstructor_attn_size");
  recnet_handle* h = new recnet_handle();
  h->c = c;
  if (h->c.global_batch_size <= 0) h->c.global_batch_size = c.batch_size;
  h->B = c.batch_size; h->F = c.encoder_output_len; h->D = c.encoder_output_size; h->E = c.embedding_size;
  h->H = c.decoder_hidden_size; h->A = c.decoder_attn_size; h->V = c.n_vocabs;
  h->R = c.reconstructor_type ? c.reconstructor_hidden_size : 0;
  h->RA = c.reconstructor_type == RECNET_REC_LOCAL ? c.reconstructor_attn_size : 0;
  h->cml = c.caption_max_len; h->Tm = c.caption_max_len + 1;
  h->kind = c.reconstructor_type; h->prec = c.precision; h->lp = c.precision == RECNET_PREC_BF16;
  h->gemm_single_group = 0;
  // Two list-valued switches (round 6: 23 variables -> 9), comma separated, read here once:
  //   RN_PER_STEP  chains that run on the per-step kernels instead of their persistent launch: dec, dec_bwd, rec, rec_bwd, loc, loc_bwd,
  //                loc_big — or all;
  //   RN_ALT       alternative forms the tests hold the default against: dec_wh_in_phase_a, dec_all_rows, dec_relayed_barrier,
  //                dec_no_xcat, rec_epilogue_0 / rec_epilogue_1, rec_bwd_narrow, rec_row_parts_1 / rec_row_parts_2, loc_no_hybrid,
  //                loc_xsplit_never / loc_xsplit_always.
  auto in_list = [](const char* var, const char* name, bool all_ok) {
    const char* e = getenv(var);
    if (!e) return false;
    const size_t n = strlen(name);
    for (const char* p = e; *p;) {
      const char* q = strchr(p, ',');
      const size_t len = q ? (size_t)(q - p) : strlen(p);
      if ((len == n && !strncmp(p, name, n)) || (all_ok && len == 3 && !strncmp(p, "all", 3))) return true;
      p = q ? q + 1 : p + len;
    }
    return false;
  };
  auto alt = [&](const char* name) { return in_list("RN_ALT", name, false); };
  auto chain_on = [&](const char* name) { return !in_list("RN_PER_STEP", name, true); };
  {
    auto env = [](const char* n, int dflt) { const char* e = getenv(n); return e ? atoi(e) : dflt; };
    h->sw.wait_chain = env("RN_WAIT_CHAIN", 1); h->sw.mse_epi = env("RN_MSE_EPILOGUE", 1); h->sw.adam_epi = env("RN_ADAM_EPILOGUE", 1);
    h->sw.gemm_group = env("RN_GEMM_GROUP", 1);
    // RN_ALT: the tested alternative forms of the chain kernels, by name (see in_list below)
    h->sw.dec_lw = !alt("dec_wh_in_phase_a"); h->sw.dec_rp = !alt("dec_all_rows"); h->sw.dec_partial = !alt("dec_relayed_barrier");
    h->sw.dec_xcat = !alt("dec_no_xcat"); h->sw.rec_epi = alt("rec_epilogue_0") ? 0 : (alt("rec_epilogue_1") ? 1 : 2); h->sw.rec_wide = !alt("rec_bwd_narrow");
    h->sw.persist_ms = alt("rec_row_parts_1") ? 1 : (alt("rec_row_parts_2") ? 2 : 0);
  }
  h->dgru = c.decoder_cell == RECNET_CELL_GRU; h->rgru = c.reconstructor_type != RECNET_REC_NONE && c.reconstructor_cell == RECNET_CELL_GRU;
  // The chain kernels exchange h_t / dgates_t through 112-row panels (RC_PAN_ROWS).  A larger batch is cut into row groups of
  // equal size (at most RN_MAX_ROW_GROUPS of them) and every chain runs once per group, one launch after the other: a group is
  // an independent batch for a chain.  Everything batched (GEMMs, CE, optimiser) still sees the whole batch.
  {
    const char* eg = getenv("RN_ROW_GROUPS");      // 0: no grouping (batches above 112 captions take the per-step kernels)
    int ng = (h->B + RC_PAN_ROWS - 1) / RC_PAN_ROWS;
    if (eg && atoi(eg) == 0 && ng > 1) ng = 0;
    h->bgrp = ng >= 1 && ng <= RN_MAX_ROW_GROUPS ? (h->B + ng - 1) / ng : h->B;
  }
  const int Bg = h->bgrp;     // rows of one chain launch
  {
    // rec_chain.hpp: every workgroup (8 hidden units) must be resident at once — one per CU
    int dev = 0, ncu = 0;
    hipGetDevice(&dev);
    hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev);
    h->ncu = ncu;
    // Tm <= 60: stamped words carry epoch << 6 | step and barrier words epoch << 7 | phase (rec_chain.hpp) — a longer
    // caption limit would let one launch's values run into the next epoch's, so it takes the per-step kernels
    h->persist_rec = chain_on("rec") && h->lp && h->kind == RECNET_REC_GLOBAL && Bg <= RC_PAN_ROWS && h->Tm <= 60 &&
                     (h->R & 7) == 0 && h->R <= 2048 && h->R / 8 <= ncu;   // <= 16 k-steps of resident weights per wave
    const int N = 4 * h->H + h->A, NA = N / 16;
    h->persist_dec = chain_on("dec") && h->lp && h->Tm <= 60 && (h->H & 7) == 0 && h->H <= 512 && h->F <= 32 + DC_XF && h->A <= 128 &&
                     (N & 15) == 0 && Bg <= RC_PAN_ROWS && (NA > Bg ? NA : Bg) + 1 <= ncu;
    h->persist_rec_bwd = chain_on("rec_bwd") && h->persist_rec && (h->R & 15) == 0;
  }
  h->bgrp_loc = h->bgrp;
  {
    // loc_chain.hpp: the local reconstructor's forward chain as one launch (unit-owner + caption workgroups + relay)
    // Row groups of the LOCAL chains (round 4): above 64 rows the forward chain splits its unit owners into two row parts
    // (2 R / 16 workgroups) and the backward chain's U' role takes all rows — at R = 2048 that is 279 workgroups, more than the
    // chip has CUs, and a batch (or row group) of more than 64 captions ran the per-step kernels there.  It now runs the local chains
    // in groups of at most 64 rows (more, smaller groups than the decoder's chains, which take up to 112): every [s][B][.] tensor is
    // indexed by the caption's row in the whole batch, so the two groupings do not have to agree.
    int Bg = h->bgrp;      // (shadows the decoder's group size inside this block)
    // (round 6: the same above R = 2048 — the hybrid forward chain has R / 16 >= 129 unit owners and no room for a second row part, and
    // the phased backward chain of loc_big.hpp takes at most 64 rows: batches up to 256 captions run both in groups)
    if (h->kind == RECNET_REC_LOCAL && Bg > 64 && (h->R > 2048 || (h->R / 16) * 2 + (Bg + LC_CPW - 1) / LC_CPW + 1 > h->ncu)) {
      const int ngl = (h->B + 63) / 64;
      if (ngl <= 2 * RN_MAX_ROW_GROUPS) { Bg = (h->B + ngl - 1) / ngl; h->bgrp_loc = Bg; }
    }
    h->lc_ms = Bg > 64 ? 2 : 1; h->lc_rb = (Bg <= 32) ? 2 : 4;
    h->lc_ng = h->R / 16; h->lc_nc = (Bg + LC_CPW - 1) / LC_CPW;
    // (the relay workgroup is dropped when there is no CU left for it: nwg - 1 == CU count, R = 3584 with 64 captions)
    int nwg = h->lc_ng * h->lc_ms + h->lc_nc + 1;
    if (nwg - 1 == h->ncu) nwg -= 1;
    // R above 2048 (a multiple of 256): the hybrid form of loc_chain_kernel (12 k-steps per wave resident in registers, the
    // rest streamed every step), forward chain only; the backward runs the per-step kernels
    const int f_hyb = !alt("loc_no_hybrid");
    h->persist_loc = chain_on("loc") && h->lp && h->kind == RECNET_REC_LOCAL && Bg <= RC_PAN_ROWS && (h->R & 31) == 0 &&
                     (h->R <= 2048 || (f_hyb && h->R <= 4096 && (h->R & 255) == 0)) && h->lc_ng <= 256 && (h->H & 31) == 0 && h->H <= 512 && h->RA <= 128 && (h->RA & 3) == 0 && h->Tm <= 32 &&
                     h->F + 1 < LC_MAX_PHASE &&   // barrier words are epoch << 7 | phase, phase <= F + 1 (loc_chain.hpp)
#ifdef LC_PROBE
                     h->F <= 64 &&                // LC_TS indexes [role][step < 64][8]
#endif
                     nwg <= h->ncu && nwg - 1 <= 256;
    // ... and its backward chain: U' all rows (RB 7) + X' two row parts above 64 captions, one part of 64 rows below
    h->lcb_msx = Bg > 64 ? 2 : 1; h->lcb_rbu = Bg > 64 ? 7 : (Bg > 32 ? 4 : 2);
    int nwb = h->lc_ng + (h->H / 16) * h->lcb_msx + h->lc_nc + 1;
    {   // X' with K split in parts of 2048 (64 columns x 32 rows x one part per workgroup) when those workgroups fit as well
      const int ksx = (4 * h->R + 2047) / 2048, nwx = h->lc_ng + ((h->H + 63) / 64) * ((Bg + 31) / 32) * ksx + h->lc_nc + 1;
      const int fx = alt("loc_xsplit_never") ? 0 : (alt("loc_xsplit_always") ? 2 : 1);
      if (fx && ksx <= 4 && (h->R >= 512 || fx == 2) && nwx <= h->ncu && nwx - 1 <= 256) { h->lcb_xs = 16; nwb = nwx; }
    }
    // (R <= 2048: its kernels hold at most 64 k-steps of K = 4R per wave; the hybrid forward chain above that runs with the
    // per-step backward)
    {   // R above 2048: the phased backward chain of loc_big.hpp (P / C / L phases of (H + R) / 64 * 4 workgroups)
      const int steps = h->R / 128, nwg = (h->H + h->R) / 64 * 4;
      h->persist_big_bwd = chain_on("loc_big") && chain_on("loc_bwd") && h->lp && h->kind == RECNET_REC_LOCAL && h->R > 2048 && h->R % 128 == 0 &&
                           (steps >= 18 && steps <= 32 && (steps & 1) == 0) && h->H % 64 == 0 && h->H <= 512 && h->RA <= 128 && (h->RA & 7) == 0 &&
                           Bg <= 64 && h->B <= 64 * 2 * RN_MAX_ROW_GROUPS && h->Tm <= 32 && h->F <= 40 && nwg <= h->ncu && nwg <= 256 && h->R / 16 < nwg && Bg < nwg;
    }
    h->persist_loc_bwd = chain_on("loc_bwd") && h->persist_loc && h->R <= 2048 && (h->H & 15) == 0 && !(Bg > 64 && h->R > 1536) &&
                         nwb <= h->ncu && nwb - 1 <= 256;
  }
  {
    h->use_wcomb_t = h->lp && (h->B <= 128 || Bg < h->B);
    int dev = 0, ncu = 0;
    hipGetDevice(&dev);
    hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev);
    const int NAb = DCB_NA(h->H);
    // RN_RESERVE_CUS: CUs left to a collective kernel (RCCL) that is resident while this chain runs — data parallel runs
    // all-reduce the reconstructor bucket under the decoder's BPTT (dp.py sets 64).  The chain would still complete
    // without the reserve (the collective does not wait for it), but its first steps would spin until CUs free up.
    const char* er = getenv("RN_RESERVE_CUS");
    const int reserve = er ? atoi(er) : 0;
    h->persist_dec_bwd = chain_on("dec_bwd") && h->persist_dec && h->use_wcomb_t && (h->H & 15) == 0 && (h->ldWS & 7) == 0 &&
                         (NAb > Bg ? NAb : Bg) + 1 + reserve <= ncu;
  }
  {
    // Split reconstructor update (mode 2 of recnet_set_deferred_reconstructor_update): d W_hh of step n runs beside the decoder
    // forward chain of step n + 1, on the CUs that chain leaves idle.  It is applied only where it fits there — estimated from
    // the product's FLOPs at the rate the grouped GEMM reaches beside a chain (~2.4 TFLOP/s per CU, measured) against ~10 us per
    // chain step; at R = 3584 the product (177 GFLOP) would take twice the chain's time and hold up the reconstructor's forward.
    const int NA = (4 * h->H + h->A) / 16, wg = (NA > Bg ? NA : Bg) + 1, free_cus = h->ncu - wg;
    const double rows = (double)((h->kind == RECNET_REC_LOCAL ? h->F : h->Tm) - 1) * h->B;
    const double flops = 2.0 * 4.0 * h->R * (double)h->R * rows;
    // (x 1.36: the Adam epilogue of the product — calibrated on C2: 58.5 GFLOP + 9.4 M parameters in 279 us on 119 CUs.
    // Leaving only SOME gate blocks of W_hh pending at R = 3584 was tried in round 4: the fragment images of the R > 2048 chains
    // have to be re-packed behind either half, and the step got slower, 3.49 against 3.29 ms.)
    const double t_pending = free_cus > 0 ? 1.36 * flops / (free_cus * 2.4e6) : 1e30, t_chain = 0.95 * 10.0 * h->Tm * ((h->B + Bg - 1) / Bg);
    // (one row group only: at B = 200, two groups of 100, the split form measured 3.21 ms against 3.10 ms for the immediate one)
    h->split_ok = h->kind != RECNET_REC_NONE && h->persist_dec && h->lp && Bg == h->B && t_pending <= t_chain;
  }
  h->need = carve(h, nullptr);
  *out = h;
  return RECNET_OK;
}

void recnet_destroy(recnet_handle* h) {
  if (!h) return;
  for (auto e : h->prof_ev) hipEventDestroy(e);
  for (int i = 0; i < 24; ++i) if (h->ev[i]) hipEventDestroy(h->ev[i]);
  if (h->s2) hipStreamDestroy(h->s2);
  if (h->s3) hipStreamDestroy(h->s3);
  delete h;
}

int recnet_set_shard(recnet_handle* h, int32_t global_batch_size, int32_t batch_offset) {
  if (!h || global_batch_size < h->B || batch_offset < 0) return fail(RECNET_EINVAL, "bad shard");
  h->c.global_batch_size = global_batch_size; h->c.batch_offset = batch_offset;
  return RECNET_OK;
}

size_t recnet_workspace_bytes(const recnet_handle* h) { return h ? h->need : 0; }

static inline void* at_off(const recnet_handle* h, void* p, size_t elems);
// destinations of the packed operand images, per parameter tensor (same order as dec_list / rec_list)
static void build_pack_tables(recnet_handle* h, int g) {
  OptGroup& o = h->og[g];
  o.pack.assign(o.ntens, PackDesc());
  for (auto& pd : o.pack) { pd.ndst = 0; pd.cols = 1; }
  auto addr = [&](int t, int cols, void* dst, int ld, int c0, int nc, int r0, int nr) {
    PackDesc& pd = o.pack[t]; pd.cols = cols;
    PackDst& d = pd.d[pd.ndst++]; d.dst = dst; d.ld = ld; d.c0 = c0; d.nc = nc; d.r0 = r0; d.nr = nr; d.mode = 0; d.pad = 0;
  };
  auto add = [&](int t, int cols, void* dst, int ld, int c0, int nc) { addr(t, cols, dst, ld, c0, nc, 0, 1 << 30); };
  // recurrent weights into the 4-block gate layout: W_ih rows as they are (GRU: 3 blocks, the 4th stays zero);
  // W_hh of a GRU: blocks (r, z) in place, block n -> packed block 3, packed block 2 stays zero
  auto add_ih = [&](int t, int cols, void* dst, int ld, int c0, int nc) { add(t, cols, dst, ld, c0, nc); };
  auto add_hh = [&](int gru, int Hd, int t, int cols, void* dst, int ld, int c0, int nc) {
    if (!gru) { add(t, cols, dst, ld, c0, nc); return; }
    addr(t, cols, dst, ld, c0, nc, 0, 2 * Hd);
    addr(t, cols, at_off(h, dst, (size_t)3 * Hd * ld), ld, c0, nc, 2 * Hd, Hd);
  };
  const int H = h->H, D = h->D, E = h->E, A = h->A, R = h->R, RA = h->RA;
  if (g == 0) {
    for (int j = 0; j < RN_FCH; ++j) add(2, H, at_off(h, h->Wcomb, (size_t)(4 * H + j * A) * h->ldH), h->ldH, 0, H);   // attn_W
    add(3, D, h->U_w, h->ldD, 0, D);                                        // attn_U
    add_ih(5, E + D, h->We_w, h->ldE, 0, E); add_ih(5, E + D, h->Wc_w, h->ldD, E, D);   // rnn.weight_ih_l0
    add_hh(h->dgru, H, 6, H, h->Wcomb, h->ldH, 0, H);                       // rnn.weight_hh_l0
    add(9, H, h->Wo_w, h->ldH, 0, H);                                       // out.weight
  } else if (h->kind == RECNET_REC_GLOBAL) {
    add_ih(0, 2 * H, h->Wih_f, h->ld2H, 0, 2 * H);
    add_hh(h->rgru, R, 1, R, h->Whh_w, h->ldR, 0, R);
    add(4, R, h->Wor_w, h->ldR, 0, R);
  } else if (h->kind == RECNET_REC_LOCAL) {
    add(1, R, h->Wr_w, h->ldR, 0, R);
    for (int j = 0; j < RN_TCH; ++j) add(1, R, at_off(h, h->Wr4_w, (size_t)j * RA * h->ldR), h->ldR, 0, R);
    add(2, H, h->Ur_w, h->ldH, 0, H);
    add_ih(4, H, h->Wihh_w, h->ldHR, 0, H);
    add_hh(h->rgru, R, 5, R, at_off(h, h->Wihh_w, (size_t)H), h->ldHR, 0, R);
    add(8, R, h->Wor_w, h->ldR, 0, R);
  }
}

static int upload_tables(recnet_handle* h, int g) {
  OptGroup& o = h->og[g];
  if (!o.bound || !h->ws) return RECNET_OK;
  build_pack_tables(h, g);
  HIPCHK(hipMemcpy(o.d_pack, o.pack.data(), o.pack.size() * sizeof(PackDesc), hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(o.d_tab, o.tab.data(), o.tab.size() * sizeof(TensorDesc), hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(o.d_chunks, o.chunks.data(), o.chunks.size() * sizeof(int2), hipMemcpyHostToDevice));
  return RECNET_OK;
}

int recnet_bind_workspace(recnet_handle* h, void* workspace, size_t bytes) {
  if (!h || !workspace) return fail(RECNET_EINVAL, "null argument");
  if (bytes < h->need) return fail(RECNET_EINVAL, "workspace too small");
  if (((uintptr_t)workspace) % 256) return fail(RECNET_EINVAL, "workspace must be 256-byte aligned");
  h->ws = (char*)workspace; h->ws_bytes = bytes;
  carve(h, h->ws);
  // everything starts from zero: operand rows a step does not write (time steps beyond its T) are read — against zero
  // gradients — by the fixed-shape products of a deferred reconstructor update, and 0 x NaN bit patterns would not be 0
  HIPCHK(hipMemset(h->ws, 0, h->need));
  // stamped exchange buffers and the launch-epoch words start from zero (a stamp is never zero)
  HIPCHK(hipMemset(h->gbar, 0, (4096 + 64 + 2240) * 4)); HIPCHK(hipMemset(h->scal, 0, 64 * 4)); HIPCHK(hipMemset(h->dc_G1, 0, (size_t)2 * h->Tm * h->B * (4 * h->H + h->A) * 4));
  HIPCHK(hipMemset(h->dc_G2, 0, (size_t)2 * h->Tm * h->B * h->H * DCB_KS * 4));
  h->gws_cur = h->gws;
  if (!h->s2) {
    h->overlap = 1;
    // (a lowest-priority side stream was measured in round 4: 2.63 against 1.82 ms — the side work is on the critical path often enough)
    // (non-default stream priorities for the side streams — lowest or highest — cost 0.12 ms at C2 and 0.07 at C3, round 6: default priority)
    HIPCHK(hipStreamCreateWithFlags(&h->s2, hipStreamNonBlocking));
    HIPCHK(hipStreamCreateWithFlags(&h->s3, hipStreamNonBlocking));
    for (int i = 0; i < 24; ++i) HIPCHK(hipEventCreateWithFlags(&h->ev[i], hipEventDisableTiming));
  }
  h->fwd_dec_done = h->fwd_rec_done = h->rec_bwd_done = 0;
  int r = upload_tables(h, 0); if (r) return r;
  return upload_tables(h, 1);
}

static void build_group(OptGroup& o, const std::vector<float*>& p, const std::vector<float*>& g,
                        const std::vector<float*>& m, const std::vector<float*>& v, const std::vector<float*>& vm,
                        const std::vector<size_t>& n) {
  o.tab.clear(); o.chunks.clear();
  for (size_t i = 0; i < p.size(); ++i) {
    TensorDesc td;
    td.p = p[i]; td.g = g[i]; td.m = m.empty() ? nullptr : m[i]; td.v = v.empty() ? nullptr : v[i];
    td.vmax = vm.empty() ? nullptr : vm[i];
    td.n = (int)n[i]; td.chunk0 = (int)o.chunks.size(); td.pad = 0;
    int nc = 0;
    for (size_t off = 0; off < n[i]; off += RN_CHUNK) { o.chunks.push_back(make_int2((int)i, (int)off)); ++nc; }
    td.nchunks = nc;
    o.tab.push_back(td);
  }
  o.ntens = (int)p.size(); o.nchunks = (int)o.chunks.size(); o.bound = true;
}

static std::vector<float*> dec_list(const recnet_decoder_tensors* t) {
  if (!t) return {};
  return {t->attn_b, t->embedding_weight, t->attn_W_weight, t->attn_U_weight, t->attn_w_weight, t->rnn_weight_ih_l0,
          t->rnn_weight_hh_l0, t->rnn_bias_ih_l0, t->rnn_bias_hh_l0, t->out_weight, t->out_bias};
}
static std::vector<float*> rec_list(const recnet_reconstructor_tensors* t, bool local) {
  if (!t) return {};
  std::vector<float*> v;
  if (local) { v.push_back(t->attn_b); v.push_back(t->attn_W_weight); v.push_back(t->attn_U_weight); v.push_back(t->attn_w_weight); }
  v.push_back(t->rnn_weight_ih_l0); v.push_back(t->rnn_weight_hh_l0); v.push_back(t->rnn_bias_ih_l0);
  v.push_back(t->rnn_bias_hh_l0); v.push_back(t->out_weight); v.push_back(t->out_bias);
  return v;
}
static bool any_null(const std::vector<float*>& v) { for (auto p : v) if (!p) return true; return false; }

int recnet_bind_decoder(recnet_handle* h, const recnet_decoder_tensors* param, const recnet_decoder_tensors* grad,
                        const recnet_decoder_tensors* exp_avg, const recnet_decoder_tensors* exp_avg_sq,
                        const recnet_decoder_tensors* max_exp_avg_sq) {
  if (!h || !param) return fail(RECNET_EINVAL, "null argument");
  auto P = dec_list(param);
  if (any_null(P)) return fail(RECNET_EINVAL, "decoder parameter pointer is null");
  h->dP = *param;
  if (grad) h->dGd = *grad; else memset(&h->dGd, 0, sizeof(h->dGd));
  const size_t V = h->V, E = h->E, H = h->H, A = h->A, D = h->D;
  const size_t NG = h->dgru ? 3 : 4;
  std::vector<size_t> n = {A, V * E, A * H, A * D, A, NG * H * (E + D), NG * H * H, NG * H, NG * H, V * H, V};
  auto G = dec_list(grad);
  if (G.empty()) G.assign(P.size(), nullptr);
  auto M = dec_list(exp_avg), Vv = dec_list(exp_avg_sq), Vm = dec_list(max_exp_avg_sq);
  if (!M.empty() && (any_null(M) || Vv.empty() || any_null(Vv))) return fail(RECNET_EINVAL, "Adam state pointer is null");
  if (h->c.decoder_use_amsgrad && !M.empty() && (Vm.empty() || any_null(Vm)))
    return fail(RECNET_EINVAL, "amsgrad needs max_exp_avg_sq");
  build_group(h->og[0], P, G, M, Vv, Vm, n);
  h->dec_bound = true;
  return upload_tables(h, 0);
}

int recnet_bind_reconstructor(recnet_handle* h, const recnet_reconstructor_tensors* param,
                              const recnet_reconstructor_tensors* grad, const recnet_reconstructor_tensors* exp_avg,
                              const recnet_reconstructor_tensors* exp_avg_sq,
                              const recnet_reconstructor_tensors* max_exp_avg_sq) {
  if (!h || !param) return fail(RECNET_EINVAL, "null argument");
  if (h->kind == RECNET_REC_NONE) return fail(RECNET_EINVAL, "handle was created without a reconstructor");
  const bool local = h->kind == RECNET_REC_LOCAL;
  auto P = rec_list(param, local);
  if (any_null(P)) return fail(RECNET_EINVAL, "reconstructor parameter pointer is null");
  h->rP = *param;
  if (grad) h->rG = *grad; else memset(&h->rG, 0, sizeof(h->rG));
  const size_t H = h->H, R = h->R, RA = h->RA;
  std::vector<size_t> n;
  const size_t NG = h->rgru ? 3 : 4;
  if (local) { n = {RA, RA * R, RA * H, RA, NG * R * H, NG * R * R, NG * R, NG * R, R * R, R}; }
  else { n = {NG * R * 2 * H, NG * R * R, NG * R, NG * R, R * R, R}; }
  auto G = rec_list(grad, local);
  if (G.empty()) G.assign(P.size(), nullptr);
  auto M = rec_list(exp_avg, local), Vv = rec_list(exp_avg_sq, local), Vm = rec_list(max_exp_avg_sq, local);
  if (!M.empty() && (any_null(M) || Vv.empty() || any_null(Vv))) return fail(RECNET_EINVAL, "Adam state pointer is null");
  if (h->c.reconstructor_use_amsgrad && !M.empty() && (Vm.empty() || any_null(Vm)))
    return fail(RECNET_EINVAL, "amsgrad needs max_exp_avg_sq");
  build_group(h->og[1], P, G, M, Vv, Vm, n);
  h->rec_bound = true;
  return upload_tables(h, 1);
}

}  // extern "C"

#include "host_common.inc"
#include "host_decoder.inc"
#include "host_reconstructor.inc"

// ================================================================================================
extern "C" {

// A pending deferred reconstructor update (recnet_set_deferred_reconstructor_update) is completed before anything else
// reads the reconstructor's parameters, packed images, gradients or Adam state.
static void refresh_rec_images(recnet_handle* h, hipStream_t st);
static int flush_pending(recnet_handle* h, hipStream_t st, int explicit_call = 0) {
  if (h->images_maybe_stale && h->rec_bound) { refresh_rec_images(h, st); h->images_maybe_stale = 0; }      // (deferred image refresh)
  // maybe_pending is the host's shadow of the device's pending word: set when a deferred step is enqueued or captured, and
  // by recnet_mark_pending when a captured one is replayed.  An explicit recnet_flush also runs while the mode is on (the
  // device word decides whether the Adam step happens; the products are recomputed from the step's own operands either way).
  if (!h->maybe_pending && !(explicit_call && h->defer_rec)) return RECNET_OK;
  h->maybe_pending = 0;
  if (h->kind == RECNET_REC_NONE || !h->og[1].bound || !h->og[1].tab[0].m) return RECNET_OK;
  return rec_pending_update(h, st, 0, h->defer_flags);
}
#define FLUSH_PENDING(h, st) do { int fr_ = flush_pending(h, (hipStream_t)(st)); if (fr_) return fr_; } while (0)

int recnet_pack_weights(recnet_handle* h, void* stream) {
  REQUIRE_WS(h);
  FLUSH_PENDING(h, stream);
  int r = pack_weights(h, (hipStream_t)stream); if (r) return r;
  LAUNCH_OK();
  return RECNET_OK;
}

#include "abi_search.inc"
#include "abi_step.inc"
#include "abi_misc.inc"
