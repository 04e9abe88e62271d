// librecnet_hip.so — C ABI + host-side launch sequencing of the RecNet train step (include/recnet_hip.h).
//
// Sequence-level design: loop-invariant products are hoisted into batched MFMA GEMMs
// (Uv = enc.U^T, Xe = emb.W_e^T, logits, reconstructor input projection, every dW), only the
// h-recurrences run step by step, each step = one split-K MFMA GEMM over the packed recurrent weights
// + one fused per-caption kernel (LSTM gates + attention).  No allocation, no host synchronisation:
// every hot-path entry point only enqueues on the caller's stream.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <vector>

#include "../../include/recnet_hip.h"
#include "kernels.hpp"
#include "rec_step.hpp"

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
  float* scal;           // [0] dec_ce [1] dec_reg [2] dec_loss [3] rec_mse [4] rec_reg [5] rec_loss [6] total [7] gnorm [8] clip
  // ---- decoder: fp32 state
  float *slab2 = nullptr;   // second slab buffer (local reconstructor backward: dWhr . W_r)
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
  void *Xcat_g, *Hr_lp, *hrmean_lp, *dout_lp, *dGr, *Xcat_r, *dUd_lp, *dWhr, *dWhrs, *Wr4_w;
  void *Wih_f, *Whh_w, *Wor_w, *Ur_w, *Wr_w, *Wihh_w;
  void* Whh_g = nullptr;   // gate-interleaved W_hh of the fused recurrent step (rec_step.hpp); global reconstructor, LSTM, bf16
  int fused_rec = 0;
  // inference search scratch (beam width <= 8)
  float *sr_logits, *sr_scores, *sr_h[2], *sr_c[2], *sr_hn, *sr_cn, *sr_cum[2], *sr_vals;
  int64_t *sr_tok[2], *sr_hist[2]; int32_t *sr_eos[2], *sr_idx;
  int64_t* in_tok = nullptr;   // [Tm][B] tokens fed by the last free-running forward (its backward scatters the embedding gradient by them)
  int free_fwd = 0;
  size_t gws_floats, slab_floats;
  float* gws2 = nullptr; float* gws_cur = nullptr;   // the side stream's split-K slabs / the one gemm() uses now
  hipStream_t s2 = nullptr; hipEvent_t ev[16] = {}; int overlap = 1;
  // bindings
  recnet_decoder_tensors dP{}, dGd{}, dM{}, dV{}, dVm{};
  recnet_reconstructor_tensors rP{}, rG{}, rM{}, rV{}, rVm{};
  bool dec_bound = false, rec_bound = false;
  OptGroup og[2];
  // state between forward and backward
  int T_last = 0, train_last = 0, fwd_dec_done = 0, fwd_rec_done = 0, rec_bwd_done = 0, early_opt_done = 0, norms_hoisted = 0;
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
  h->scal = take(64);
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
    h->dmpd = take(Tm * B * H); h->dmp = take(B * H);
    h->Xcat_g = takev(Tm * B * (size_t)h->ld2H); h->Hr_lp = takev(Tm * B * ldR); h->hrmean_lp = takev(B * ldR);
    h->dout_lp = takev(B * ldR); h->dGr = takev(Tm * B * ld4R);
    h->Wih_f = takev(4 * R * (size_t)h->ld2H); h->Whh_w = takev(4 * R * ldR);
    h->Whh_g = takev(4 * R * ldR);
  } else if (h->kind == RECNET_REC_LOCAL) {
    h->Ud = take(Tm * B * RA);
    h->Hr = take(F * B * R); h->Cr = take(F * B * R); h->acts_r = take(F * B * 4 * R);
    h->beta = take(F * B * Tm); h->Whr = take(F * B * RA); h->outl = take(F * B * R); h->dHr = take(F * B * R);
    h->dUd = take(Tm * B * RA); h->dwacc_r = take(RN_TCH * B * RA);
    h->Xcat_r = takev(F * B * ldHR); h->Hr_lp = takev(F * B * ldR); h->dout_lp = takev(F * B * ldR);
    h->dGr = takev(F * B * ld4R); h->dUd_lp = takev(Tm * B * ldRA); h->dWhr = takev(F * B * (size_t)h->ldRA4);
    h->dWhrs = takev(F * B * ldRA); h->Wr4_w = takev(RN_TCH * RA * ldR);
    h->Ur_w = takev(RA * ldH); h->Wr_w = takev(RA * ldR); h->Wihh_w = takev(4 * R * ldHR);
    h->slab2 = take(16 * B * R);
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
  h->dgru = c.decoder_cell == RECNET_CELL_GRU; h->rgru = c.reconstructor_type != RECNET_REC_NONE && c.reconstructor_cell == RECNET_CELL_GRU;
  {
    // fused recurrent step of the global reconstructor (rec_step.hpp): bf16 path, LSTM, B <= 112, R % 8 == 0, K <= 2048.
    // Opt-in (RN_FUSED_REC=1): at B=100, R=1536 it runs 15-18 us per step against 17.4 us for the GEMM + pointwise pair
    // (tools/micro/rec_probe.hip) — every workgroup re-reads the whole activation block from L2, which costs more than
    // the split-K slab round trip it removes; see DESIGN.md section 5.
    const char* e = getenv("RN_FUSED_REC");
    const int want = e ? atoi(e) : 0;
    h->fused_rec = want && h->lp && h->kind == RECNET_REC_GLOBAL && !h->rgru && h->B <= 112 && (h->R & 7) == 0 && h->R <= 2048;
  }
  {
    const char* e = getenv("RN_DEC_BWD_NT");
    h->use_wcomb_t = (e ? atoi(e) : 1) && h->lp && h->B <= 128;
  }
  h->need = carve(h, nullptr);
  *out = h;
  return RECNET_OK;
}

void recnet_destroy(recnet_handle* h) {
  if (!h) return;
  for (auto e : h->prof_ev) hipEventDestroy(e);
  for (int i = 0; i < 16; ++i) if (h->ev[i]) hipEventDestroy(h->ev[i]);
  if (h->s2) hipStreamDestroy(h->s2);
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
    if (h->fused_rec) { addr(1, R, h->Whh_g, h->ldR, 0, R, 0, R); o.pack[1].d[o.pack[1].ndst - 1].mode = 1; }
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
  h->gws_cur = h->gws;
  if (!h->s2) {
    const char* ov = getenv("RN_OVERLAP");
    h->overlap = ov ? atoi(ov) : 1;
    HIPCHK(hipStreamCreateWithFlags(&h->s2, hipStreamNonBlocking));
    for (int i = 0; i < 16; ++i) HIPCHK(hipEventCreateWithFlags(&h->ev[i], hipEventDisableTiming));
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

// ================================================================================================
// internal helpers
// ================================================================================================
#define REQUIRE_WS(h) do { if (!(h) || !(h)->ws) return fail(RECNET_ESTATE, "workspace not bound"); } while (0)
// kernels that read / write operand copies are templated on the operand type AT (bf16 | float)
#define LAUNCH_AT(h, kern, grid, block, smem, st, ...)                                          \
  do { if ((h)->lp) hipLaunchKernelGGL((kern<bf16_t>), grid, block, smem, st, __VA_ARGS__);     \
       else hipLaunchKernelGGL((kern<float>), grid, block, smem, st, __VA_ARGS__); } while (0)

inline void* at_off(const recnet_handle* h, void* p, size_t elems) {
  return (char*)p + elems * (h->lp ? 2 : 4);
}

// batched GEMM on operand buffers (AT) with automatic split-K; fp32 output
static void gemm(recnet_handle* h, const void* A, int a_col, int lda, const void* Bm, int b_col, int ldb, float* C, int ldc,
                 const float* bias, int M, int N, int K, float alpha, int acc, hipStream_t st) {
  int s = rn_pick_splitk(h->prec, M, N, K, 16, 0);
  while (s > 1 && (size_t)s * M * N > h->gws_floats) s >>= 1;
  rn_launch_gemm(h->prec, A, h->lp, a_col, lda, Bm, h->lp, b_col, ldb, C, ldc, bias, M, N, K, alpha, acc, s, h->gws_cur, 1, st);
}
// same, output written as an operand buffer (AT) — direct epilogue only
static void gemm_to_at(recnet_handle* h, const void* A, int a_col, int lda, const void* Bm, int b_col, int ldb, void* C,
                       int ldc, int M, int N, int K, hipStream_t st) {
  rn_launch_gemm(h->prec, A, h->lp, a_col, lda, Bm, h->lp, b_col, ldb, (float*)C, ldc, nullptr, M, N, K, 1.f, 0, 1, nullptr, 0,
                 st, 0, h->lp);
}
// recurrent-step GEMM: partial slabs only; returns the slab count the consumer must sum
static int gemm_slabs(recnet_handle* h, int tag, const void* A, int lda, const void* Bm, int b_col, int ldb, int M, int N,
                      int K, hipStream_t st, float* dst = nullptr) {
  float* slab = dst ? dst : h->slab;
  // split-K caps per site: more slices than this buy nothing for the GEMM (measured) and every slab is re-read by
  // the consumer kernel
  static const int cap_env = getenv("RN_SLAB_CAP") ? atoi(getenv("RN_SLAB_CAP")) : 0;
  static const int dbwd_cap = getenv("RN_DBWD_CAP") ? atoi(getenv("RN_DBWD_CAP")) : 16;
  int cap = (tag == RN_TAG_DEC_FWD) ? 4 : (tag == RN_TAG_DEC_BWD ? (b_col ? 8 : dbwd_cap) : 16);
  if (dst) cap = 8;
  if (cap_env) cap = cap_env;
  int s = rn_pick_splitk(h->prec, M, N, K, cap, 1);
  while (s > 1 && (size_t)s * M * N > h->slab_floats) s >>= 1;
  if (s < 2) s = 2;  // always use the slab path so the consumer code is uniform
  s = rn_effective_splitk(h->prec, K, s);
  hipEvent_t e0 = nullptr, e1 = nullptr;
  if (h->prof_on == tag) {
    if (h->prof_used + 2 > h->prof_ev.size()) {
      hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
      h->prof_ev.push_back(a); h->prof_ev.push_back(b);
    }
    e0 = h->prof_ev[h->prof_used++]; e1 = h->prof_ev[h->prof_used++];
    hipEventRecord(e0, st);
  }
  if (s < 2) {   // K fits one tile: the single product goes to slab 0 through the direct epilogue
    rn_launch_gemm(h->prec, A, h->lp, 0, lda, Bm, h->lp, b_col, ldb, slab, N, nullptr, M, N, K, 1.f, 0, 1, nullptr, 0, st, tag);
    s = 1;
  } else {
    rn_launch_gemm(h->prec, A, h->lp, 0, lda, Bm, h->lp, b_col, ldb, nullptr, N, nullptr, M, N, K, 1.f, 0, s, slab, 0, st, tag);
  }
  if (e1) hipEventRecord(e1, st);
  return s;
}
template <typename ST>
static void colsum_t(const ST* X, int rows, int cols, int ld, float* out, hipStream_t st, int zeroed = 0) {
  if (rows >= 64 && (ld & 7) == 0 && (((uintptr_t)X) & 15) == 0) {
    int rs = rows / 96; rs = rs < 1 ? 1 : (rs > 64 ? 64 : rs);
    if (!zeroed) hipMemsetAsync(out, 0, (size_t)cols * 4, st);
    hipLaunchKernelGGL(colsum_vec_kernel<ST>, dim3(cdiv(cols, 512), rs), dim3(256), 0, st, X, rows, cols, ld, out);
    return;
  }
  int rs = rows >= 512 ? 8 : 1;
  if (rs > 1 && !zeroed) hipMemsetAsync(out, 0, (size_t)cols * 4, st);
  hipLaunchKernelGGL(colsum_kernel<ST>, dim3(cdiv(cols, 64), rs), dim3(256), 0, st, X, rows, cols, ld, out, rs > 1 ? 1 : 0);
}
// zeroed: `out` is one of the buffers the hoisted zero_list_kernel of this step cleared
static void colsum_at(recnet_handle* h, const void* X, int rows, int cols, int ld, float* out, hipStream_t st, int zeroed = 0) {
  if (h->lp) colsum_t<bf16_t>((const bf16_t*)X, rows, cols, ld, out, st, zeroed);
  else colsum_t<float>((const float*)X, rows, cols, ld, out, st, zeroed);
}
// fork: `side` continues after everything enqueued on `main` so far; join: `main` waits for `side`.  Under
// stream capture these become graph edges, so independent work runs in parallel branches of the hipGraph.
static void gate_bias_grad(recnet_handle* h, const void* dG, int rows, int Hd, int ld, float* dbih, float* dbhh, int gru, hipStream_t st) {
  float* tmp = (dbih == h->dGd.rnn_bias_ih_l0) ? h->bsum4 : h->bsum4r;      // decoder / reconstructor
  colsum_at(h, dG, rows, 4 * Hd, ld, tmp, st, h->prezeroed);
  hipLaunchKernelGGL(gate_bias_grad_kernel, dim3(cdiv(4 * Hd, 256)), dim3(256), 0, st, tmp, dbih, dbhh, Hd, gru);
}
// dW_hh = dG^T . Hprev over `rows` rows.  LSTM: the 4 gate blocks as they are.  GRU: master blocks (r, z) come from
// packed blocks (0, 1) and master block n from packed block 3 (the hidden-side n pre-activation, see gru_point).
static void dW_hh(recnet_handle* h, int gru, int Hd, const void* dG, int ld_dg, const void* Hprev, int ld_h, float* dW, int rows,
                  int acc, hipStream_t st) {
  if (!gru) { gemm(h, dG, 1, ld_dg, Hprev, 1, ld_h, dW, Hd, nullptr, 4 * Hd, Hd, rows, 1.f, acc, st); return; }
  gemm(h, dG, 1, ld_dg, Hprev, 1, ld_h, dW, Hd, nullptr, 2 * Hd, Hd, rows, 1.f, acc, st);
  gemm(h, at_off(h, (void*)dG, (size_t)3 * Hd), 1, ld_dg, Hprev, 1, ld_h, dW + (size_t)2 * Hd * Hd, Hd, nullptr, Hd, Hd, rows, 1.f, acc, st);
}
static void fork_to(recnet_handle* h, int e, hipStream_t main, hipStream_t side) {
  hipEventRecord(h->ev[e], main); hipStreamWaitEvent(side, h->ev[e], 0);
}
static void join_from(recnet_handle* h, int e, hipStream_t main, hipStream_t side) {
  hipEventRecord(h->ev[e], side); hipStreamWaitEvent(main, h->ev[e], 0);
}
static void copyf(const float* x, float* y, size_t n, hipStream_t st) {
  hipMemcpyAsync(y, x, n * 4, hipMemcpyDeviceToDevice, st);
}
// dst (AT) [rows][ld_dst] <- scale * src fp32 [rows][.. ld_src], zero padded
static void pack_block(recnet_handle* h, void* dst, int ld_dst, const float* src, int ld_src, int rows, int cols, float scale,
                       hipStream_t st) {
  const size_t n = (size_t)rows * ld_dst;
  if (h->lp) hipLaunchKernelGGL(pack_block_kernel<bf16_t>, dim3(ew_blocks(n)), dim3(256), 0, st, (bf16_t*)dst, ld_dst, src, ld_src, rows, cols, scale);
  else hipLaunchKernelGGL(pack_block_kernel<float>, dim3(ew_blocks(n)), dim3(256), 0, st, (float*)dst, ld_dst, src, ld_src, rows, cols, scale);
}
static void param_norms(recnet_handle* h, int g, float* sum_out, hipStream_t st) {
  OptGroup& o = h->og[g];
  hipLaunchKernelGGL(sumsq_chunk_kernel, dim3(o.nchunks), dim3(256), 0, st, o.d_tab, o.d_chunks, 0, (const float*)nullptr, 0.f, o.d_partial);
  hipLaunchKernelGGL(tensor_norm_kernel, dim3(o.ntens), dim3(256), 0, st, o.d_tab, o.d_partial, o.d_pnorm);
  hipLaunchKernelGGL(norm_finalize_kernel, dim3(1), dim3(64), 0, st, o.d_pnorm, o.ntens, 0.f, (float*)nullptr, (float*)nullptr, sum_out);
}
__global__ void fill_i64_kernel(int64_t* p, int64_t v, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}
// out = a + k * b
__global__ void axpb_kernel(const float* a, const float* b, float k, float* out) { *out = *a + k * *b; }
__global__ void export_scalars_kernel(const float* scal, recnet_scalars* out) {
  out->dec_ce = scal[0]; out->dec_reg = scal[1]; out->dec_loss = scal[2]; out->rec_mse = scal[3];
  out->rec_reg = scal[4]; out->rec_loss = scal[5]; out->total_loss = scal[6]; out->dec_grad_norm = scal[7];
}

static inline GateMap gmap_ih(int gru) { GateMap m; m.m[0] = 0; m.m[1] = 1; m.m[2] = 2; m.m[3] = gru ? -1 : 3; return m; }
static inline GateMap gmap_hh(int gru) { GateMap m; m.m[0] = 0; m.m[1] = 1; m.m[2] = gru ? -1 : 2; m.m[3] = gru ? 2 : 3; return m; }
// recurrent weight image in the 4-block gate layout: dst[4 Hd][ld] = [src1 cols | src2 cols | 0]
static void pack_gates(recnet_handle* h, void* dst, int ld_dst, int Hd, const float* src1, int ld1, int c1, GateMap m1,
                       const float* src2, int ld2, int c2, GateMap m2, hipStream_t st) {
  const size_t n = (size_t)4 * Hd * ld_dst;
  if (h->lp) hipLaunchKernelGGL(pack_gates_kernel<bf16_t>, dim3(ew_blocks(n)), dim3(256), 0, st, (bf16_t*)dst, ld_dst, Hd, src1, ld1, c1, m1, src2, ld2, c2, m2);
  else hipLaunchKernelGGL(pack_gates_kernel<float>, dim3(ew_blocks(n)), dim3(256), 0, st, (float*)dst, ld_dst, Hd, src1, ld1, c1, m1, src2, ld2, c2, m2);
}
static void gate_bias(const float* bih, const float* bhh, float* out, int Hd, int gru, hipStream_t st) {
  hipLaunchKernelGGL(gate_bias_kernel, dim3(cdiv(4 * Hd, 256)), dim3(256), 0, st, bih, bhh, out, Hd, gru);
}
// both bias gradients from the 4-block gate gradients dG [rows][4 Hd]
static void gate_bias_grad(recnet_handle* h, const void* dG, int rows, int Hd, int ld, float* dbih, float* dbhh, int gru, hipStream_t st);

// WcombT = Wcomb^T (after Wcomb changed: pack_weights, the decoder's Adam step)
static void refresh_wcomb_t(recnet_handle* h, hipStream_t st) {
  if (!h->use_wcomb_t) return;
  const int KW = 4 * h->H + RN_FCH * h->A, H = h->H;
  dim3 grid(cdiv(H, 32), cdiv(h->ldKW, 32));           // source = Wcomb [KW][ldH] (rows beyond KW read as zero -> pad)
  hipLaunchKernelGGL(transpose_at_kernel<bf16_t>, grid, dim3(256), 0, st, (const bf16_t*)h->Wcomb, h->ldH, KW, H, (bf16_t*)h->WcombT, h->ldKW);
}

// Packed operand images of the weights (AT, zero padded leading dimensions), refreshed after every optimiser step.
static int pack_weights(recnet_handle* h, hipStream_t st) {
  const int H = h->H, D = h->D, E = h->E, A = h->A, V = h->V, R = h->R, RA = h->RA;
  if (h->dec_bound) {
    pack_block(h, h->U_w, h->ldD, h->dP.attn_U_weight, D, A, D, 1.f, st);
    const GateMap none = gmap_ih(0);
    pack_gates(h, h->Wc_w, h->ldD, H, h->dP.rnn_weight_ih_l0 + E, E + D, D, gmap_ih(h->dgru), nullptr, 0, 0, none, st);
    pack_gates(h, h->We_w, h->ldE, H, h->dP.rnn_weight_ih_l0, E + D, E, gmap_ih(h->dgru), nullptr, 0, 0, none, st);
    pack_gates(h, h->Wcomb, h->ldH, H, h->dP.rnn_weight_hh_l0, H, H, gmap_hh(h->dgru), nullptr, 0, 0, none, st);   // [W_hh ; W ; W ; W ; W]
    for (int j = 0; j < RN_FCH; ++j)
      pack_block(h, at_off(h, h->Wcomb, (size_t)(4 * H + j * A) * h->ldH), h->ldH, h->dP.attn_W_weight, H, A, H, 1.f, st);
    pack_block(h, h->Wo_w, h->ldH, h->dP.out_weight, H, V, H, 1.f, st);
    refresh_wcomb_t(h, st);
  }
  if (h->rec_bound) {
    pack_block(h, h->Wor_w, h->ldR, h->rP.out_weight, R, R, R, 1.f, st);
    if (h->kind == RECNET_REC_GLOBAL) {
      const GateMap none = gmap_ih(0);
      pack_gates(h, h->Wih_f, h->ld2H, R, h->rP.rnn_weight_ih_l0, 2 * H, 2 * H, gmap_ih(h->rgru), nullptr, 0, 0, none, st);
      pack_gates(h, h->Whh_w, h->ldR, R, h->rP.rnn_weight_hh_l0, R, R, gmap_hh(h->rgru), nullptr, 0, 0, none, st);
      if (h->fused_rec)
        hipLaunchKernelGGL(pack_interleave_kernel<bf16_t>, dim3(ew_blocks((size_t)4 * R * h->ldR)), dim3(256), 0, st, (bf16_t*)h->Whh_g, h->ldR, R, h->rP.rnn_weight_hh_l0, R, R);
    } else {
      pack_block(h, h->Ur_w, h->ldH, h->rP.attn_U_weight, H, RA, H, 1.f, st);
      pack_block(h, h->Wr_w, h->ldR, h->rP.attn_W_weight, R, RA, R, 1.f, st);
      for (int j = 0; j < RN_TCH; ++j)   // [W_r ; W_r ; W_r ; W_r]: sums the per-chunk dWhr partials in the GEMM's K loop
        pack_block(h, at_off(h, h->Wr4_w, (size_t)j * RA * h->ldR), h->ldR, h->rP.attn_W_weight, R, RA, R, 1.f, st);
      // [W_ih | W_hh | 0]
      pack_gates(h, h->Wihh_w, h->ldHR, R, h->rP.rnn_weight_ih_l0, H, H, gmap_ih(h->rgru), h->rP.rnn_weight_hh_l0, R, R, gmap_hh(h->rgru), st);
    }
  }
  return RECNET_OK;
}

static void launch_dec_cell(recnet_handle* h, const DecCellArgs& a, hipStream_t st) {
  static const int vec_env = getenv("RN_DEC_VEC") ? atoi(getenv("RN_DEC_VEC")) : 1;
  if (vec_env && h->lp && (h->H & 7) == 0 && h->F <= 32 && h->A <= 128 && (h->ld4H & 7) == 0) {
    const size_t smv = (size_t)(h->A + ((h->F + 3) & ~3) + 4 * 512 + 16) * 4;
    hipLaunchKernelGGL(dec_cell_vec_kernel<bf16_t>, dim3(h->B, cdiv(h->H, 512)), dim3(256), smv, st, a);
    return;
  }
  // units per workgroup: 256 (1024 threads) when H allows, so a caption is covered by H/256 workgroups
  int uc = h->H >= 256 ? 256 : (h->H >= 128 ? 128 : 64);
  static const char* e = getenv("RN_DEC_UC");
  if (e) uc = atoi(e);
  dim3 grid(h->B, cdiv(h->H, uc));
  const size_t sm = (size_t)(h->A + h->F + 4 * uc + 16) * 4;
  LAUNCH_AT(h, dec_cell_kernel, grid, dim3(4 * uc), sm, st, a);
}
// loop-invariant products of the decoder: Uv = enc . U^T (decoder.py:54) and P = enc . W_ih[:, E:]^T
static void dec_invariants(recnet_handle* h, const float* enc, hipStream_t st) {
  const int B = h->B, F = h->F, D = h->D, H = h->H, A = h->A;
  gate_bias(h->dP.rnn_bias_ih_l0, h->dP.rnn_bias_hh_l0, h->bsum_d, H, h->dgru, st);
  pack_block(h, h->enc_lp, h->ldD, enc, D, B * F, D, 1.f, st);
  gemm(h, h->enc_lp, 0, h->ldD, h->U_w, 0, h->ldD, h->Uv, A, nullptr, B * F, A, D, 1.f, 0, st);
  gemm_to_at(h, h->enc_lp, 0, h->ldD, h->Wc_w, 0, h->ldD, h->P, h->ld4H, B * F, 4 * H, D, st);
}
static void embed_fwd(recnet_handle* h, const int64_t* targets, const int64_t* tokens, int rows, int train, int t0, hipStream_t st,
                      size_t row0 = 0) {
  const DropDesc dd = mkdrop(h, RN_SITE_DEC_EMBED, h->c.embedding_dropout, train);
  void* dst = at_off(h, h->emb_lp, row0 * h->ldE);
  if (h->lp) hipLaunchKernelGGL(embed_fwd_kernel<bf16_t>, dim3(rows), dim3(128), 0, st, h->dP.embedding_weight, targets, tokens, (bf16_t*)dst, h->ldE, h->B, h->E, h->V, h->c.embedding_scale, dd, t0);
  else hipLaunchKernelGGL(embed_fwd_kernel<float>, dim3(rows), dim3(128), 0, st, h->dP.embedding_weight, targets, tokens, (float*)dst, h->ldE, h->B, h->E, h->V, h->c.embedding_scale, dd, t0);
}

// ---------------------------------------------------------------------------------------------- decoder forward
// the dependent chain: invariants, embeddings, T x (GEMM + cell kernel)
// free_tokens != nullptr: free-running decoding (train.py:46-51) — the input of step t+1 is the argmax of step t's
// (dropped-out) logits, written to free_tokens [T][B]; the per-step embedding / input projection / vocabulary projection
// then sit inside the chain.  Forward only.
static int dec_fwd_chain(recnet_handle* h, const float* enc, const int64_t* targets, int T, int train, hipStream_t st,
                         int64_t* free_tokens = nullptr) {
  const int B = h->B, F = h->F, E = h->E, H = h->H, A = h->A, V = h->V;
  if (!h->norms_hoisted) param_norms(h, 0, h->scal + 1, st);
  dec_invariants(h, enc, st);
  if (!free_tokens) {
    // all T teacher-forced input embeddings at once    (decoder.py:46-48, train.py:25,45)
    embed_fwd(h, targets, nullptr, T * B, train, 0, st);
    // Xe = emb . W_ih[:, :E]^T + b_ih + b_hh
    gemm(h, h->emb_lp, 0, h->ldE, h->We_w, 0, h->ldE, h->Xe, 4 * H, h->bsum_d, T * B, 4 * H, E, 1.f, 0, st);
  } else {
    hipLaunchKernelGGL(fill_i64_kernel, dim3(cdiv(B, 256)), dim3(256), 0, st, h->sr_tok[0], (int64_t)1, B);   // <SOS>, train.py:25
  }
  DecCellArgs a;
  a.B = B; a.F = F; a.H = H; a.A = A;
  a.P = h->P; a.ldp = h->ld4H; a.Uv = h->Uv; a.ab = h->dP.attn_b; a.w = h->dP.attn_w_weight;
  a.ld_hlp = h->ldH; a.gru = h->dgru;
  const float* prev_state = h->dgru ? h->Hs : h->Cs;   // what the pointwise part carries: h_{t-1} (GRU) / c_{t-1}
  for (int t = 0; t < T; ++t) {
    if (free_tokens) {
      const int64_t* fed = t == 0 ? h->sr_tok[0] : free_tokens + (size_t)(t - 1) * B;
      hipMemcpyAsync(h->in_tok + (size_t)t * B, fed, (size_t)B * 8, hipMemcpyDeviceToDevice, st);
      embed_fwd(h, nullptr, fed, B, train, t, st, (size_t)t * B);
      gemm(h, at_off(h, h->emb_lp, (size_t)t * B * h->ldE), 0, h->ldE, h->We_w, 0, h->ldE, h->Xe + (size_t)t * B * 4 * H, 4 * H,
           h->bsum_d, B, 4 * H, E, 1.f, 0, st);
    }
    int S = 0;
    if (t > 0)   // h_{t-1} . [W_hh ; attn_W]^T  -> recurrent gate part + Wh of the attention
      S = gemm_slabs(h, RN_TAG_DEC_FWD, at_off(h, h->Hs_lp, (size_t)(t - 1) * B * h->ldH), h->ldH, h->Wcomb, 0, h->ldH, B, 4 * H + A, H, st);
    a.t = t; a.S = S; a.slab = t > 0 ? h->slab : nullptr;
    a.Xe = h->Xe + (size_t)t * B * 4 * H;
    a.c_prev = t > 0 ? prev_state + (size_t)(t - 1) * B * H : nullptr;
    a.h_out = h->Hs + (size_t)t * B * H; a.c_out = h->Cs + (size_t)t * B * H;
    a.h_lp = at_off(h, h->Hs_lp, (size_t)t * B * h->ldH);
    a.acts = h->acts + (size_t)t * B * 4 * H;
    a.Wh_out = h->Wh + (size_t)t * B * A; a.att_out = h->att + (size_t)t * B * F;
    launch_dec_cell(h, a, st);
    if (free_tokens) {   // logits_t, then argmax of what Decoder.forward returns (the dropped-out logits, decoder.py:68-69)
      float* lg = h->logits + (size_t)t * B * V;
      gemm(h, at_off(h, h->Hs_lp, (size_t)t * B * h->ldH), 0, h->ldH, h->Wo_w, 0, h->ldH, lg, V, h->dP.out_bias, B, V, H, 1.f, 0, st);
      const float* pick = lg;
      if (train && h->c.decoder_out_dropout > 0.f) {
        copyf(lg, h->sr_logits, (size_t)B * V, st);
        hipLaunchKernelGGL(logits_drop_kernel, dim3(ew_blocks((size_t)B * V)), dim3(256), 0, st, h->sr_logits, B, V,
                           mkdrop(h, RN_SITE_DEC_LOGIT, h->c.decoder_out_dropout, train), t);
        pick = h->sr_logits;
      }
      hipLaunchKernelGGL(argmax_rows_kernel, dim3(B), dim3(256), 0, st, pick, V, V, free_tokens + (size_t)t * B);
    }
  }
  h->T_last = T; h->train_last = train;
  return RECNET_OK;
}
// vocabulary projection + loss for all steps (decoder.py:68-69, train.py:54-68); independent of the reconstructor
static int dec_fwd_loss(recnet_handle* h, const int64_t* targets, int T, const float* stepw, int train, hipStream_t st,
                        int have_logits = 0) {
  const int B = h->B, H = h->H, V = h->V;
  if (!have_logits) gemm(h, h->Hs_lp, 0, h->ldH, h->Wo_w, 0, h->ldH, h->logits, V, h->dP.out_bias, T * B, V, H, 1.f, 0, st);
  copyf(stepw, h->stepw, T, st);
  {
    const DropDesc dd = mkdrop(h, RN_SITE_DEC_LOGIT, h->c.decoder_out_dropout, train);
    if (h->lp) hipLaunchKernelGGL(ce_kernel<bf16_t>, dim3(T * B), dim3(256), 0, st, h->logits, targets, h->stepw, h->rowloss, (bf16_t*)h->dlog_lp, h->ldV, B, V, dd);
    else hipLaunchKernelGGL(ce_kernel<float>, dim3(T * B), dim3(256), 0, st, h->logits, targets, h->stepw, h->rowloss, (float*)h->dlog_lp, h->ldV, B, V, dd);
  }
  hipLaunchKernelGGL(reduce_sum_kernel, dim3(1), dim3(256), 0, st, h->rowloss, T * B, h->scal + 0, 1.0f);
  hipLaunchKernelGGL(axpb_kernel, dim3(1), dim3(1), 0, st, h->scal + 0, h->scal + 1, h->c.decoder_lambda_reg, h->scal + 2);
  hipLaunchKernelGGL(axpb_kernel, dim3(1), dim3(1), 0, st, h->scal + 2, h->scal + 2, 0.f, h->scal + 6);
  return RECNET_OK;
}
static int fwd_decoder(recnet_handle* h, const float* enc, const int64_t* targets, int T, const float* stepw,
                       int train, float* hiddens_out, hipStream_t st, int64_t* free_tokens = nullptr) {
  int r = dec_fwd_chain(h, enc, targets, T, train, st, free_tokens); if (r) return r;
  r = dec_fwd_loss(h, targets, T, stepw, train, st, free_tokens != nullptr); if (r) return r;
  if (hiddens_out) copyf(h->Hs, hiddens_out, (size_t)T * h->B * h->H, st);
  // free_fwd: the backward scatters the embedding gradient by the tokens that were fed (in_tok), not by the targets
  h->fwd_dec_done = 1; h->free_fwd = free_tokens ? 1 : 0; h->fwd_rec_done = 0; h->rec_bwd_done = 0;
  return RECNET_OK;
}

// ---------------------------------------------------------------------------------------------- decoder backward
// output layer: dHs_out = dlogits . W_o, dW_o, db_o — needs only the decoder forward
static int dec_bwd_out(recnet_handle* h, float gscale, hipStream_t st) {
  const int B = h->B, H = h->H, V = h->V, T = h->T_last, TB = T * B;
  if (gscale != 1.0f) {
    const size_t n = (size_t)TB * h->ldV;
    if (h->lp) hipLaunchKernelGGL(scale_kernel<bf16_t>, dim3(ew_blocks(n)), dim3(256), 0, st, (bf16_t*)h->dlog_lp, n, gscale);
    else hipLaunchKernelGGL(scale_kernel<float>, dim3(ew_blocks(n)), dim3(256), 0, st, (float*)h->dlog_lp, n, gscale);
  }
  gemm(h, h->dlog_lp, 0, h->ldV, h->Wo_w, 1, h->ldH, h->dHs, H, nullptr, TB, H, V, 1.f, 0, st);
  // dW_o = dlogits^T . Hs ; db_o = colsum(dlogits)
  gemm(h, h->dlog_lp, 1, h->ldV, h->Hs_lp, 1, h->ldH, h->dGd.out_weight, H, nullptr, V, H, TB, 1.f, 0, st);
  colsum_at(h, h->dlog_lp, TB, V, h->ldV, h->dGd.out_bias, st, h->prezeroed);
  return RECNET_OK;
}
// BPTT chain; dh_t (direct) = dHs_out[t] + dhid[t] (the reconstructor's gradient w.r.t. the hidden states)
static int dec_bwd_chain(recnet_handle* h, const float* dhid, hipStream_t st) {
  const int B = h->B, F = h->F, H = h->H, A = h->A, T = h->T_last;
  const int ldWS = h->ldWS, KW = 4 * H + RN_FCH * A;
  // per step one fused kernel over (caption, frame chunk) + one split-K GEMM (dgates_t | dWh_t chunks) . [W_hh ; W x4]
  DecCellBwdArgs a;
  a.B = B; a.F = F; a.H = H; a.A = A;
  a.P = h->P; a.ldp = h->ld4H; a.Uv = h->Uv; a.ab = h->dP.attn_b; a.w = h->dP.attn_w_weight;
  a.dUv = h->dUv; a.dwacc = h->dwacc; a.ld_dgx = ldWS; a.dUv_lp = h->dUv_lp; a.ld_dUv = h->ldA; a.gru = h->dgru;
  const float* prev_state = h->dgru ? h->Hs : h->Cs;
  const int Asz = A <= 256 ? 256 : A;
  const size_t sm = (size_t)(4 * H + F + 2 * Asz + 16) * 4;
  int S = 0;
  for (int t = T - 1; t >= 0; --t) {
    a.t = t; a.S = S; a.slab = (t < T - 1) ? h->slab : nullptr; a.first = (t == T - 1); a.last = (t == 0);
    a.dHs = h->dHs + (size_t)t * B * H; a.dHs2 = dhid ? dhid + (size_t)t * B * H : nullptr;
    a.acts = h->acts + (size_t)t * B * 4 * H;
    a.c = h->Cs + (size_t)t * B * H;
    a.c_prev = t > 0 ? prev_state + (size_t)(t - 1) * B * H : nullptr;
    a.dc_in = h->dc_carry + (size_t)((t + 1) & 1) * B * H; a.dc_out = h->dc_carry + (size_t)(t & 1) * B * H;
    a.dGx = at_off(h, h->dGx, (size_t)t * B * ldWS);
    a.Wh = h->Wh + (size_t)t * B * A;
    LAUNCH_AT(h, dec_cell_bwd_kernel, dim3(B, RN_FCH), dim3(256), sm, st, a);
    if (t > 0)
      S = h->use_wcomb_t ? gemm_slabs(h, RN_TAG_DEC_BWD, at_off(h, h->dGx, (size_t)t * B * ldWS), ldWS, h->WcombT, 0, h->ldKW, B, H, KW, st)
                         : gemm_slabs(h, RN_TAG_DEC_BWD, at_off(h, h->dGx, (size_t)t * B * ldWS), ldWS, h->Wcomb, 1, h->ldH, B, H, KW, st);
  }
  return RECNET_OK;
}
// Deferred weight gradients of the decoder for the time steps [t0, t1) (rows [t0 B, t1 B) of dGx): every product whose
// contraction runs over (t, b).  acc = 0 for the first range processed (it also zeroes dEmb), 1 afterwards.  Ranges can be
// issued as soon as the BPTT chain has produced their rows, on another stream, while the chain goes on.
static int dec_bwd_deferred_rows(recnet_handle* h, const float* enc, const int64_t* targets, int t0, int t1, int acc, hipStream_t st) {
  const int B = h->B, F = h->F, D = h->D, E = h->E, H = h->H, A = h->A, V = h->V;
  const int train = h->train_last, ldWS = h->ldWS, nrow = (t1 - t0) * B;
  const size_t r0 = (size_t)t0 * B;
  const int GH = (h->dgru ? 3 : 4) * H;   // rows of the master W_ih / W_hh: gate blocks (r, z, n) or (i, f, g, o)
  void* dG = at_off(h, h->dGx, r0 * ldWS);
  // dgates live in columns [0,4H) of dGx, dWh chunks behind them
  {
    gemm(h, dG, 0, ldWS, h->We_w, 1, h->ldE, h->demb + r0 * E, E, nullptr, nrow, E, 4 * H, 1.f, 0, st);
    if (!acc && !h->prezeroed) hipMemsetAsync(h->dGd.embedding_weight, 0, (size_t)V * E * 4, st);
    hipLaunchKernelGGL(embed_bwd_kernel, dim3(nrow), dim3(128), 0, st, h->dGd.embedding_weight, targets, h->demb, B, E, V,
                       h->c.embedding_scale, mkdrop(h, RN_SITE_DEC_EMBED, h->c.embedding_dropout, train), (int)r0,
                       (const int64_t*)(h->free_fwd ? h->in_tok : nullptr));
    hipLaunchKernelGGL(embed_bwd_hot_kernel, dim3(cdiv(E, 128), cdiv(nrow, RN_HOT_ROWS)), dim3(128), 0, st, h->dGd.embedding_weight, targets, h->demb, B, E, V,
                       h->c.embedding_scale, mkdrop(h, RN_SITE_DEC_EMBED, h->c.embedding_dropout, train), (int)r0, nrow,
                       (const int64_t*)(h->free_fwd ? h->in_tok : nullptr));
    gemm(h, dG, 1, ldWS, at_off(h, h->emb_lp, r0 * h->ldE), 1, h->ldE, h->dGd.rnn_weight_ih_l0, E + D, nullptr, GH, E, nrow, 1.f, acc, st);
  }
  // ctx_t = (1/F) sum_f a_t[f] enc[b,f] for these steps (only needed here), then dW_ih[:, E:] (+)= dgates^T . ctx
  {
    dim3 grid(B, cdiv(h->ldD, 256));
    const float* att = h->att + r0 * F;
    void* ctx = at_off(h, h->ctx_lp, r0 * h->ldD);
    const int Tn = t1 - t0;
    if (h->Tm <= 32) {
      if (h->lp) hipLaunchKernelGGL(ctx_all_kernel<bf16_t>, grid, dim3(256), (size_t)32 * F * 4, st, att, enc, (bf16_t*)ctx, h->ldD, Tn, B, F, D);
      else hipLaunchKernelGGL(ctx_all_kernel<float>, grid, dim3(256), (size_t)32 * F * 4, st, att, enc, (float*)ctx, h->ldD, Tn, B, F, D);
    } else {
      if (h->lp) hipLaunchKernelGGL(ctx_all_slow_kernel<bf16_t>, grid, dim3(256), 0, st, att, enc, (bf16_t*)ctx, h->ldD, Tn, B, F, D);
      else hipLaunchKernelGGL(ctx_all_slow_kernel<float>, grid, dim3(256), 0, st, att, enc, (float*)ctx, h->ldD, Tn, B, F, D);
    }
    gemm(h, dG, 1, ldWS, ctx, 1, h->ldD, h->dGd.rnn_weight_ih_l0 + E, E + D, nullptr, GH, D, nrow, 1.f, acc, st);
  }
  // dWh_t = sum of its RN_FCH frame-chunk partials (operand of dW_attn and source of d attn_b)
  {
    const size_t n = (size_t)nrow * h->ldA;
    void* dst = at_off(h, h->dWhs, r0 * h->ldA);
    if (h->lp) hipLaunchKernelGGL(sum_chunks_kernel<bf16_t>, dim3(ew_blocks(n)), dim3(256), 0, st, (bf16_t*)dst, h->ldA, (const bf16_t*)dG + 4 * H, ldWS, nrow, A, RN_FCH);
    else hipLaunchKernelGGL(sum_chunks_kernel<float>, dim3(ew_blocks(n)), dim3(256), 0, st, (float*)dst, h->ldA, (const float*)dG + 4 * H, ldWS, nrow, A, RN_FCH);
  }
  // dW_hh (+)= sum_{t>=1} dgates_t^T h_{t-1} ; dW_attn (+)= sum_{t>=1} dWh_t^T h_{t-1}   (h_{-1} = 0)
  const int ta = t0 > 1 ? t0 : 1;
  if (t1 > ta) {
    const int nr = (t1 - ta) * B;
    const void* hp = at_off(h, h->Hs_lp, (size_t)(ta - 1) * B * h->ldH);
    dW_hh(h, h->dgru, H, at_off(h, h->dGx, (size_t)ta * B * ldWS), ldWS, hp, h->ldH, h->dGd.rnn_weight_hh_l0, nr, acc, st);
    gemm(h, at_off(h, h->dWhs, (size_t)ta * B * h->ldA), 1, h->ldA, hp, 1, h->ldH, h->dGd.attn_W_weight, H, nullptr, A, H, nr, 1.f, acc, st);
  } else if (!acc) {
    hipMemsetAsync(h->dGd.rnn_weight_hh_l0, 0, (size_t)GH * H * 4, st);
    hipMemsetAsync(h->dGd.attn_W_weight, 0, (size_t)A * H * 4, st);
  }
  return RECNET_OK;
}
// what needs the whole chain: bias gradients (column sums over all rows), d attn_U (dUv is complete after step 0), d attn_w
static int dec_bwd_deferred_tail(recnet_handle* h, hipStream_t st) {
  const int B = h->B, F = h->F, D = h->D, H = h->H, A = h->A, T = h->T_last, TB = T * B, ldWS = h->ldWS;
  gate_bias_grad(h, h->dGx, TB, H, ldWS, h->dGd.rnn_bias_ih_l0, h->dGd.rnn_bias_hh_l0, h->dgru, st);
  gemm(h, h->dUv_lp, 1, h->ldA, h->enc_lp, 1, h->ldD, h->dGd.attn_U_weight, D, nullptr, A, D, B * F, 1.f, 0, st);
  colsum_t<float>(h->dwacc, RN_FCH * B, A, A, h->dGd.attn_w_weight, st, h->prezeroed);
  colsum_at(h, h->dWhs, TB, A, h->ldA, h->dGd.attn_b, st, h->prezeroed);
  return RECNET_OK;
}
static int dec_bwd_deferred(recnet_handle* h, const float* enc, const int64_t* targets, hipStream_t st) {
  int r = dec_bwd_deferred_rows(h, enc, targets, 0, h->T_last, 0, st); if (r) return r;
  return dec_bwd_deferred_tail(h, st);
}
static int bwd_decoder(recnet_handle* h, const float* enc, const int64_t* targets, const float* dhid, float gscale,
                       hipStream_t st) {
  int r = dec_bwd_out(h, gscale, st); if (r) return r;
  r = dec_bwd_chain(h, dhid, st); if (r) return r;
  return dec_bwd_deferred(h, enc, targets, st);
}

// ---------------------------------------------------------------------------------------------- global reconstructor
static void lstm_pw(recnet_handle* h, int Hd, int S, int slab_ld, const float* X, int x_ld, const float* b1, const float* b2,
                    const float* c_prev, float* h_out, void* h_lp, int hlp_ld, void* h_lp2, int hlp2_ld, float* c_out,
                    float* acts, hipStream_t st) {
  LstmPwArgs p;
  p.B = h->B; p.Hd = Hd; p.S = S; p.gru = h->rgru; p.slab = h->slab; p.slab_stride = (size_t)h->B * slab_ld; p.slab_ld = slab_ld;
  p.X = X; p.x_ld = x_ld; p.b1 = b1; p.b2 = b2; p.c_prev = c_prev; p.h_out = h_out; p.h_ld = Hd;
  p.h_lp = h_lp; p.hlp_ld = hlp_ld; p.hlp_pad_from = Hd; p.h_lp2 = h_lp2; p.hlp2_ld = hlp2_ld; p.c_out = c_out; p.acts = acts;
  LAUNCH_AT(h, lstm_pw_kernel, dim3(cdiv((long)h->B * Hd, 256)), dim3(256), 0, st, p);
}
static void mean_over_t(recnet_handle* h, const float* X, int T, int Cn, float scale, float* out, void* out_lp, int ld_lp, hipStream_t st) {
  const size_t n = (size_t)h->B * (out_lp ? ld_lp : Cn);
  if (h->lp) hipLaunchKernelGGL(mean_over_t_kernel<bf16_t>, dim3(ew_blocks(n)), dim3(256), 0, st, X, T, h->B, Cn, scale, out, (bf16_t*)out_lp, ld_lp);
  else hipLaunchKernelGGL(mean_over_t_kernel<float>, dim3(ew_blocks(n)), dim3(256), 0, st, X, T, h->B, Cn, scale, out, (float*)out_lp, ld_lp);
}

static int fwd_rec_global(recnet_handle* h, const float* enc, int T, int train, hipStream_t st) {
  const int B = h->B, F = h->F, D = h->D, H = h->H, R = h->R;
  if (!h->norms_hoisted) {
    param_norms(h, 1, h->scal + 4, st);
    gate_bias(h->rP.rnn_bias_ih_l0, h->rP.rnn_bias_hh_l0, h->bsum_r, R, h->rgru, st);
  }
  // mean-pooled decoder states, rescaled by caption_max_len / T (global_reconstructor.py:33-37): (cml / T^2) sum_t h_t
  mean_over_t(h, h->Hs, T, H, (float)h->cml / ((float)T * (float)T), h->mp, nullptr, 0, st);
  {
    const size_t n = (size_t)T * B * h->ld2H;
    const DropDesc dd = mkdrop(h, RN_SITE_REC_INPUT, h->c.reconstructor_decoder_dropout, train);
    if (h->lp) hipLaunchKernelGGL(xcat_global_kernel<bf16_t>, dim3(ew_blocks(n)), dim3(256), 0, st, (const bf16_t*)h->Hs_lp, h->ldH, h->mp, (bf16_t*)h->Xcat_g, h->ld2H, T, B, H, dd);
    else hipLaunchKernelGGL(xcat_global_kernel<float>, dim3(ew_blocks(n)), dim3(256), 0, st, (const float*)h->Hs_lp, h->ldH, h->mp, (float*)h->Xcat_g, h->ld2H, T, B, H, dd);
  }
  // Xg = [h_t ; drop_t(mp)] . W_ih^T + b_ih + b_hh, batched over T (only h_r . W_hh^T is recurrent)
  gemm(h, h->Xcat_g, 0, h->ld2H, h->Wih_f, 0, h->ld2H, h->Xg, 4 * R, h->bsum_r, T * B, 4 * R, 2 * H, 1.f, 0, st);
  for (int t = 0; t < T; ++t) {
    int S = 0;
    if (t > 0 && h->fused_rec) {
      RecStepArgs a;
      a.B = B; a.R = R; a.K = R;
      a.A = (const bf16_t*)at_off(h, h->Hr_lp, (size_t)(t - 1) * B * h->ldR); a.lda = h->ldR;
      a.W = (const bf16_t*)h->Whh_g; a.ldw = h->ldR;
      a.X = h->Xg + (size_t)t * B * 4 * R; a.ldx = 4 * R;
      a.c_prev = h->Cr + (size_t)(t - 1) * B * R;
      a.h_out = h->Hr + (size_t)t * B * R; a.c_out = h->Cr + (size_t)t * B * R;
      a.acts = h->acts_r + (size_t)t * B * 4 * R;
      a.h_lp = (bf16_t*)at_off(h, h->Hr_lp, (size_t)t * B * h->ldR); a.ld_hlp = h->ldR;
      hipEvent_t e1 = nullptr;
      if (h->prof_on == RN_TAG_REC_FWD) {
        if (h->prof_used + 2 > h->prof_ev.size()) { hipEvent_t x, y; hipEventCreate(&x); hipEventCreate(&y); h->prof_ev.push_back(x); h->prof_ev.push_back(y); }
        hipEventRecord(h->prof_ev[h->prof_used++], st); e1 = h->prof_ev[h->prof_used++];
      }
      if (R <= 1536) hipLaunchKernelGGL((rec_step_fused_kernel<12, 3>), dim3(R / 8), dim3(256), 0, st, a);
      else hipLaunchKernelGGL((rec_step_fused_kernel<16, 3>), dim3(R / 8), dim3(256), 0, st, a);
      if (e1) hipEventRecord(e1, st);
      continue;
    }
    if (t > 0) S = gemm_slabs(h, RN_TAG_REC_FWD, at_off(h, h->Hr_lp, (size_t)(t - 1) * B * h->ldR), h->ldR, h->Whh_w, 0, h->ldR, B, 4 * R, R, st);
    lstm_pw(h, R, S, 4 * R, h->Xg + (size_t)t * B * 4 * R, 4 * R, nullptr, nullptr,
            t > 0 ? (h->rgru ? h->Hr : h->Cr) + (size_t)(t - 1) * B * R : nullptr, h->Hr + (size_t)t * B * R,
            at_off(h, h->Hr_lp, (size_t)t * B * h->ldR), h->ldR, nullptr, 0, h->Cr + (size_t)t * B * R,
            h->acts_r + (size_t)t * B * 4 * R, st);
  }
  // mean_t out_t = (mean_t hr_t) . W_o^T + b_o  (train.py:96-98; `out` is linear so the mean commutes)
  mean_over_t(h, h->Hr, T, R, 1.0f / (float)T, h->hrmean, h->hrmean_lp, h->ldR, st);
  gemm(h, h->hrmean_lp, 0, h->ldR, h->Wor_w, 0, h->ldR, h->outm, R, h->rP.out_bias, B, R, R, 1.f, 0, st);
  hipLaunchKernelGGL(mean_over_f_kernel, dim3(ew_blocks((size_t)B * D)), dim3(256), 0, st, enc, B, F, D, h->encmean);
  const double cnt = (double)h->c.global_batch_size * R;
  const int nb = 256;
  hipLaunchKernelGGL(mse_kernel, dim3(nb), dim3(256), 0, st, h->outm, h->encmean, 1, B, R, (size_t)R, (size_t)0,
                     (float)(2.0 / (cnt * T)), h->msep);
  hipLaunchKernelGGL(reduce_sum_kernel, dim3(1), dim3(256), 0, st, h->msep, nb, h->scal + 3, (float)(1.0 / (cnt * T)));
  return RECNET_OK;
}

static void lstm_bwd(recnet_handle* h, int Hd, int S, int slab_ld, int slab_col0, const float* dh_direct, int dhd_ld,
                     float dh_scale, const float* acts, const float* c, const float* c_prev, float* dc_carry, int first,
                     void* dG, int ld_dg, hipStream_t st, const float* slab2 = nullptr, int S2 = 0) {
  LstmBwdArgs p;
  p.B = h->B; p.Hd = Hd; p.S = S; p.gru = h->rgru; p.dh_direct = dh_direct; p.dhd_ld = dhd_ld; p.dh_scale = dh_scale;
  p.slab = h->slab; p.slab_stride = (size_t)h->B * slab_ld; p.slab_ld = slab_ld; p.slab_col0 = slab_col0;
  p.slab2 = slab2; p.slab2_stride = (size_t)h->B * Hd; p.S2 = S2;
  p.acts = acts; p.c = c; p.c_prev = c_prev; p.dc_carry = dc_carry; p.first = first; p.dG = dG; p.ld_dg = ld_dg;
  LAUNCH_AT(h, lstm_bwd_kernel, dim3(cdiv((long)h->B * Hd, 256)), dim3(256), 0, st, p);
}

static int bwd_rec_global(recnet_handle* h, float gscale, float* dhid_out, hipStream_t st) {
  const int B = h->B, H = h->H, R = h->R, T = h->T_last, TB = T * B, ld4R = h->ld4R;
  const int train = h->train_last;
  // dout (operand copy) = gscale * d loss / d out_mean
  pack_block(h, h->dout_lp, h->ldR, h->outm, R, B, R, gscale, st);
  gemm(h, h->dout_lp, 1, h->ldR, h->hrmean_lp, 1, h->ldR, h->rG.out_weight, R, nullptr, R, R, B, 1.f, 0, st);
  colsum_at(h, h->dout_lp, B, R, h->ldR, h->rG.out_bias, st, h->prezeroed);
  gemm(h, h->dout_lp, 0, h->ldR, h->Wor_w, 1, h->ldR, h->dhrmean, R, nullptr, B, R, R, 1.f, 0, st);
  int S = 0;
  for (int t = T - 1; t >= 0; --t) {
    lstm_bwd(h, R, S, R, 0, h->dhrmean, R, 1.0f / (float)T, h->acts_r + (size_t)t * B * 4 * R, h->Cr + (size_t)t * B * R,
             t > 0 ? (h->rgru ? h->Hr : h->Cr) + (size_t)(t - 1) * B * R : nullptr, h->dcr_carry, t == T - 1, at_off(h, h->dGr, (size_t)t * B * ld4R), ld4R, st);
    if (t > 0) S = gemm_slabs(h, RN_TAG_REC_BWD, at_off(h, h->dGr, (size_t)t * B * ld4R), ld4R, h->Whh_w, 1, h->ldR, B, R, 4 * R, st);
  }
  // input-side gradients, batched
  gemm(h, h->dGr, 0, ld4R, h->Wih_f, 1, h->ld2H, dhid_out, H, nullptr, TB, H, 4 * R, 1.f, 0, st);
  gemm(h, h->dGr, 0, ld4R, at_off(h, h->Wih_f, (size_t)H), 1, h->ld2H, h->dmpd, H, nullptr, TB, H, 4 * R, 1.f, 0, st);
  const size_t nBH = (size_t)B * H;
  hipLaunchKernelGGL(bcast_drop_bwd_kernel, dim3(ew_blocks(nBH)), dim3(256), 0, st, h->dmpd, h->dmp, T, B, H,
                     mkdrop(h, RN_SITE_REC_INPUT, h->c.reconstructor_decoder_dropout, train));
  hipLaunchKernelGGL(add_bcast_kernel, dim3(ew_blocks((size_t)T * nBH)), dim3(256), 0, st, dhid_out, h->dmp, T, nBH,
                     (float)h->cml / ((float)T * (float)T), 1);
  return RECNET_OK;
}
static int bwd_rec_global_deferred(recnet_handle* h, hipStream_t st) {
  const int B = h->B, H = h->H, R = h->R, T = h->T_last, TB = T * B, ld4R = h->ld4R;
  const int GR = (h->rgru ? 3 : 4) * R;
  gemm(h, h->dGr, 1, ld4R, h->Xcat_g, 1, h->ld2H, h->rG.rnn_weight_ih_l0, 2 * H, nullptr, GR, 2 * H, TB, 1.f, 0, st);   // d W_ih = dG^T . [h_t ; drop(mp)]
  if (T > 1)
    dW_hh(h, h->rgru, R, at_off(h, h->dGr, (size_t)B * ld4R), ld4R, h->Hr_lp, h->ldR, h->rG.rnn_weight_hh_l0, (T - 1) * B, 0, st);
  else
    hipMemsetAsync(h->rG.rnn_weight_hh_l0, 0, (size_t)GR * R * 4, st);
  gate_bias_grad(h, h->dGr, TB, R, ld4R, h->rG.rnn_bias_ih_l0, h->rG.rnn_bias_hh_l0, h->rgru, st);
  return RECNET_OK;
}

// ---------------------------------------------------------------------------------------------- local reconstructor
static int fwd_rec_local(recnet_handle* h, const float* enc, int T, int train, hipStream_t st) {
  const int B = h->B, F = h->F, D = h->D, H = h->H, R = h->R, RA = h->RA, ldHR = h->ldHR;
  const size_t esz = h->lp ? 2 : 4;
  if (!h->norms_hoisted) {
    param_norms(h, 1, h->scal + 4, st);
    gate_bias(h->rP.rnn_bias_ih_l0, h->rP.rnn_bias_hh_l0, h->bsum_r, R, h->rgru, st);
  }
  // Ud = hiddens . U_r^T   (local_reconstructor.py:42, hoisted)
  gemm(h, h->Hs_lp, 0, h->ldH, h->Ur_w, 0, h->ldH, h->Ud, RA, nullptr, T * B, RA, H, 1.f, 0, st);
  hipMemsetAsync(h->Xcat_r, 0, (size_t)F * B * ldHR * esz, st);   // hr_{-1} = 0 and the zero padding of every row
  LocAttnArgs a;
  a.B = B; a.T = T; a.H = H; a.A = RA; a.Ud = h->Ud; a.ab = h->rP.attn_b; a.w = h->rP.attn_w_weight; a.Hs = h->Hs;
  a.xcat_ld = ldHR; a.dd = mkdrop(h, RN_SITE_REC_INPUT, h->c.reconstructor_decoder_dropout, train);
  const size_t sm = (size_t)(RA + T + 16) * 4;
  for (int s = 0; s < F; ++s) {
    int Sa = 0;
    if (s > 0) Sa = gemm_slabs(h, RN_TAG_REC_ATT, at_off(h, h->Hr_lp, (size_t)(s - 1) * B * h->ldR), h->ldR, h->Wr_w, 0, h->ldR, B, RA, R, st);
    a.s = s; a.S = Sa; a.slab = s > 0 ? h->slab : nullptr;
    a.Whr_out = h->Whr + (size_t)s * B * RA; a.beta_out = h->beta + (size_t)s * B * T;
    a.xcat = at_off(h, h->Xcat_r, (size_t)s * B * ldHR);
    LAUNCH_AT(h, loc_attn_fwd_kernel, dim3(B, cdiv(H, 256)), dim3(256), sm, st, a);
    const int Sb = gemm_slabs(h, RN_TAG_REC_FWD, at_off(h, h->Xcat_r, (size_t)s * B * ldHR), ldHR, h->Wihh_w, 0, ldHR, B, 4 * R, H + R, st);
    lstm_pw(h, R, Sb, 4 * R, nullptr, 0, h->bsum_r, nullptr,
            s > 0 ? (h->rgru ? h->Hr : h->Cr) + (size_t)(s - 1) * B * R : nullptr, h->Hr + (size_t)s * B * R,
            at_off(h, h->Hr_lp, (size_t)s * B * h->ldR), h->ldR,
            s + 1 < F ? at_off(h, h->Xcat_r, (size_t)(s + 1) * B * ldHR + H) : nullptr, ldHR, h->Cr + (size_t)s * B * R,
            h->acts_r + (size_t)s * B * 4 * R, st);
  }
  gemm(h, h->Hr_lp, 0, h->ldR, h->Wor_w, 0, h->ldR, h->outl, R, h->rP.out_bias, F * B, R, R, 1.f, 0, st);
  const double cnt = (double)h->c.global_batch_size * F * D;
  const int nb = 512;
  hipLaunchKernelGGL(mse_kernel, dim3(nb), dim3(256), 0, st, h->outl, enc, F, B, R, (size_t)F * D, (size_t)D,
                     (float)(2.0 / cnt), h->msep);
  hipLaunchKernelGGL(reduce_sum_kernel, dim3(1), dim3(256), 0, st, h->msep, nb, h->scal + 3, (float)(1.0 / cnt));
  return RECNET_OK;
}

static int bwd_rec_local(recnet_handle* h, float gscale, float* dhid_out, hipStream_t st) {
  const int B = h->B, F = h->F, H = h->H, R = h->R, RA = h->RA, T = h->T_last, TB = T * B, FB = F * B;
  const int train = h->train_last, ld4R = h->ld4R, ldHR = h->ldHR;
  pack_block(h, h->dout_lp, h->ldR, h->outl, R, FB, R, gscale, st);
  gemm(h, h->dout_lp, 1, h->ldR, h->Hr_lp, 1, h->ldR, h->rG.out_weight, R, nullptr, R, R, FB, 1.f, 0, st);
  colsum_at(h, h->dout_lp, FB, R, h->ldR, h->rG.out_bias, st, h->prezeroed);
  gemm(h, h->dout_lp, 0, h->ldR, h->Wor_w, 1, h->ldR, h->dHr, R, nullptr, FB, R, R, 1.f, 0, st);
  LocBwdArgs a;
  a.B = B; a.T = T; a.H = H; a.R = R; a.A = RA;
  a.Hs = h->Hs; a.Ud = h->Ud; a.ab = h->rP.attn_b; a.w = h->rP.attn_w_weight;
  a.dHs = dhid_out; a.dUd = h->dUd; a.dwacc = h->dwacc_r; a.slab = h->slab;
  a.ld_dwhr = h->ldRA4; a.dUd_lp = h->dUd_lp; a.ld_dUd = h->ldRA;
  a.dd = mkdrop(h, RN_SITE_REC_INPUT, h->c.reconstructor_decoder_dropout, train);
  const int Asz = RA <= 256 ? 256 : RA;
  const size_t sm = (size_t)(H + 2 * T + 2 * Asz + 16) * 4;
  // per step: LSTM backward -> GEMM dGr_s . [W_ih | W_hh] -> attention backward -> GEMM dWhr_s . W_r
  int S1 = 0, S2 = 0;
  for (int s = F - 1; s >= 0; --s) {
    lstm_bwd(h, R, S1, H + R, H, h->dHr + (size_t)s * B * R, R, 1.0f, h->acts_r + (size_t)s * B * 4 * R, h->Cr + (size_t)s * B * R,
             s > 0 ? (h->rgru ? h->Hr : h->Cr) + (size_t)(s - 1) * B * R : nullptr, h->dcr_carry, s == F - 1, at_off(h, h->dGr, (size_t)s * B * ld4R), ld4R, st,
             h->slab2, S2);
    S1 = gemm_slabs(h, RN_TAG_REC_BWD, at_off(h, h->dGr, (size_t)s * B * ld4R), ld4R, h->Wihh_w, 1, ldHR, B, H + R, 4 * R, st);
    a.s = s; a.S = S1; a.first = (s == F - 1); a.last = (s == 0);
    a.Whr = h->Whr + (size_t)s * B * RA; a.beta = h->beta + (size_t)s * B * T;
    a.dWhr = at_off(h, h->dWhr, (size_t)s * B * h->ldRA4);
    LAUNCH_AT(h, loc_attn_bwd_kernel, dim3(B, RN_TCH), dim3(256), sm, st, a);
    if (s > 0)   // d hr_{s-1} (attention path) = (dWhr_s chunk partials) . [W_r ; .. ; W_r]
      S2 = gemm_slabs(h, RN_TAG_REC_ATT_BWD, at_off(h, h->dWhr, (size_t)s * B * h->ldRA4), h->ldRA4, h->Wr4_w, 1, h->ldR, B, R, RN_TCH * RA, st, h->slab2);
  }
  gemm(h, h->dUd_lp, 0, h->ldRA, h->Ur_w, 1, h->ldH, dhid_out, H, nullptr, TB, H, RA, 1.f, 1, st);
  return RECNET_OK;
}
static int bwd_rec_local_deferred(recnet_handle* h, hipStream_t st) {
  const int B = h->B, F = h->F, H = h->H, R = h->R, RA = h->RA, T = h->T_last, TB = T * B, FB = F * B;
  const int ld4R = h->ld4R, ldHR = h->ldHR, GR = (h->rgru ? 3 : 4) * R;
  gemm(h, h->dUd_lp, 1, h->ldRA, h->Hs_lp, 1, h->ldH, h->rG.attn_U_weight, H, nullptr, RA, H, TB, 1.f, 0, st);
  {
    const size_t n = (size_t)FB * h->ldRA;
    if (h->lp) hipLaunchKernelGGL(sum_chunks_kernel<bf16_t>, dim3(ew_blocks(n)), dim3(256), 0, st, (bf16_t*)h->dWhrs, h->ldRA, (const bf16_t*)h->dWhr, h->ldRA4, FB, RA, RN_TCH);
    else hipLaunchKernelGGL(sum_chunks_kernel<float>, dim3(ew_blocks(n)), dim3(256), 0, st, (float*)h->dWhrs, h->ldRA, (const float*)h->dWhr, h->ldRA4, FB, RA, RN_TCH);
  }
  if (F > 1) {
    gemm(h, at_off(h, h->dWhrs, (size_t)B * h->ldRA), 1, h->ldRA, h->Hr_lp, 1, h->ldR, h->rG.attn_W_weight, R, nullptr, RA, R, (F - 1) * B, 1.f, 0, st);
    dW_hh(h, h->rgru, R, at_off(h, h->dGr, (size_t)B * ld4R), ld4R, h->Hr_lp, h->ldR, h->rG.rnn_weight_hh_l0, (F - 1) * B, 0, st);
  } else {
    hipMemsetAsync(h->rG.attn_W_weight, 0, (size_t)RA * R * 4, st);
    hipMemsetAsync(h->rG.rnn_weight_hh_l0, 0, (size_t)GR * R * 4, st);
  }
  colsum_at(h, h->dWhrs, FB, RA, h->ldRA, h->rG.attn_b, st);
  colsum_t<float>(h->dwacc_r, RN_TCH * B, RA, RA, h->rG.attn_w_weight, st);
  gemm(h, h->dGr, 1, ld4R, h->Xcat_r, 1, ldHR, h->rG.rnn_weight_ih_l0, H, nullptr, GR, H, FB, 1.f, 0, st);
  gate_bias_grad(h, h->dGr, FB, R, ld4R, h->rG.rnn_bias_ih_l0, h->rG.rnn_bias_hh_l0, h->rgru, st);
  return RECNET_OK;
}

static int fwd_rec(recnet_handle* h, const float* enc, int T, int train, hipStream_t st) {
  int r = h->kind == RECNET_REC_GLOBAL ? fwd_rec_global(h, enc, T, train, st) : fwd_rec_local(h, enc, T, train, st);
  if (r) return r;
  // rec_loss = mse + lambda_reg * reg ; total = dec_loss + lambda_recon * rec_loss
  hipLaunchKernelGGL(axpb_kernel, dim3(1), dim3(1), 0, st, h->scal + 3, h->scal + 4, h->c.reconstructor_lambda_reg, h->scal + 5);
  hipLaunchKernelGGL(axpb_kernel, dim3(1), dim3(1), 0, st, h->scal + 2, h->scal + 5, h->c.lambda_recon, h->scal + 6);
  h->T_last = T; h->train_last = train; h->fwd_rec_done = 1;
  return RECNET_OK;
}
// chain part (ends with d loss / d hiddens, which the decoder backward needs) and the deferred weight gradients
static int bwd_rec_chain(recnet_handle* h, float gscale, float* dhid_out, hipStream_t st) {
  return h->kind == RECNET_REC_GLOBAL ? bwd_rec_global(h, gscale, dhid_out, st) : bwd_rec_local(h, gscale, dhid_out, st);
}
static int bwd_rec_deferred(recnet_handle* h, hipStream_t st) {
  return h->kind == RECNET_REC_GLOBAL ? bwd_rec_global_deferred(h, st) : bwd_rec_local_deferred(h, st);
}
static int bwd_rec(recnet_handle* h, float gscale, float* dhid_out, hipStream_t st) {
  int r = bwd_rec_chain(h, gscale, dhid_out, st); if (r) return r;
  return bwd_rec_deferred(h, st);
}

static int optimizer_step(recnet_handle* h, int flags, hipStream_t st, int only_group = -1) {
  // decoder: total grad norm (incl. the regulariser gradient), clip coefficient, AMSGrad step
  const int include_reg = flags & RECNET_OPT_REG;
  for (int g = 0; g < 2; ++g) {
    OptGroup& o = h->og[g];
    if (!o.bound) continue;
    if (only_group >= 0 && g != only_group) continue;
    if (g == 0 && (flags & RECNET_OPT_SKIP_DECODER)) continue;
    if (g == 1 && ((flags & RECNET_OPT_SKIP_RECONSTRUCTOR) || h->kind == RECNET_REC_NONE)) continue;
    if (!o.tab[0].m || !o.tab[0].g) return fail(RECNET_ESTATE, "gradients / Adam state not bound");
    const float lam = g == 0 ? h->c.decoder_lambda_reg : h->c.reconstructor_lambda_reg;
    const float coef = include_reg ? lam * (g == 0 ? 1.0f : h->c.lambda_recon) : 0.f;
    const float* clip = nullptr;
    if (g == 0 && (flags & RECNET_OPT_CLIP) && h->c.gradient_clip > 0.f) {
      hipLaunchKernelGGL(sumsq_chunk_kernel, dim3(o.nchunks), dim3(256), 0, st, o.d_tab, o.d_chunks, 1, o.d_pnorm, coef, o.d_partial);
      hipLaunchKernelGGL(tensor_norm_kernel, dim3(o.ntens), dim3(256), 0, st, o.d_tab, o.d_partial, o.d_gnorm);
      hipLaunchKernelGGL(norm_finalize_kernel, dim3(1), dim3(64), 0, st, o.d_gnorm, o.ntens, h->c.gradient_clip, h->scal + 7, h->scal + 8, (float*)nullptr);
      clip = h->scal + 8;
    }
    AdamHyper hp;
    hp.lr = g == 0 ? h->c.decoder_learning_rate : h->c.reconstructor_learning_rate;
    hp.wd = (float)(g == 0 ? h->c.decoder_weight_decay : h->c.reconstructor_weight_decay);
    hp.beta1 = h->c.adam_beta1; hp.beta2 = h->c.adam_beta2; hp.eps = (float)h->c.adam_eps;
    hp.one_m_b1 = (float)(1.0 - h->c.adam_beta1); hp.beta2f = (float)h->c.adam_beta2; hp.one_m_b2 = (float)(1.0 - h->c.adam_beta2);
    hp.amsgrad = g == 0 ? h->c.decoder_use_amsgrad : h->c.reconstructor_use_amsgrad;
    hp.reg_coef = coef;
    // the kernel also writes the packed operand images of the tensors it updates (no separate re-pack pass)
    hipLaunchKernelGGL(adam_chunk_kernel, dim3(o.nchunks), dim3(256), 0, st, o.d_tab, o.d_chunks, hp, o.d_pnorm, clip,
                       (const int32_t*)(h->ctrl + 1), (const PackDesc*)o.d_pack, h->lp);
    if (g == 0) refresh_wcomb_t(h, st);
  }
  return RECNET_OK;
}

// ================================================================================================
extern "C" {

int recnet_pack_weights(recnet_handle* h, void* stream) {
  REQUIRE_WS(h);
  int r = pack_weights(h, (hipStream_t)stream); if (r) return r;
  LAUNCH_OK();
  return RECNET_OK;
}

// One decode step on already prepared loop invariants (Uv, P, bias sum): embedding, input projection,
// h . [W_hh ; attn_W]^T, cell kernel, vocabulary projection.  Rows [0,B) / [B,2B) of Hs_lp are scratch.
static int dec_step_core(recnet_handle* h, const int64_t* tokens, const float* h_in, const float* c_in, float* logits,
                         float* h_out, float* c_out, int train, int t, hipStream_t st) {
  const int B = h->B, F = h->F, E = h->E, H = h->H, A = h->A, V = h->V;
  embed_fwd(h, nullptr, tokens, B, train, t, st);
  gemm(h, h->emb_lp, 0, h->ldE, h->We_w, 0, h->ldE, h->Xe, 4 * H, h->bsum_d, B, 4 * H, E, 1.f, 0, st);
  int S = 0;
  if (h_in) {   // operand copy of the incoming hidden state, then h . [W_hh ; attn_W]^T
    pack_block(h, h->Hs_lp, h->ldH, h_in, H, B, H, 1.f, st);
    S = gemm_slabs(h, RN_TAG_DEC_FWD, h->Hs_lp, h->ldH, h->Wcomb, 0, h->ldH, B, 4 * H + A, H, st);
  }
  DecCellArgs a;
  a.t = t; a.B = B; a.F = F; a.H = H; a.A = A; a.S = S; a.slab = h_in ? h->slab : nullptr;
  a.Xe = h->Xe; a.P = h->P; a.ldp = h->ld4H; a.Uv = h->Uv; a.ab = h->dP.attn_b; a.w = h->dP.attn_w_weight;
  a.gru = h->dgru; a.c_prev = h->dgru ? h_in : c_in; a.h_out = h_out; a.c_out = c_out; a.acts = nullptr; a.Wh_out = nullptr; a.att_out = nullptr;
  a.h_lp = at_off(h, h->Hs_lp, (size_t)B * h->ldH); a.ld_hlp = h->ldH;
  launch_dec_cell(h, a, st);
  gemm(h, at_off(h, h->Hs_lp, (size_t)B * h->ldH), 0, h->ldH, h->Wo_w, 0, h->ldH, logits, V, h->dP.out_bias, B, V, H, 1.f, 0, st);
  if (train && h->c.decoder_out_dropout > 0.f)
    hipLaunchKernelGGL(logits_drop_kernel, dim3(ew_blocks((size_t)B * V)), dim3(256), 0, st, logits, B, V,
                       mkdrop(h, RN_SITE_DEC_LOGIT, h->c.decoder_out_dropout, train), t);
  h->fwd_dec_done = 0;
  return RECNET_OK;
}

int recnet_decoder_prepare(recnet_handle* h, const float* enc, void* stream) {
  REQUIRE_WS(h);
  if (!h->dec_bound) return fail(RECNET_ESTATE, "decoder not bound");
  if (!enc) return fail(RECNET_EINVAL, "null argument");
  dec_invariants(h, enc, (hipStream_t)stream);
  h->fwd_dec_done = 0;
  LAUNCH_OK();
  return RECNET_OK;
}

int recnet_decoder_step(recnet_handle* h, const int64_t* tokens, const float* h_in, const float* c_in,
                        const float* enc, float* logits, float* h_out, float* c_out, int32_t train,
                        uint32_t seed, int32_t t, void* stream) {
  REQUIRE_WS(h);
  if (!h->dec_bound) return fail(RECNET_ESTATE, "decoder not bound");
  if (!tokens || !logits || !h_out || (!c_out && !h->dgru)) return fail(RECNET_EINVAL, "null argument");
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(set_u32_kernel, dim3(1), dim3(1), 0, st, h->ctrl, seed);
  if (enc) dec_invariants(h, enc, st);   // enc == NULL: reuse what recnet_decoder_prepare / the last call computed
  int r = dec_step_core(h, tokens, h_in, c_in, logits, h_out, c_out, train, t, st); if (r) return r;
  LAUNCH_OK();
  return RECNET_OK;
}

int recnet_greedy_search(recnet_handle* h, const float* enc, int64_t* tokens_out, int32_t* n_steps_out, void* stream) {
  REQUIRE_WS(h);
  if (!h->dec_bound) return fail(RECNET_ESTATE, "decoder not bound");
  if (!enc || !tokens_out || !n_steps_out) return fail(RECNET_EINVAL, "null argument");
  hipStream_t st = (hipStream_t)stream;
  const int B = h->B, H = h->H, V = h->V, Tm = h->Tm;
  dec_invariants(h, enc, st);
  hipMemsetAsync(n_steps_out, 0, 4, st);
  hipMemsetAsync(h->sr_h[0], 0, (size_t)B * H * 4, st);
  hipMemsetAsync(h->sr_c[0], 0, (size_t)B * H * 4, st);
  // <SOS> = 1 for every caption (eval.py:131)
  hipLaunchKernelGGL(fill_i64_kernel, dim3(cdiv(B, 256)), dim3(256), 0, st, h->sr_tok[0], (int64_t)1, B);
  const int64_t* tok = h->sr_tok[0];
  int cur = 0;
  for (int t = 0; t < Tm; ++t) {
    int r = dec_step_core(h, tok, h->sr_h[cur], h->sr_c[cur], h->sr_logits, h->sr_h[cur ^ 1], h->sr_c[cur ^ 1], 0, t, st);
    if (r) return r;
    int64_t* out_t = tokens_out + (size_t)t * B;
    hipLaunchKernelGGL(argmax_rows_kernel, dim3(B), dim3(256), 0, st, h->sr_logits, V, V, out_t);
    hipLaunchKernelGGL(search_stop_kernel, dim3(1), dim3(256), 0, st, out_t, B, t, n_steps_out);
    tok = out_t; cur ^= 1;
  }
  hipLaunchKernelGGL(search_finish_kernel, dim3(1), dim3(1), 0, st, n_steps_out, Tm);
  LAUNCH_OK();
  return RECNET_OK;
}

int recnet_beam_search(recnet_handle* h, const float* enc, int32_t beam_width, int64_t* best_out, int32_t* n_steps_out,
                       void* stream) {
  REQUIRE_WS(h);
  if (!h->dec_bound) return fail(RECNET_ESTATE, "decoder not bound");
  if (!enc || !best_out || !n_steps_out) return fail(RECNET_EINVAL, "null argument");
  if (beam_width < 1 || beam_width > 8) return fail(RECNET_EINVAL, "beam_width must be in [1, 8]");
  hipStream_t st = (hipStream_t)stream;
  const int B = h->B, H = h->H, V = h->V, Tm = h->Tm, bw = beam_width;
  dec_invariants(h, enc, st);
  hipMemsetAsync(n_steps_out, 0, 4, st);
  // one initial hypothesis per caption: <SOS>, zero state, log-prob 0, no <EOS> (eval.py:37-42)
  hipMemsetAsync(h->sr_h[0], 0, (size_t)B * H * 4, st);
  hipMemsetAsync(h->sr_c[0], 0, (size_t)B * H * 4, st);
  hipMemsetAsync(h->sr_cum[0], 0, (size_t)B * 4, st);
  hipMemsetAsync(h->sr_hist[0], 0, (size_t)B * Tm * 8, st);
  hipMemsetAsync(h->sr_eos[0], 0xFF, (size_t)B * 4, st);       // -1
  hipLaunchKernelGGL(fill_i64_kernel, dim3(cdiv(B, 256)), dim3(256), 0, st, h->sr_tok[0], (int64_t)1, B);
  int cur = 0, nb = 1;
  for (int t = 0; t < Tm; ++t) {
    for (int i = 0; i < nb; ++i) {
      int r = dec_step_core(h, h->sr_tok[cur] + (size_t)i * B, h->sr_h[cur] + (size_t)i * B * H, h->sr_c[cur] + (size_t)i * B * H,
                            h->sr_logits, h->sr_hn + (size_t)i * B * H, h->sr_cn + (size_t)i * B * H, 0, t, st);
      if (r) return r;
      hipLaunchKernelGGL(beam_score_kernel, dim3(B), dim3(256), 0, st, h->sr_logits, h->sr_cum[cur], h->sr_eos[cur], h->sr_scores,
                         B, V, nb, i, t);
    }
    hipLaunchKernelGGL(topk_rows_kernel, dim3(B), dim3(256), 0, st, h->sr_scores, nb * V, bw, h->sr_vals, h->sr_idx);
    BeamUpdArgs u;
    u.B = B; u.H = H; u.V = V; u.Tm = Tm; u.bw = bw; u.t = t;
    u.vals = h->sr_vals; u.idx = h->sr_idx; u.h_next = h->sr_hn; u.c_next = h->sr_cn;
    u.last_eos_old = h->sr_eos[cur]; u.hist_old = h->sr_hist[cur];
    u.h_new = h->sr_h[cur ^ 1]; u.c_new = h->sr_c[cur ^ 1]; u.cum_new = h->sr_cum[cur ^ 1];
    u.last_eos_new = h->sr_eos[cur ^ 1]; u.hist_new = h->sr_hist[cur ^ 1]; u.tok_new = h->sr_tok[cur ^ 1];
    u.n_steps = n_steps_out;
    hipLaunchKernelGGL(beam_update_kernel, dim3(bw, B), dim3(128), 0, st, u);
    hipLaunchKernelGGL(search_stop_kernel, dim3(1), dim3(256), 0, st, h->sr_tok[cur ^ 1], bw * B, t, n_steps_out);
    cur ^= 1; nb = bw;
  }
  hipLaunchKernelGGL(search_finish_kernel, dim3(1), dim3(1), 0, st, n_steps_out, Tm);
  hipLaunchKernelGGL(beam_best_kernel, dim3(cdiv(B * Tm, 256)), dim3(256), 0, st, h->sr_hist[cur], best_out, B, Tm);
  LAUNCH_OK();
  return RECNET_OK;
}

static int check_T(const recnet_handle* h, int T) { return (T >= 1 && T <= h->Tm) ? 0 : 1; }

int recnet_forward_decoder(recnet_handle* h, const float* enc, const int64_t* targets, int32_t T,
                           const float* step_weight, int32_t train, uint32_t seed, float* hiddens_out,
                           recnet_scalars* scalars, void* stream) {
  REQUIRE_WS(h);
  if (!h->dec_bound) return fail(RECNET_ESTATE, "decoder not bound");
  if (!enc || !targets || !step_weight) return fail(RECNET_EINVAL, "null argument");
  if (check_T(h, T)) return fail(RECNET_EINVAL, "T out of range");
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(set_u32_kernel, dim3(1), dim3(1), 0, st, h->ctrl, seed);
  int r = fwd_decoder(h, enc, targets, T, step_weight, train, hiddens_out, st); if (r) return r;
  if (scalars) hipLaunchKernelGGL(export_scalars_kernel, dim3(1), dim3(1), 0, st, h->scal, scalars);
  LAUNCH_OK();
  return RECNET_OK;
}

int recnet_forward_decoder_free(recnet_handle* h, const float* enc, const int64_t* targets, int32_t T,
                                const float* step_weight, int32_t train, uint32_t seed, float* hiddens_out,
                                int64_t* output_indices, recnet_scalars* scalars, void* stream) {
  REQUIRE_WS(h);
  if (!h->dec_bound) return fail(RECNET_ESTATE, "decoder not bound");
  if (!enc || !targets || !step_weight || !output_indices) return fail(RECNET_EINVAL, "null argument");
  if (check_T(h, T)) return fail(RECNET_EINVAL, "T out of range");
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(set_u32_kernel, dim3(1), dim3(1), 0, st, h->ctrl, seed);
  int r = fwd_decoder(h, enc, targets, T, step_weight, train, hiddens_out, st, output_indices); if (r) return r;
  if (scalars) hipLaunchKernelGGL(export_scalars_kernel, dim3(1), dim3(1), 0, st, h->scal, scalars);
  LAUNCH_OK();
  return RECNET_OK;
}

int recnet_forward_reconstructor(recnet_handle* h, const float* enc, const float* hiddens, int32_t T,
                                 int32_t train, uint32_t seed, recnet_scalars* scalars, void* stream) {
  REQUIRE_WS(h);
  if (h->kind == RECNET_REC_NONE || !h->rec_bound) return fail(RECNET_ESTATE, "reconstructor not bound");
  if (!enc) return fail(RECNET_EINVAL, "null argument");
  if (check_T(h, T)) return fail(RECNET_EINVAL, "T out of range");
  if (!hiddens && (!h->fwd_dec_done || h->T_last != T)) return fail(RECNET_ESTATE, "no decoder hidden states for this T");
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(set_u32_kernel, dim3(1), dim3(1), 0, st, h->ctrl, seed);
  if (hiddens) {   // hidden states handed in by the caller: refresh the fp32 image and its operand copy
    copyf(hiddens, h->Hs, (size_t)T * h->B * h->H, st);
    pack_block(h, h->Hs_lp, h->ldH, hiddens, h->H, T * h->B, h->H, 1.f, st);
  }
  int r = fwd_rec(h, enc, T, train, st); if (r) return r;
  if (scalars) hipLaunchKernelGGL(export_scalars_kernel, dim3(1), dim3(1), 0, st, h->scal, scalars);
  LAUNCH_OK();
  return RECNET_OK;
}

int recnet_backward_reconstructor(recnet_handle* h, const float* enc, float grad_scale, float* dhiddens_out, void* stream) {
  REQUIRE_WS(h);
  h->prezeroed = 0;
  if (!h->fwd_rec_done) return fail(RECNET_ESTATE, "backward_reconstructor before forward_reconstructor");
  if (!h->rG.out_weight) return fail(RECNET_ESTATE, "reconstructor gradients not bound");
  hipStream_t st = (hipStream_t)stream;
  float* dh = dhiddens_out ? dhiddens_out : h->dHsrec;
  int r = bwd_rec(h, grad_scale, dh, st);
  if (r) return r;
  if (dhiddens_out) copyf(dhiddens_out, h->dHsrec, (size_t)h->T_last * h->B * h->H, st);
  h->rec_bwd_done = 1;
  (void)enc;
  LAUNCH_OK();
  return RECNET_OK;
}

int recnet_backward_decoder(recnet_handle* h, const float* enc, const int64_t* targets, const float* dhiddens,
                            float grad_scale, void* stream) {
  REQUIRE_WS(h);
  h->prezeroed = 0;
  if (!h->fwd_dec_done) return fail(RECNET_ESTATE, "backward_decoder before forward_decoder");
  if (!h->dGd.out_weight) return fail(RECNET_ESTATE, "decoder gradients not bound");
  if (!enc || !targets) return fail(RECNET_EINVAL, "null argument");
  int r = bwd_decoder(h, enc, targets, dhiddens, grad_scale, (hipStream_t)stream); if (r) return r;
  h->fwd_dec_done = 0;   // dlogits were consumed in place
  LAUNCH_OK();
  return RECNET_OK;
}

int recnet_add_reg_grad(recnet_handle* h, int32_t which, float grad_scale, void* stream) {
  REQUIRE_WS(h);
  if (which < 0 || which > 1 || !h->og[which].bound) return fail(RECNET_EINVAL, "model not bound");
  OptGroup& o = h->og[which];
  const float lam = which == 0 ? h->c.decoder_lambda_reg : h->c.reconstructor_lambda_reg;
  hipLaunchKernelGGL(add_reg_grad_kernel, dim3(o.nchunks), dim3(256), 0, (hipStream_t)stream, o.d_tab, o.d_chunks, o.d_pnorm, lam * grad_scale);
  LAUNCH_OK();
  return RECNET_OK;
}

int recnet_clip_grad_norm(recnet_handle* h, int32_t which, float max_norm, float* total_norm_out, void* stream) {
  REQUIRE_WS(h);
  if (which < 0 || which > 1 || !h->og[which].bound || !h->og[which].tab[0].g) return fail(RECNET_EINVAL, "gradients not bound");
  OptGroup& o = h->og[which];
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(sumsq_chunk_kernel, dim3(o.nchunks), dim3(256), 0, st, o.d_tab, o.d_chunks, 1, o.d_pnorm, 0.f, o.d_partial);
  hipLaunchKernelGGL(tensor_norm_kernel, dim3(o.ntens), dim3(256), 0, st, o.d_tab, o.d_partial, o.d_gnorm);
  hipLaunchKernelGGL(norm_finalize_kernel, dim3(1), dim3(64), 0, st, o.d_gnorm, o.ntens, max_norm, h->scal + 7, h->scal + 8, (float*)nullptr);
  hipLaunchKernelGGL(scale_grads_kernel, dim3(o.nchunks), dim3(256), 0, st, o.d_tab, o.d_chunks, h->scal + 8);
  if (total_norm_out) copyf(h->scal + 7, total_norm_out, 1, st);
  LAUNCH_OK();
  return RECNET_OK;
}

int recnet_optimizer_step(recnet_handle* h, int32_t step, int32_t flags, recnet_scalars* scalars, void* stream) {
  REQUIRE_WS(h);
  if (step < 1) return fail(RECNET_EINVAL, "step must be >= 1");
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(set_u32_kernel, dim3(1), dim3(1), 0, st, h->ctrl + 1, (uint32_t)step);
  int r = optimizer_step(h, flags, st); if (r) return r;
  if (scalars) hipLaunchKernelGGL(export_scalars_kernel, dim3(1), dim3(1), 0, st, h->scal, scalars);
  LAUNCH_OK();
  return RECNET_OK;
}

// The whole forward + backward.  The four dependent chains (decoder fwd, reconstructor fwd, reconstructor bwd,
// decoder bwd) stay on `st`; work that no chain waits for runs on the side stream beside them: the vocabulary
// projection + CE + output-layer gradients while the reconstructor runs, the reconstructor's deferred weight
// gradients while the decoder BPTT runs.  The chains are latency-bound (~1 workgroup per CU), so the batched
// GEMMs fill CUs that would otherwise idle.
// phase 0: everything.  phase 1: up to and including every reconstructor gradient (so a data-parallel caller can
// start all-reducing the reconstructor bucket).  phase 2: the decoder BPTT + its deferred gradients.
// early_opt >= 0 (single-rank fused step): the reconstructor's optimiser step (flags = early_opt) is issued on the side
// stream as soon as its gradients are complete, i.e. it runs under the decoder BPTT; the caller then steps the decoder only.
static int fwd_bwd_impl(recnet_handle* h, const float* enc, const int64_t* targets, int T, const float* stepw, hipStream_t st,
                        int phase, int early_opt) {
  const bool rec = h->kind != RECNET_REC_NONE;
  hipStream_t sd = h->overlap ? h->s2 : st;
  const bool par = sd != st;
  int r;
  const float* dh = rec ? h->dHsrec : nullptr;
  h->early_opt_done = 0;
  if (phase != 2) h->free_fwd = 0;                 // the fused step is teacher-forced
  static const int f_hoist = getenv("RN_HOIST_NORMS") ? atoi(getenv("RN_HOIST_NORMS")) : 1;
  static const int f_early = getenv("RN_EARLY_OPT") ? atoi(getenv("RN_EARLY_OPT")) : 1;
  if (phase != 2) {
    // parameter norms (regulariser values; the optimiser needs them again at the end) and the gate bias of the
    // reconstructor do not depend on the batch: side stream, under the decoder chain
    hipStream_t sn = (par && f_hoist) ? sd : st;
    if (sn != st) fork_to(h, 4, st, sd);
    param_norms(h, 0, h->scal + 1, sn);
    if (rec) { param_norms(h, 1, h->scal + 4, sn); gate_bias(h->rP.rnn_bias_ih_l0, h->rP.rnn_bias_hh_l0, h->bsum_r, h->R, h->rgru, sn); }
    {   // the targets of this step's atomic column sums and of the embedding scatter-add, zeroed in one launch
      ZeroList z; z.cnt = 0;
      auto add = [&](float* p, size_t n) { if (p && n) { z.p[z.cnt] = p; z.n[z.cnt] = n; ++z.cnt; } };
      add(h->dGd.embedding_weight, (size_t)h->V * h->E); add(h->dGd.out_bias, h->V); add(h->bsum4, (size_t)4 * h->H);
      add(h->dGd.attn_b, h->A); add(h->dGd.attn_w_weight, h->A);
      if (rec) { add(h->rG.out_bias, h->R); add(h->bsum4r, (size_t)4 * h->R); }
      hipLaunchKernelGGL(zero_list_kernel, dim3(512), dim3(256), 0, sn, z);
      h->prezeroed = 1;
    }
    if (sn != st) hipEventRecord(h->ev[5], sd);       // done long before the decoder chain ends
    h->norms_hoisted = 1;
    r = dec_fwd_chain(h, enc, targets, T, 1, st);
    if (r) { h->norms_hoisted = 0; return r; }
    if (par) { fork_to(h, 0, st, sd); h->gws_cur = h->gws2; }
    r = dec_fwd_loss(h, targets, T, stepw, 1, sd);
    if (!r) r = dec_bwd_out(h, 1.0f, sd);
    h->gws_cur = h->gws;
    if (r) { h->norms_hoisted = 0; return r; }
    if (rec) {
      if (par && f_hoist) hipStreamWaitEvent(st, h->ev[5], 0);   // bsum_r and the reconstructor's norm
      r = fwd_rec(h, enc, T, 1, st);
      h->norms_hoisted = 0;
      if (r) return r;
      r = bwd_rec_chain(h, h->c.lambda_recon, h->dHsrec, st); if (r) return r;
      if (par) join_from(h, 1, st, sd);              // the decoder BPTT needs dHs_out; scal[2] is final
      hipLaunchKernelGGL(axpb_kernel, dim3(1), dim3(1), 0, st, h->scal + 2, h->scal + 5, h->c.lambda_recon, h->scal + 6);
      if (phase == 1) {                              // no decoder BPTT to hide behind: stay on the main stream
        r = bwd_rec_deferred(h, st); if (r) return r;
      }
    } else if (par) {
      join_from(h, 1, st, sd);                       // dHs_out of the vocabulary projection
    }
    h->norms_hoisted = 0;
    if (phase == 1) return RECNET_OK;
  }
  if (par) { fork_to(h, 2, st, sd); }
  if (phase == 0 && rec) {
    if (par) h->gws_cur = h->gws2;
    r = bwd_rec_deferred(h, sd); if (r) return r;
    h->gws_cur = h->gws;
    if (early_opt >= 0 && par && f_early) {
      r = optimizer_step(h, early_opt, sd, 1); if (r) return r;
      h->early_opt_done = 1;
    }
  }
  // Issuing the decoder's deferred GEMMs for finished parts of the chain while it is still running was measured and
  // is a loss (+0.22 ms): the batched GEMMs occupy the CUs the latency-bound chain kernels need at every step.
  // (So is splitting the deferred gradients over two streams after the chain, +0.11 ms: in a replayed graph every extra
  // fork / join costs more than the concurrency returns.)
  r = dec_bwd_chain(h, dh, st); if (r) return r;
  r = dec_bwd_deferred(h, enc, targets, st); if (r) return r;
  if (par) join_from(h, 3, st, sd);
  h->fwd_dec_done = 0;
  h->prezeroed = 0;
  return RECNET_OK;
}

static int fwd_bwd(recnet_handle* h, const float* enc, const int64_t* targets, int T, const float* stepw, hipStream_t st,
                   int phase = 0, int early_opt = -1) {
  const int r = fwd_bwd_impl(h, enc, targets, T, stepw, st, phase, early_opt);
  if (r) { h->prezeroed = 0; h->norms_hoisted = 0; h->gws_cur = h->gws; }
  return r;
}

int recnet_train_step_fwd_bwd(recnet_handle* h, const float* enc, const int64_t* targets, int32_t T,
                              const float* step_weight, uint32_t seed, recnet_scalars* scalars, void* stream) {
  REQUIRE_WS(h);
  if (!h->dec_bound || (h->kind != RECNET_REC_NONE && !h->rec_bound)) return fail(RECNET_ESTATE, "models not bound");
  if (!h->dGd.out_weight || (h->kind != RECNET_REC_NONE && !h->rG.out_weight)) return fail(RECNET_ESTATE, "gradients not bound");
  if (!enc || !targets || !step_weight) return fail(RECNET_EINVAL, "null argument");
  if (check_T(h, T)) return fail(RECNET_EINVAL, "T out of range");
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(set_u32_kernel, dim3(1), dim3(1), 0, st, h->ctrl, seed);
  int r = fwd_bwd(h, enc, targets, T, step_weight, st); if (r) return r;
  if (scalars) hipLaunchKernelGGL(export_scalars_kernel, dim3(1), dim3(1), 0, st, h->scal, scalars);
  h->fwd_dec_done = 0;
  LAUNCH_OK();
  return RECNET_OK;
}

static int check_step_args(recnet_handle* h, const float* enc, const int64_t* targets, int T, const float* step_weight) {
  if (!h->dec_bound || (h->kind != RECNET_REC_NONE && !h->rec_bound)) return fail(RECNET_ESTATE, "models not bound");
  if (!h->dGd.out_weight || (h->kind != RECNET_REC_NONE && !h->rG.out_weight)) return fail(RECNET_ESTATE, "gradients not bound");
  if (!enc || !targets || !step_weight) return fail(RECNET_EINVAL, "null argument");
  if (check_T(h, T)) return fail(RECNET_EINVAL, "T out of range");
  return RECNET_OK;
}

int recnet_train_step(recnet_handle* h, const float* enc, const int64_t* targets, int32_t T,
                      const float* step_weight, uint32_t seed, int32_t step, recnet_scalars* scalars, void* stream) {
  REQUIRE_WS(h);
  int r = check_step_args(h, enc, targets, T, step_weight); if (r) return r;
  if (step < 1) return fail(RECNET_EINVAL, "step must be >= 1");
  hipStream_t st = (hipStream_t)stream;
  const int flags = RECNET_OPT_REG | RECNET_OPT_CLIP;
  hipLaunchKernelGGL(set_u32_kernel, dim3(1), dim3(1), 0, st, h->ctrl, seed);
  hipLaunchKernelGGL(set_u32_kernel, dim3(1), dim3(1), 0, st, h->ctrl + 1, (uint32_t)step);
  // the reconstructor's optimiser step is issued inside, under the decoder BPTT; the decoder's here
  r = fwd_bwd(h, enc, targets, T, step_weight, st, 0, flags); if (r) return r;
  r = optimizer_step(h, flags, st, h->early_opt_done ? 0 : -1); if (r) return r;
  if (scalars) hipLaunchKernelGGL(export_scalars_kernel, dim3(1), dim3(1), 0, st, h->scal, scalars);
  h->fwd_dec_done = 0;
  LAUNCH_OK();
  return RECNET_OK;
}

int recnet_train_step_dev(recnet_handle* h, const float* enc, const int64_t* targets, int32_t T, const float* step_weight,
                          uint32_t seed_base, int32_t flags, recnet_scalars* scalars, void* stream) {
  REQUIRE_WS(h);
  int r = check_step_args(h, enc, targets, T, step_weight); if (r) return r;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(advance_step_kernel, dim3(1), dim3(1), 0, st, (int32_t*)(h->ctrl + 1), h->ctrl, seed_base);
  r = fwd_bwd(h, enc, targets, T, step_weight, st, 0, flags); if (r) return r;
  r = optimizer_step(h, flags, st, h->early_opt_done ? 0 : -1); if (r) return r;
  if (scalars) hipLaunchKernelGGL(export_scalars_kernel, dim3(1), dim3(1), 0, st, h->scal, scalars);
  h->fwd_dec_done = 0;
  LAUNCH_OK();
  return RECNET_OK;
}

int recnet_set_step(recnet_handle* h, int32_t step, void* stream) {
  REQUIRE_WS(h);
  hipLaunchKernelGGL(set_u32_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, h->ctrl + 1, (uint32_t)step);
  LAUNCH_OK();
  return RECNET_OK;
}

int recnet_train_step_fwd_bwd_dev(recnet_handle* h, const float* enc, const int64_t* targets, int32_t T,
                                  const float* step_weight, uint32_t seed_base, recnet_scalars* scalars, void* stream) {
  REQUIRE_WS(h);
  if (!h->dec_bound || (h->kind != RECNET_REC_NONE && !h->rec_bound)) return fail(RECNET_ESTATE, "models not bound");
  if (!h->dGd.out_weight || (h->kind != RECNET_REC_NONE && !h->rG.out_weight)) return fail(RECNET_ESTATE, "gradients not bound");
  if (!enc || !targets || !step_weight) return fail(RECNET_EINVAL, "null argument");
  if (check_T(h, T)) return fail(RECNET_EINVAL, "T out of range");
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(advance_step_kernel, dim3(1), dim3(1), 0, st, (int32_t*)(h->ctrl + 1), h->ctrl, seed_base);
  int r = fwd_bwd(h, enc, targets, T, step_weight, st); if (r) return r;
  if (scalars) hipLaunchKernelGGL(export_scalars_kernel, dim3(1), dim3(1), 0, st, h->scal, scalars);
  h->fwd_dec_done = 0;
  LAUNCH_OK();
  return RECNET_OK;
}

int recnet_train_step_part_dev(recnet_handle* h, int32_t part, const float* enc, const int64_t* targets, int32_t T,
                               const float* step_weight, uint32_t seed_base, recnet_scalars* scalars, void* stream) {
  REQUIRE_WS(h);
  if (part != 1 && part != 2) return fail(RECNET_EINVAL, "part must be 1 or 2");
  if (!h->dec_bound || (h->kind != RECNET_REC_NONE && !h->rec_bound)) return fail(RECNET_ESTATE, "models not bound");
  if (!h->dGd.out_weight || (h->kind != RECNET_REC_NONE && !h->rG.out_weight)) return fail(RECNET_ESTATE, "gradients not bound");
  if (!enc || !targets || !step_weight) return fail(RECNET_EINVAL, "null argument");
  if (check_T(h, T)) return fail(RECNET_EINVAL, "T out of range");
  hipStream_t st = (hipStream_t)stream;
  if (part == 1) hipLaunchKernelGGL(advance_step_kernel, dim3(1), dim3(1), 0, st, (int32_t*)(h->ctrl + 1), h->ctrl, seed_base);
  else if (h->T_last != T) return fail(RECNET_ESTATE, "part 2 without a matching part 1");
  int r = fwd_bwd(h, enc, targets, T, step_weight, st, part); if (r) return r;
  if (scalars) hipLaunchKernelGGL(export_scalars_kernel, dim3(1), dim3(1), 0, st, h->scal, scalars);
  LAUNCH_OK();
  return RECNET_OK;
}

int recnet_optimizer_step_dev(recnet_handle* h, int32_t flags, recnet_scalars* scalars, void* stream) {
  REQUIRE_WS(h);
  hipStream_t st = (hipStream_t)stream;
  int r = optimizer_step(h, flags, st); if (r) return r;
  if (scalars) hipLaunchKernelGGL(export_scalars_kernel, dim3(1), dim3(1), 0, st, h->scal, scalars);
  LAUNCH_OK();
  return RECNET_OK;
}

int recnet_profile_begin(recnet_handle* h, int32_t site) {
  if (!h || site < 1 || site > 5) return fail(RECNET_EINVAL, "bad profile site");
  h->prof_on = site; h->prof_used = 0;
  return RECNET_OK;
}

int recnet_profile_null_launch(recnet_handle* h, int32_t count, void* stream) {
  if (!h || !h->ws) return fail(RECNET_ESTATE, "workspace not bound");
  if (!h->prof_on) return fail(RECNET_ESTATE, "not profiling");
  hipStream_t st = (hipStream_t)stream;
  if (h->prof_used + 2 > h->prof_ev.size()) {
    hipEvent_t a, b; HIPCHK(hipEventCreate(&a)); HIPCHK(hipEventCreate(&b));
    h->prof_ev.push_back(a); h->prof_ev.push_back(b);
  }
  hipEvent_t e0 = h->prof_ev[h->prof_used++], e1 = h->prof_ev[h->prof_used++];
  HIPCHK(hipEventRecord(e0, st));
  for (int i = 0; i < count; ++i) hipLaunchKernelGGL(set_u32_kernel, dim3(1), dim3(1), 0, st, h->ctrl + 8, 0u);
  HIPCHK(hipEventRecord(e1, st));
  LAUNCH_OK();
  return RECNET_OK;
}

int recnet_profile_read(recnet_handle* h, int32_t* n_launches, double* total_ms) {
  if (!h || !n_launches || !total_ms) return fail(RECNET_EINVAL, "null argument");
  double tot = 0; int n = 0;
  for (size_t i = 0; i + 1 < h->prof_used; i += 2) {
    HIPCHK(hipEventSynchronize(h->prof_ev[i + 1]));
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, h->prof_ev[i], h->prof_ev[i + 1]));
    tot += ms; ++n;
  }
  *n_launches = n; *total_ms = tot;
  return RECNET_OK;
}

int recnet_profile_end(recnet_handle* h, int32_t* n_launches, double* total_ms) {
  if (!h || !n_launches || !total_ms) return fail(RECNET_EINVAL, "null argument");
  h->prof_on = 0;
  double tot = 0; int n = 0;
  for (size_t i = 0; i + 1 < h->prof_used; i += 2) {
    HIPCHK(hipEventSynchronize(h->prof_ev[i + 1]));
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, h->prof_ev[i], h->prof_ev[i + 1]));
    tot += ms; ++n;
  }
  *n_launches = n; *total_ms = tot;
  h->prof_used = 0;
  return RECNET_OK;
}

int recnet_gemm(int32_t precision, const float* A, int32_t a_col, int32_t lda, const float* B, int32_t b_col,
                int32_t ldb, float* C, int32_t ldc, const float* bias, int32_t M, int32_t N, int32_t K,
                float alpha, int32_t accumulate, int32_t splitk, float* splitk_ws, void* stream) {
  if (!A || !B || !C || M <= 0 || N <= 0 || K <= 0) return fail(RECNET_EINVAL, "bad gemm argument");
  if (splitk > 1 && !splitk_ws) return fail(RECNET_EINVAL, "split-K needs a workspace");
  rn_launch_gemm(precision, A, 0, a_col, lda, B, 0, b_col, ldb, C, ldc, bias, M, N, K, alpha, accumulate, splitk, splitk_ws, 1,
                 (hipStream_t)stream);
  LAUNCH_OK();
  return RECNET_OK;
}

int recnet_gemm_bf16(const void* A, int32_t a_col, int32_t lda, const void* B, int32_t b_col, int32_t ldb, float* C,
                     int32_t ldc, const float* bias, int32_t M, int32_t N, int32_t K, float alpha, int32_t accumulate,
                     int32_t splitk, float* splitk_ws, int32_t tag, void* stream) {
  if (!A || !B || !C || M <= 0 || N <= 0 || K <= 0) return fail(RECNET_EINVAL, "bad gemm argument");
  if (splitk > 1 && !splitk_ws) return fail(RECNET_EINVAL, "split-K needs a workspace");
  if (tag < 0 || tag > 5) return fail(RECNET_EINVAL, "bad tag");
  rn_launch_gemm(RN_PREC_BF16, A, 1, a_col, lda, B, 1, b_col, ldb, C, ldc, bias, M, N, K, alpha, accumulate, splitk, splitk_ws, 1,
                 (hipStream_t)stream, tag);
  LAUNCH_OK();
  return RECNET_OK;
}

double recnet_recurrent_step_bytes(const recnet_handle* h, int32_t which) {
  if (!h) return 0;
  const double wb = h->prec == RN_PREC_BF16 ? 2.0 : 4.0;
  // weights streamed once + activation block read + fp32 partial results written
  if (which == 0) return (double)(4 * h->H + h->A) * h->H * wb + (double)h->B * h->H * wb + (double)h->B * (4 * h->H + h->A) * 4;
  if (h->kind == RECNET_REC_GLOBAL) return (double)4 * h->R * h->R * wb + (double)h->B * h->R * wb + (double)h->B * 4 * h->R * 4;
  if (h->kind == RECNET_REC_LOCAL) return (double)4 * h->R * (h->H + h->R) * wb + (double)h->B * (h->H + h->R) * wb + (double)h->B * 4 * h->R * 4;
  return 0;
}

}  // extern "C"
