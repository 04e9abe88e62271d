/* recnet_hip.h — C ABI of the MI355X (gfx950) RecNet train-step library (librecnet_hip.so).
 *
 * The reference (hobincar/reconstruction-network-for-video-captioning) has no FFI layer: its hot
 * path sits behind Python callables that dispatch to stock PyTorch kernels.  This header is the
 * boundary a maintainer would bind instead (ctypes stub: INTEGRATION.md).  Every entry point names
 * the reference interface it replaces.  All pointers are raw DEVICE pointers unless a parameter is
 * documented "host"; `stream` is a hipStream_t passed as void*; every function only enqueues work
 * on that stream (no allocation, no synchronisation => hipGraph-capturable) and returns 0 on
 * success or a negative RECNET_E* code (recnet_last_error() gives the message).
 *
 * Tensors are fp32, row-major, with the reference's shapes and state_dict names;
 * token ids are int64 like the reference's LongTensors.
 */
#ifndef RECNET_HIP_H
#define RECNET_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RECNET_ABI_VERSION 7
#define RECNET_ATTN_NONE 0
#define RECNET_ATTN_SOFTMAX 1
#define RECNET_OK 0
#define RECNET_EINVAL (-1)      /* bad dimension / null pointer / unsupported variant */
#define RECNET_ESTATE (-2)      /* call order violated (e.g. backward before forward) */
#define RECNET_EHIP (-3)        /* a HIP runtime call failed */

#define RECNET_REC_NONE 0
#define RECNET_REC_GLOBAL 1     /* models/global_reconstructor.py */
#define RECNET_REC_LOCAL 2      /* models/local_reconstructor.py  */
#define RECNET_CELL_LSTM 0      /* torch.nn.LSTM, gate rows (i, f, g, o) */
#define RECNET_CELL_GRU 1       /* torch.nn.GRU, gate rows (r, z, n); the hidden state is a single tensor (train.py:33-35) */
#define RECNET_PREC_F32 0       /* exact fp32 MFMA (v_mfma_f32_16x16x4_f32) */
#define RECNET_PREC_BF16 1      /* bf16 MFMA operands, fp32 accumulate / state / losses */

/* Shapes + hyper-parameters.  Field names follow config.py:27-93 (TrainConfig). */
typedef struct recnet_config {
  int32_t batch_size;                 /* B: captions on THIS rank                       config.py:51  */
  int32_t encoder_output_len;         /* F                                              config.py:63  */
  int32_t encoder_output_size;        /* D                                              config.py:62  */
  int32_t embedding_size;             /* E                                              config.py:57  */
  int32_t decoder_hidden_size;        /* H                                              config.py:67  */
  int32_t decoder_attn_size;          /* A                                              config.py:68  */
  int32_t n_vocabs;                   /* V                                              train.py:222  */
  int32_t reconstructor_hidden_size;  /* R (local reconstructor requires R == D)        config.py:78  */
  int32_t reconstructor_attn_size;    /* local reconstructor only                       config.py:82  */
  int32_t caption_max_len;            /* decoder runs <= caption_max_len + 1 steps      config.py:50  */
  int32_t reconstructor_type;         /* RECNET_REC_*                                   config.py:76  */
  int32_t precision;                  /* RECNET_PREC_*                                                */
  int32_t global_batch_size;          /* data parallel: B summed over ranks (== B single GPU)         */
  int32_t batch_offset;               /* data parallel: global index of this rank's first caption     */
  int32_t decoder_use_amsgrad, reconstructor_use_amsgrad;       /* config.py:90-91 */
  int32_t decoder_cell;               /* RECNET_CELL_*: decoder_model                    config.py:31  */
  int32_t reconstructor_cell;         /* RECNET_CELL_*: reconstructor_model              config.py:77  */
  int32_t decoder_attn_normalize;     /* RECNET_ATTN_NONE (the reference: decoder.py:30's softmax is constructed but never
                                         called, decoder.py:55-61) | RECNET_ATTN_SOFTMAX: softmax over the frames of the
                                         attention energies before the weighted mean (opt-in; north_star's wording) */
  float embedding_scale;              /*                                                config.py:59  */
  float embedding_dropout;            /*                                                config.py:58  */
  float decoder_out_dropout;          /* dropout applied to the logits, decoder.py:69   config.py:70  */
  float reconstructor_decoder_dropout;/*                                                config.py:79  */
  float gradient_clip;                /* <= 0 disables clipping                         config.py:92-93 */
  float decoder_lambda_reg;           /* train.py:151 (1e-3) */
  float reconstructor_lambda_reg;     /* train.py:188 (1e-2) */
  float lambda_recon;                 /* train.py:225 (1.0)  */
  /* optimiser hyper-parameters are doubles, like the Python floats torch.optim.Adam receives */
  double decoder_learning_rate, reconstructor_learning_rate;    /* config.py:86-87 */
  double decoder_weight_decay, reconstructor_weight_decay;      /* config.py:88-89 */
  double adam_beta1, adam_beta2, adam_eps;                      /* torch.optim.Adam defaults 0.9 / 0.999 / 1e-8 */
} recnet_config;

/* Decoder parameter set — state_dict keys of models/decoder.py:22-42 (LSTM, 1 layer). */
typedef struct recnet_decoder_tensors {
  float* attn_b;           /* [A]          */
  float* embedding_weight; /* [V, E]       */
  float* attn_W_weight;    /* [A, H]       */
  float* attn_U_weight;    /* [A, D]       */
  float* attn_w_weight;    /* [1, A]       */
  float* rnn_weight_ih_l0; /* [4H, E + D]  */
  float* rnn_weight_hh_l0; /* [4H, H]      */
  float* rnn_bias_ih_l0;   /* [4H]         */
  float* rnn_bias_hh_l0;   /* [4H]         */
  float* out_weight;       /* [V, H]       */
  float* out_bias;         /* [V]          */
} recnet_decoder_tensors;

/* Reconstructor parameter set — models/global_reconstructor.py:17-28 / local_reconstructor.py:17-35.
 * attn_* are NULL for the global reconstructor. */
typedef struct recnet_reconstructor_tensors {
  float* attn_b;           /* [RA]                              (local) */
  float* attn_W_weight;    /* [RA, R]                           (local) */
  float* attn_U_weight;    /* [RA, H]                           (local) */
  float* attn_w_weight;    /* [1, RA]                           (local) */
  float* rnn_weight_ih_l0; /* [4R, 2H] global | [4R, H] local           */
  float* rnn_weight_hh_l0; /* [4R, R]   */
  float* rnn_bias_ih_l0;   /* [4R]      */
  float* rnn_bias_hh_l0;   /* [4R]      */
  float* out_weight;       /* [R, R]    */
  float* out_bias;         /* [R]       */
} recnet_reconstructor_tensors;

/* Scalars produced on the device by the forward / optimiser calls (one float each). */
typedef struct recnet_scalars {
  float dec_ce;        /* sum_t mean_b CE / sum_t n_t            train.py:56-68  */
  float dec_reg;       /* sum_p ||p||_2 over decoder tensors     train.py:69     */
  float dec_loss;      /* dec_ce + lambda_reg * dec_reg          train.py:70     */
  float rec_mse;       /* MSE term (global: already / T)         train.py:101-102 / :128 */
  float rec_reg;       /*                                        train.py:103 / :129 */
  float rec_loss;      /*                                        train.py:104 / :130 */
  float total_loss;    /* dec_loss + lambda_recon * rec_loss     train.py:260    */
  float dec_grad_norm; /* total decoder grad norm before clipping, clip_grad_norm_ train.py:270 */
} recnet_scalars;

typedef struct recnet_handle recnet_handle;

int recnet_abi_version(void);
const char* recnet_last_error(void);

/* Host-only object describing one (config) problem; owns no device memory. */
int recnet_create(const recnet_config* cfg, recnet_handle** out);
void recnet_destroy(recnet_handle* h);
/* Update the data-parallel placement / learning rates without re-creating (host fields only). */
int recnet_set_shard(recnet_handle* h, int32_t global_batch_size, int32_t batch_offset);

/* Bytes of device scratch the caller must provide (activations saved for backward, split-K slabs,
 * packed weights).  Must stay alive and untouched between a forward and its backward. */
size_t recnet_workspace_bytes(const recnet_handle* h);
int recnet_bind_workspace(recnet_handle* h, void* workspace, size_t bytes);

/* Parameters, gradients and Adam state.  `grad` tensors are overwritten (not accumulated) by the
 * backward calls — the reference zero_grad()s before every backward (train.py:265-267).
 * exp_avg / exp_avg_sq / max_exp_avg_sq mirror torch.optim.Adam's per-parameter state; max_* may be
 * NULL structs' pointers when amsgrad is off. */
int recnet_bind_decoder(recnet_handle* h, const recnet_decoder_tensors* param, const recnet_decoder_tensors* grad,
                        const recnet_decoder_tensors* exp_avg, const recnet_decoder_tensors* exp_avg_sq,
                        const recnet_decoder_tensors* max_exp_avg_sq);
int recnet_bind_reconstructor(recnet_handle* h, const recnet_reconstructor_tensors* param,
                              const recnet_reconstructor_tensors* grad,
                              const recnet_reconstructor_tensors* exp_avg,
                              const recnet_reconstructor_tensors* exp_avg_sq,
                              const recnet_reconstructor_tensors* max_exp_avg_sq);

/* Re-derive the packed (concatenated / bf16) weight images after the fp32 parameters changed
 * (optimiser step, load_state_dict).  recnet_optimizer_step does this itself. */
int recnet_pack_weights(recnet_handle* h, void* stream);

/* ---- Decoder.forward, models/decoder.py:45-70: ONE decode step (the API eval.py's greedy / beam
 * search drive, eval.py:22,48).  tokens [B] int64; h_in/c_in/h_out/c_out [B,H]; enc [B,F,D];
 * logits [B,V].  `train` != 0 applies the embedding / logits dropout with (seed, t).  enc == NULL reuses the invariants
 * of the previous call / of recnet_decoder_prepare. */
int recnet_decoder_step(recnet_handle* h, const int64_t* tokens, const float* h_in, const float* c_in,
                        const float* enc, float* logits, float* h_out, float* c_out, int32_t train,
                        uint32_t seed, int32_t t, void* stream);

/* Compute the loop invariants of `enc` once (Uv = enc . U^T, P = enc . W_ih[:, E:]^T); later recnet_decoder_step calls
 * with enc == NULL reuse them (the search loops of eval.py call the step 31 x beam times on the same features). */
int recnet_decoder_prepare(recnet_handle* h, const float* enc, void* stream);

/* ---- greedy_search, eval.py:19-33, as one device-side loop (no per-sample Python, no host sync): start from <SOS> and
 * zero state (eval.py:131-141), argmax feedback.  tokens_out [caption_max_len+1][B] int64 (time-major);
 * n_steps_out (device int32) = number of steps the reference's loop produces (it stops after the first step whose
 * tokens are all <PAD>, eval.py:30): rows >= n_steps are to be ignored. */
int recnet_greedy_search(recnet_handle* h, const float* enc, int64_t* tokens_out, int32_t* n_steps_out, void* stream);
/* ---- beam_search, eval.py:36-120: log(sigmoid(logit)) scores (eval.py:61), cumulative score divided by
 * length^0.7 at every step (eval.py:53-59, length = position of the hypothesis' last <EOS>, else t+1), top beam_width of
 * beam_width * V continuations (eval.py:64).  best_out [caption_max_len+1][B] = the top-1 hypothesis (eval.py:119),
 * n_steps_out as above (eval.py:116).  beam_width <= 8. */
int recnet_beam_search(recnet_handle* h, const float* enc, int32_t beam_width, int64_t* best_out, int32_t* n_steps_out,
                       void* stream);

/* ---- forward_decoder, train.py:17-75 (teacher forcing, train.py:38,45).
 * enc [B,F,D]; targets [caption_max_len+1, B] int64 (time-major, <PAD>=0, <EOS>=2);
 * T = number of steps the reference's loop would run (train.py:66), computed by the caller from the
 * caption lengths (host) — no device sync; step_weight [T] = 1 / (n_t * sum_t n_t) with GLOBAL counts
 * (SURVEY.md §8e); hiddens_out [T,1,B,H] (may be NULL: they stay in the workspace for the
 * reconstructor either way); scalars: dec_ce / dec_reg / dec_loss are written. */
int recnet_forward_decoder(recnet_handle* h, const float* enc, const int64_t* targets, int32_t T,
                           const float* step_weight, int32_t train, uint32_t seed, float* hiddens_out,
                           recnet_scalars* scalars, void* stream);

/* ---- forward_decoder without teacher forcing, train.py:46-51 — the validation pass (train.py:327 calls
 * forward_decoder with the default teacher_forcing_ratio = 0): the input of step t+1 is the arg-max of the logits
 * Decoder.forward returned at step t.  Same arguments and loss as recnet_forward_decoder (targets / masks only enter the
 * loss and T); output_indices [T,B] int64 receives the fed-back tokens (train.py:50, `output_indices`).  A following
 * recnet_backward_decoder differentiates this pass (the arg-max passes no gradient; the embedding gradient is scattered to
 * the rows of the tokens that were fed); the hidden states feed recnet_forward_reconstructor as usual. */
int recnet_forward_decoder_free(recnet_handle* h, const float* enc, const int64_t* targets, int32_t T,
                                const float* step_weight, int32_t train, uint32_t seed, float* hiddens_out,
                                int64_t* output_indices, recnet_scalars* scalars, void* stream);

/* ---- forward_global_reconstructor (train.py:78-105) / forward_local_reconstructor (train.py:108-131),
 * selected by cfg.reconstructor_type.  Consumes the hidden states left by recnet_forward_decoder
 * (or `hiddens` [T,1,B,H] when not NULL).  mse_count = GLOBAL element count of the MSE mean
 * (global: B_global*R, local: B_global*F*D).  Writes rec_mse / rec_reg / rec_loss / total_loss. */
int recnet_forward_reconstructor(recnet_handle* h, const float* enc, const float* hiddens, int32_t T,
                                 int32_t train, uint32_t seed, recnet_scalars* scalars, void* stream);

/* ---- loss.backward(), train.py:268, for loss = dec_loss + lambda_recon * rec_loss (train.py:260).
 * Reconstructor first (produces d loss / d hiddens), then the decoder BPTT.  grad_scale multiplies
 * the incoming gradient (1.0 in the reference).  The norm-regulariser gradient lambda * p/||p|| is
 * NOT included here (it is rank-independent; recnet_optimizer_step / recnet_add_reg_grad add it
 * once, after the data-parallel all-reduce — SURVEY.md §8e). */
int recnet_backward_reconstructor(recnet_handle* h, const float* enc, float grad_scale, float* dhiddens_out,
                                  void* stream);
int recnet_backward_decoder(recnet_handle* h, const float* enc, const int64_t* targets, const float* dhiddens,
                            float grad_scale, void* stream);
/* grad += lambda_reg * grad_scale * p / ||p||  for every tensor of the model (autograd-compatible path). */
int recnet_add_reg_grad(recnet_handle* h, int32_t which /*0 decoder, 1 reconstructor*/, float grad_scale,
                        void* stream);

/* ---- train.py:269-273: clip_grad_norm_(decoder, gradient_clip) + decoder Adam(amsgrad) step +
 * reconstructor Adam step (torch.optim.Adam semantics, coupled weight decay); then re-packs the
 * weights.  `step` is the 1-based optimiser step count.  flags: */
#define RECNET_OPT_REG 1                 /* fold the regulariser gradient lambda * p/||p|| in        */
#define RECNET_OPT_CLIP 2                /* clip the decoder gradient to cfg.gradient_clip first      */
#define RECNET_OPT_SKIP_DECODER 4
#define RECNET_OPT_SKIP_RECONSTRUCTOR 8
int recnet_optimizer_step(recnet_handle* h, int32_t step, int32_t flags, recnet_scalars* scalars, void* stream);
/* torch.nn.utils.clip_grad_norm_ (train.py:270) on the bound gradients of one model, in place;
 * total_norm_out: device float or NULL. */
int recnet_clip_grad_norm(recnet_handle* h, int32_t which, float max_norm, float* total_norm_out, void* stream);

/* ---- The whole train-step body, train.py:248-273, on one stream: forward decoder, forward
 * reconstructor, backward, clip, both optimiser steps.  With world size > 1 the caller instead runs
 * recnet_train_step_fwd_bwd, all-reduces (SUM) the gradient tensors, then recnet_optimizer_step. */
int recnet_train_step_fwd_bwd(recnet_handle* h, const float* enc, const int64_t* targets, int32_t T,
                              const float* step_weight, uint32_t seed, recnet_scalars* scalars, void* stream);
int recnet_train_step(recnet_handle* h, const float* enc, const int64_t* targets, int32_t T,
                      const float* step_weight, uint32_t seed, int32_t step, recnet_scalars* scalars,
                      void* stream);

/* The fused step for graph replay (single rank): recnet_train_step with the step count and the dropout seed taken from
 * device memory (advanced by the call itself, seed = seed_base + step) — forward, backward, then the optimiser with
 * `flags`.  The reconstructor's optimiser step is issued as soon as its gradients are complete, under the decoder's
 * backward chain; the decoder's follows the clip (train.py:265-273 as one capturable sequence). */
int recnet_train_step_dev(recnet_handle* h, const float* enc, const int64_t* targets, int32_t T, const float* step_weight,
                          uint32_t seed_base, int32_t flags, recnet_scalars* scalars, void* stream);

/* hipGraph-replay-friendly forms: the optimiser step count and the dropout seed live in device memory.
 * recnet_set_step initialises the counter; *_fwd_bwd_dev first does step += 1, seed = seed_base + step
 * on the device, so replaying a captured graph advances both; *_optimizer_step_dev reads the counter. */
int recnet_set_step(recnet_handle* h, int32_t step, void* stream);
int recnet_train_step_fwd_bwd_dev(recnet_handle* h, const float* enc, const int64_t* targets, int32_t T,
                                  const float* step_weight, uint32_t seed_base, recnet_scalars* scalars, void* stream);
/* The same forward + backward in two launches, so that a data-parallel caller can all-reduce the reconstructor's
 * gradient bucket while the decoder's backward is still running: part 1 = step += 1, decoder forward, loss,
 * reconstructor forward + complete backward (all reconstructor gradients final, d loss / d hiddens kept);
 * part 2 = decoder BPTT + deferred decoder gradients. */
int recnet_train_step_part_dev(recnet_handle* h, int32_t part, const float* enc, const int64_t* targets, int32_t T,
                               const float* step_weight, uint32_t seed_base, recnet_scalars* scalars, void* stream);
int recnet_optimizer_step_dev(recnet_handle* h, int32_t flags, recnet_scalars* scalars, void* stream);
/* Deferred reconstructor update (opt-in; the fused single-rank step recnet_train_step[_dev] only).  train.py:272-273 steps
 * the two optimisers one after the other at the end of an iteration; nothing reads the reconstructor's parameters again
 * before the NEXT iteration's reconstructor forward (train.py:256).  With on != 0 a fused step therefore leaves the
 * reconstructor's weight-gradient products + Adam update PENDING (its gate gradients and saved activations stay in the
 * workspace), and the next fused step runs them on a third stream under its decoder forward chain — CUs that chain leaves
 * idle — instead of under / behind this step's decoder BPTT.  Same arithmetic in the same order: parameters after
 * recnet_flush are bit-identical to the non-deferred step's.  recnet_flush completes a pending update on `stream` (a no-op
 * on the device when nothing is pending); every other entry point of the handle that reads the reconstructor's parameters,
 * gradients or Adam state flushes first, callers that read them through their own pointers (state_dict, checkpoints)
 * call recnet_flush themselves.
 * on == 2 (global reconstructor): SPLIT update — only the recurrent weights (rnn.weight_hh_l0: 60 % of the reconstructor's
 * weight-gradient work) are left pending; the input-side weights, the output layer and the biases are updated inside the step,
 * beside the decoder's BPTT chain.  The pending product then fits under the next step's decoder forward chain (137 workgroups,
 * 119 CUs idle) without delaying that step's reconstructor, and this step's BPTT window is not overfilled. */
int recnet_set_deferred_reconstructor_update(recnet_handle* h, int32_t on, void* stream);
int recnet_flush(recnet_handle* h, void* stream);
/* A hipGraph that captured a fused step was replayed (replays run no host code): tells the handle that — with the deferred
 * update on — an update may be pending, so that its other entry points catch up before they touch the reconstructor. */
int recnet_mark_pending(recnet_handle* h);

/* Data-parallel step inside ONE stream-ordered sequence (or one captured graph): with on != 0, part 1 of
 * recnet_train_step_part_dev returns with the reconstructor's weight-gradient products still running on the library's side
 * stream; recnet_join_side makes `stream` — the stream the caller launches the all-reduce of the reconstructor bucket from — wait
 * for them, while the caller's main stream goes straight on to part 2 (the decoder's BPTT).  Without it part 1 waits for those
 * products itself (+0.2 ms in front of the BPTT at the benchmark shape).  train.py:264-273; SURVEY.md section 8e. */
int recnet_set_dp_overlap(recnet_handle* h, int32_t on);
int recnet_join_side(recnet_handle* h, void* stream);
/* A stream capture that contained calls of this library was abandoned (e.g. api.GraphedStep: a collective could not be captured,
 * train.py:264-273 under data parallelism): the enqueue-time bookkeeping of the step those calls were part of — open side
 * branches, recorded-but-unjoined fork events, the data-parallel overlap switch — is put back to "between two steps".  No device
 * work, no effect on parameters, gradients or a pending deferred update. */
int recnet_abort_step(recnet_handle* h);

/* ---- plumbing exposed for tests and profiling */
/* Between begin and end every recurrent-step GEMM launch of one site (the dependent-chain kernels: one
 * per decoder / reconstructor time step) is bracketed by hipEvents on its stream; end() synchronises and
 * returns the launch count and summed duration.  Not for use under graph capture. */
#define RECNET_SITE_DEC_FWD 1   /* gates_t   = [ctx_t, h_{t-1}] . [W_ih[:,E:] | W_hh]^T       */
#define RECNET_SITE_DEC_BWD 2   /* d[ctx, h] = dgates_t . [W_ih[:,E:] | W_hh]                   */
#define RECNET_SITE_REC_FWD 3   /* reconstructor gates (global: hr . W_hh^T; local: [x, hr] . [W_ih | W_hh]^T) */
#define RECNET_SITE_REC_BWD 4
#define RECNET_SITE_REC_ATT 5   /* local reconstructor: hr . attn_W^T                           */
#define RECNET_SITE_REC_ATT_BWD 6
#define RECNET_SITE_REC_CHAIN_FWD 7   /* global reconstructor, bf16: the whole forward chain as one persistent launch */
#define RECNET_SITE_REC_CHAIN_BWD 8   /* ... and the whole backward chain                                              */
#define RECNET_SITE_DEC_CHAIN_FWD 9   /* decoder, bf16, teacher-forced: the whole forward chain as one persistent launch */
#define RECNET_SITE_DEC_CHAIN_BWD 10  /* ... and the whole BPTT chain                                                    */
int recnet_profile_begin(recnet_handle* h, int32_t site);
int recnet_profile_end(recnet_handle* h, int32_t* n_launches, double* total_ms);
/* Same read-out without leaving profiling mode.  Brackets are taken around EAGER launches only: on a capturing stream the
 * launches of the site are not bracketed (round 4: event-record nodes read back after every replay aborted inside the HIP
 * runtime in about 1 of 100 bench runs; a replayed graph is described by recnet_read_stamps instead). */
int recnet_profile_read(recnet_handle* h, int32_t* n_launches, double* total_ms);
/* Phase stamps of the LAST train step, written by the step's own kernels (no tracer, no extra launch, also in a replayed
 * hipGraph): 14 values of the device's 100 MHz wall clock (10 ns units) —
 *   [0] step start (the step-counter kernel of recnet_train_step[_fwd_bwd]_dev)
 *   [1 + 2k], [2 + 2k] workgroup 0 of chain kernel k started running / left, k = 0 decoder forward chain, 1 decoder BPTT chain,
 *                      2 global reconstructor forward, 3 global reconstructor backward, 4 local reconstructor forward, 5 local backward
 *                      (the last launch of that chain; a chain that did not run keeps its old stamps)
 *   [13] step end (the kernel that exports the step's scalars)
 *   n >= 30: [14 + 2j], [15 + 2j] first workgroup started / last workgroup left of the grouped GEMM launch of site j:
 *            0 decoder prologue (Xe, Uv, P), 1 reconstructor weight gradients inside the step, 2 the pending half of a split
 *            reconstructor update (mode 2), 3 decoder weight gradients.
 * Synchronises `stream`.  train.py:248-273 is the span between [0] and [13]. */
int recnet_read_stamps(recnet_handle* h, uint64_t* out, int32_t n, void* stream);
/* Calibration of the bracket itself: `count` launches of an empty kernel bracketed by the same two event records (call
 * between recnet_profile_begin and _read/_end, in the same eager / captured mode as the measurement).  bracket(count) =
 * E + count * f: E is what the two event records add to every bracketed duration, f the dispatch-to-completion time of
 * an empty kernel; a bracketed kernel's own dispatch-to-completion time is its bracket minus E. */
int recnet_profile_null_launch(recnet_handle* h, int32_t count, void* stream);
/* C[M,N] (+)= alpha * op(A) op(B)^T + bias.  a_col / b_col: operand stored with the contraction index
 * as the ROW index (see csrc/gemm.hpp).  All fp32 device pointers. */
int recnet_gemm(int32_t precision, const float* A, int32_t a_col, int32_t lda, const float* B, int32_t b_col,
                int32_t ldb, float* C, int32_t ldc, const float* bias, int32_t M, int32_t N, int32_t K,
                float alpha, int32_t accumulate, int32_t splitk, float* splitk_ws, void* stream);
/* Same with bf16 operands in memory (the production kernel of the bf16 path, csrc/gemm_lds.hpp): needs 16-byte
 * aligned bases and leading dimensions that are multiples of 8.  tag 0 = batched form, 1..5 = chain-site forms. */
int recnet_gemm_bf16(const void* A, int32_t a_col, int32_t lda, const void* B, int32_t b_col, int32_t ldb, float* C,
                     int32_t ldc, const float* bias, int32_t M, int32_t N, int32_t K, float alpha, int32_t accumulate,
                     int32_t splitk, float* splitk_ws, int32_t tag, void* stream);
/* Grouped form: the tiles of n <= 8 products of one operand layout in ONE launch (csrc/gemm_lds.hpp: gemm_group_kernel); a
 * product the scheduler splits along K is summed inside the launch by its last-arriving slice.  Arrays of n entries; splitk_ws:
 * fp32 slabs (may be null: no splitting), counters: zero-initialised words the launch leaves zeroed (null: no splitting).
 * The batched products of train.py:258-273 (the decoder's and reconstructor's weight gradients) are issued this way. */
int recnet_gemm_group_bf16(int32_t a_col, int32_t b_col, int32_t n, const void* const* A, const int32_t* lda, const void* const* B,
                           const int32_t* ldb, float* const* C, const int32_t* ldc, const float* const* bias, const int32_t* M,
                           const int32_t* N, const int32_t* K, const float* alpha, const int32_t* accumulate, float* splitk_ws,
                           int64_t ws_floats, uint32_t* counters, int32_t n_counters, void* stream);
/* Probe builds only (csrc: make probe): in-kernel wall-clock stamps of the local chain kernels' last launch,
 * [role][step][8] uint64 ticks; RECNET_ESTATE in the product build. */
int recnet_probe_read(recnet_handle* h, uint64_t* out, int32_t n);
/* Test hook: fills the LDS of every CU with NaN patterns, so that a kernel reading LDS it never wrote fails the parity
 * tests deterministically. */
int recnet_debug_poison_lds(recnet_handle* h, void* stream);
/* Dimensions a handle was created with (the torch.ops layer sizes its outputs from these). */
#define RECNET_DIM_B 0
#define RECNET_DIM_F 1
#define RECNET_DIM_D 2
#define RECNET_DIM_E 3
#define RECNET_DIM_H 4
#define RECNET_DIM_A 5
#define RECNET_DIM_V 6
#define RECNET_DIM_R 7
#define RECNET_DIM_RA 8
#define RECNET_DIM_TM 9      /* caption_max_len + 1: rows of `targets` */
#define RECNET_DIM_SPLIT_FITS 10   /* 1: mode 2 of recnet_set_deferred_reconstructor_update is applied at this shape (the pending
                                    * product fits beside the decoder's forward chain); 0: the step updates immediately */
int32_t recnet_dim(const recnet_handle* h, int32_t which);
/* One reconstructor step with the reference's per-step semantics — GlobalReconstructor.forward(input, hidden,
 * decoder_hiddens) (models/global_reconstructor.py:30-46, called at train.py:94) and LocalReconstructor.forward(hidden,
 * decoder_hiddens) (models/local_reconstructor.py:37-55, called at train.py:123).  input [B][H] (global only: the layer-0
 * slice of decoder_hiddens[t]); hr_in / cr_in [B][R] or NULL for the zero state (GRU: cr_* unused); decoder_hiddens
 * [T][B][H] fp32, or NULL to reuse the loop invariants of the previous call; out [B][R] = out(h'), hr_out / cr_out
 * [B][R].  t indexes the dropout mask of this call (the reference draws a fresh mask per call).  Forward only. */
int recnet_reconstructor_step(recnet_handle* h, const float* input, const float* hr_in, const float* cr_in,
                              const float* decoder_hiddens, int32_t T, float* out, float* hr_out, float* cr_out,
                              int32_t train, uint32_t seed, int32_t t, void* stream);
/* Health of the persistent chain kernels (bounded waits): status bits 1/2 reconstructor fwd/bwd chain, 4/8 decoder
 * fwd/BPTT chain, 32/64 local reconstructor fwd/bwd chain gave up a wait; 256 the step's loss was poisoned (NaN) and the
 * optimiser kernels skip their updates.  Synchronises `stream`.  recnet_chain_reset clears the sticky words;
 * disable_persistent != 0 switches the handle to the per-step kernels for every later call. */
/* Gradient transport of the data-parallel step in its direct form (one reduce-scatter message per peer, i.e. per xGMI link, then an
 * all-gather: SURVEY.md section 8e; train.py:264-273 under data parallelism).  The collectives themselves are RCCL's; these are
 * the staging kernels around them, on `stream`, no handle needed:
 *   recnet_dp_cast    dst[i] = (dst type) src[i], n elements; *_bf16 = 0: fp32, 1: bf16 (fp32 -> wire type before the reduce-scatter,
 *                     wire type -> the fp32 gradient buffer after the all-gather)
 *   recnet_dp_reduce  red[j] = (wire type) sum_{r = 0 .. world-1, in rank order} (float) recv[r * chunk + j]: fp32 accumulation at the
 *                     destination, one rounding of the sum (world <= 16) */
int recnet_dp_cast(const void* src, int32_t src_bf16, void* dst, int32_t dst_bf16, int64_t n, void* stream);
int recnet_dp_reduce(const void* recv, int32_t world, int64_t chunk, void* red, int32_t wire_bf16, void* stream);
/* Measurement hook: the last seven step-start stamps (out16[1..7], written by the step's first kernel; out16[0] = how many steps ever
 * started) and step-end stamps (out16[9..15] / out16[8]) of recnet_train_step_dev, 100 MHz wall clock, slot of step n = 1 + (n - 1) % 7
 * counting from the handle's first step: the idle time between back-to-back replays of a captured step, read directly.
 * Synchronises the stream. */
int recnet_read_step_ring(recnet_handle* h, uint64_t* out16, void* stream);
int recnet_chain_status(recnet_handle* h, int32_t* status_out, void* stream);
int recnet_chain_reset(recnet_handle* h, int32_t disable_persistent, void* stream);
/* Test hook: what a chain kernel does when it gives up a bounded wait (rec_chain.hpp: rc_give_up) — raises the sticky word
 * of chain `chain_bit` (one of 1, 2, 4, 8, 32, 64) and the poison word, stream-ordered. */
int recnet_debug_raise_give_up(recnet_handle* h, int32_t chain_bit, void* stream);
/* Test hook: a kernel of `n_workgroups` workgroups that each take a whole CU (160 KB of LDS) and spin for `microseconds` —
 * what a resident collective (RCCL) kernel looks like to the persistent chain kernels.  Launch it on another stream. */
int recnet_debug_occupy(recnet_handle* h, int32_t n_workgroups, int32_t microseconds, void* stream);
/* Test hook: the packed operand images of the weights (bf16 / fp32 copies, transposes, streamed fragments — what the kernels
 * read instead of the master parameters; rewritten by the optimiser kernels, by the Adam epilogue of the pending d W_hh product
 * and by recnet_pack_weights) are copied aside, re-packed from the master parameters, and compared: *n_diff_out = number of
 * 16-bit words that differ (0 = every image was up to date).  Completes a pending deferred update first; synchronises the
 * stream; leaves the images freshly packed.  Holds train.py:271-273 "the next forward uses the updated weights" for the images. */
int64_t recnet_debug_images_bytes(recnet_handle* h);       /* bytes of device scratch recnet_debug_images_stale needs (256-byte aligned) */
int recnet_debug_images_stale(recnet_handle* h, void* scratch_dev, int64_t scratch_bytes, int64_t* n_diff_out, void* stream);
/* Test hook: byte offset inside the bound workspace of a saved tensor of the local reconstructor's forward pass
 * (0: Whr [F][B][RA], 1: beta [F][B][T], 2: Hr [F][B][R], 3: acts [F][B][4R]); -1 if unknown. */
int64_t recnet_debug_offset(const recnet_handle* h, int32_t which);
/* Name / start / duration of the dominant kernel's launches inside the last train step are measured
 * by the caller with hipEvents; this returns the ALGORITHMIC bytes of one launch: loop invariants once, every input /
 * saved tensor once.  The step-to-step exchange blocks of a persistent chain kernel are not part of it. */
double recnet_recurrent_step_bytes(const recnet_handle* h, int32_t which /*0 decoder fwd step, 1 reconstructor fwd step,
    2 reconstructor bwd step, 3 / 4 one launch of the persistent reconstructor fwd / bwd chain over the last T,
    5 / 6 one launch of the persistent decoder fwd / BPTT chain*/);
/* Bytes of the exchange blocks (h_t / dgates_t panels, stamped words) one launch of chain kernel `which` (3..6) writes
 * and reads back: implementation traffic, reported beside the algorithmic bytes. */
double recnet_chain_exchange_bytes(const recnet_handle* h, int32_t which);

#ifdef __cplusplus
}
#endif
#endif /* RECNET_HIP_H */
