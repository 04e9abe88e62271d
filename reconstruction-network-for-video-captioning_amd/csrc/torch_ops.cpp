// torch.ops.recnet.* — PyTorch-ROCm custom-op registration of the RecNet hot path (BASELINE
// north_star: "called from Python via PyTorch-ROCm custom ops").  A thin layer over the C ABI of librecnet_hip.so
// (include/recnet_hip.h): every op validates its tensors (TORCH_CHECK -> RuntimeError: dtype, shape, contiguity,
// residency), allocates its outputs through torch's caching allocator, takes the current HIP stream and makes exactly
// one C-ABI call.  No arithmetic happens here.  The engine handle (recnet_create) travels as an int.
//
// Paired forward / backward ops (autograd is wired in api.py with torch.autograd.Function over these ops):
//   forward_decoder / backward_decoder                  train.py:17-75 and its BPTT
//   forward_decoder_free                                train.py:46-51 (validation pass)
//   forward_reconstructor / backward_reconstructor      train.py:78-131 and its BPTT
//   train_step_fwd_bwd, train_step, optimizer_step      train.py:248-273
//   decoder_step, reconstructor_step                    Decoder.forward / *Reconstructor.forward (per-step API)
//   greedy_search, beam_search                          eval.py:19-120
#include <ATen/ATen.h>
#include <c10/hip/HIPStream.h>
#include <torch/library.h>

#include "../../include/recnet_hip.h"

namespace {

recnet_handle* H(int64_t h) {
  TORCH_CHECK(h != 0, "recnet: null engine handle");
  return reinterpret_cast<recnet_handle*>(static_cast<intptr_t>(h));
}
void* stream() { return (void*)c10::hip::getCurrentHIPStream().stream(); }
void ok(int rc, const char* what) { TORCH_CHECK(rc == 0, "recnet::", what, " failed (code ", rc, "): ", recnet_last_error()); }
void chk(const at::Tensor& t, at::ScalarType ty, const char* name) {
  TORCH_CHECK(t.is_cuda(), "recnet: ", name, " must be a CUDA (HIP) tensor — there is no CPU fallback");
  TORCH_CHECK(t.scalar_type() == ty, "recnet: ", name, " has dtype ", t.scalar_type(), ", expected ", ty);
  TORCH_CHECK(t.is_contiguous(), "recnet: ", name, " must be contiguous");
}
const float* fptr(const c10::optional<at::Tensor>& t, const char* name) {
  if (!t.has_value() || !t->defined()) return nullptr;
  chk(*t, at::kFloat, name);
  return t->data_ptr<float>();
}
at::Tensor scalars_like(const at::Tensor& ref) { return at::zeros({8}, ref.options().dtype(at::kFloat)); }
int64_t dim(int64_t h, int which) { return recnet_dim(H(h), which); }
// Shapes come from the handle (recnet_dim), never from the caller's tensors: a wrong-shaped tensor is a RuntimeError,
// not an out-of-bounds device write.
void shape(const at::Tensor& t, std::initializer_list<int64_t> want, const char* name) {
  int64_t n = 1;
  for (auto w : want) n *= w;
  TORCH_CHECK(t.numel() == n, "recnet: ", name, " has ", t.numel(), " elements (sizes ", t.sizes(), "), the engine expects ",
              at::IntArrayRef(want.begin(), want.size()));
}
void chk_enc(int64_t h, const at::Tensor& enc) {
  chk(enc, at::kFloat, "encoder_outputs");
  TORCH_CHECK(enc.dim() == 3 && enc.size(0) == dim(h, RECNET_DIM_B) && enc.size(1) == dim(h, RECNET_DIM_F) && enc.size(2) == dim(h, RECNET_DIM_D),
              "recnet: encoder_outputs has sizes ", enc.sizes(), ", the engine expects [", dim(h, RECNET_DIM_B), ", ", dim(h, RECNET_DIM_F), ", ",
              dim(h, RECNET_DIM_D), "]");
}
void chk_targets(int64_t h, const at::Tensor& targets, int64_t T) {
  chk(targets, at::kLong, "targets");
  TORCH_CHECK(targets.dim() == 2 && targets.size(1) == dim(h, RECNET_DIM_B) && targets.size(0) >= T && T >= 1 && T <= dim(h, RECNET_DIM_TM),
              "recnet: targets has sizes ", targets.sizes(), ", the engine expects [>= T = ", T, " (<= ", dim(h, RECNET_DIM_TM), "), ",
              dim(h, RECNET_DIM_B), "]");
}
void chk_T(int64_t h, int64_t T) { TORCH_CHECK(T >= 1 && T <= dim(h, RECNET_DIM_TM), "recnet: T = ", T, " outside [1, ", dim(h, RECNET_DIM_TM), "]"); }

// ---------------------------------------------------------------- sequence level
std::tuple<at::Tensor, at::Tensor, at::Tensor> forward_decoder(int64_t h, const at::Tensor& enc, const at::Tensor& targets, int64_t T,
                                                               const at::Tensor& step_weight, bool train, int64_t seed) {
  chk_enc(h, enc); chk_targets(h, targets, T); chk(step_weight, at::kFloat, "step_weight");
  TORCH_CHECK(step_weight.numel() == T, "recnet: step_weight must have T entries");
  auto sc = scalars_like(enc);
  auto hid = at::empty({T, 1, enc.size(0), recnet_dim(H(h), RECNET_DIM_H)}, enc.options());
  ok(recnet_forward_decoder(H(h), enc.data_ptr<float>(), targets.data_ptr<int64_t>(), (int32_t)T, step_weight.data_ptr<float>(),
                            train, (uint32_t)seed, hid.data_ptr<float>(), (recnet_scalars*)sc.data_ptr<float>(), stream()), "forward_decoder");
  return {sc.select(0, 2).clone(), hid, sc};
}
std::tuple<at::Tensor, at::Tensor, at::Tensor, at::Tensor> forward_decoder_free(int64_t h, const at::Tensor& enc, const at::Tensor& targets,
                                                                                int64_t T, const at::Tensor& step_weight, bool train,
                                                                                int64_t seed) {
  chk_enc(h, enc); chk_targets(h, targets, T); chk(step_weight, at::kFloat, "step_weight");
  TORCH_CHECK(step_weight.numel() == T, "recnet: step_weight must have T entries");
  auto sc = scalars_like(enc);
  auto hid = at::empty({T, 1, enc.size(0), recnet_dim(H(h), RECNET_DIM_H)}, enc.options());
  auto idx = at::empty({T, enc.size(0)}, targets.options());
  ok(recnet_forward_decoder_free(H(h), enc.data_ptr<float>(), targets.data_ptr<int64_t>(), (int32_t)T, step_weight.data_ptr<float>(),
                                 train, (uint32_t)seed, hid.data_ptr<float>(), idx.data_ptr<int64_t>(),
                                 (recnet_scalars*)sc.data_ptr<float>(), stream()), "forward_decoder_free");
  return {sc.select(0, 2).clone(), hid, idx, sc};
}
void backward_decoder(int64_t h, const at::Tensor& enc, const at::Tensor& targets, const c10::optional<at::Tensor>& dhiddens,
                      double grad_scale) {
  chk_enc(h, enc); chk(targets, at::kLong, "targets");
  TORCH_CHECK(targets.dim() == 2 && targets.size(1) == dim(h, RECNET_DIM_B), "recnet: targets must be [Tm, B]");
  if (dhiddens.has_value() && dhiddens->defined())
    TORCH_CHECK(dhiddens->numel() % (dim(h, RECNET_DIM_B) * dim(h, RECNET_DIM_H)) == 0 && dhiddens->numel() > 0 &&
                    dhiddens->numel() / (dim(h, RECNET_DIM_B) * dim(h, RECNET_DIM_H)) <= dim(h, RECNET_DIM_TM),
                "recnet: dhiddens must be [T,1,B,H] of the forward pass, got ", dhiddens->sizes());
  ok(recnet_backward_decoder(H(h), enc.data_ptr<float>(), targets.data_ptr<int64_t>(), fptr(dhiddens, "dhiddens"), (float)grad_scale,
                             stream()), "backward_decoder");
}
std::tuple<at::Tensor, at::Tensor> forward_reconstructor(int64_t h, const at::Tensor& enc, const c10::optional<at::Tensor>& hiddens,
                                                         int64_t T, bool train, int64_t seed) {
  chk_enc(h, enc); chk_T(h, T);
  if (hiddens.has_value() && hiddens->defined()) shape(*hiddens, {T, 1, dim(h, RECNET_DIM_B), dim(h, RECNET_DIM_H)}, "decoder_hiddens");
  auto sc = scalars_like(enc);
  ok(recnet_forward_reconstructor(H(h), enc.data_ptr<float>(), fptr(hiddens, "decoder_hiddens"), (int32_t)T, train, (uint32_t)seed,
                                  (recnet_scalars*)sc.data_ptr<float>(), stream()), "forward_reconstructor");
  return {sc.select(0, 5).clone(), sc};
}
at::Tensor backward_reconstructor(int64_t h, const at::Tensor& enc, int64_t T, double grad_scale) {
  chk_enc(h, enc); chk_T(h, T);
  auto dh = at::empty({T, 1, dim(h, RECNET_DIM_B), dim(h, RECNET_DIM_H)}, enc.options());
  ok(recnet_backward_reconstructor(H(h), enc.data_ptr<float>(), (float)grad_scale, dh.data_ptr<float>(), stream()), "backward_reconstructor");
  return dh;
}
void add_reg_grad(int64_t h, int64_t which, double grad_scale) { ok(recnet_add_reg_grad(H(h), (int32_t)which, (float)grad_scale, stream()), "add_reg_grad"); }

// ---------------------------------------------------------------- fused step (train.py:248-273)
at::Tensor train_step_fwd_bwd(int64_t h, const at::Tensor& enc, const at::Tensor& targets, int64_t T, const at::Tensor& step_weight,
                              int64_t seed) {
  chk_enc(h, enc); chk_targets(h, targets, T); chk(step_weight, at::kFloat, "step_weight");
  TORCH_CHECK(step_weight.numel() == T, "recnet: step_weight must have T entries");
  auto sc = scalars_like(enc);
  ok(recnet_train_step_fwd_bwd(H(h), enc.data_ptr<float>(), targets.data_ptr<int64_t>(), (int32_t)T, step_weight.data_ptr<float>(),
                               (uint32_t)seed, (recnet_scalars*)sc.data_ptr<float>(), stream()), "train_step_fwd_bwd");
  return sc;
}
at::Tensor train_step(int64_t h, const at::Tensor& enc, const at::Tensor& targets, int64_t T, const at::Tensor& step_weight, int64_t seed,
                      int64_t step) {
  chk_enc(h, enc); chk_targets(h, targets, T); chk(step_weight, at::kFloat, "step_weight");
  TORCH_CHECK(step_weight.numel() == T, "recnet: step_weight must have T entries");
  auto sc = scalars_like(enc);
  ok(recnet_train_step(H(h), enc.data_ptr<float>(), targets.data_ptr<int64_t>(), (int32_t)T, step_weight.data_ptr<float>(), (uint32_t)seed,
                       (int32_t)step, (recnet_scalars*)sc.data_ptr<float>(), stream()), "train_step");
  return sc;
}
void optimizer_step(int64_t h, int64_t step, int64_t flags) { ok(recnet_optimizer_step(H(h), (int32_t)step, (int32_t)flags, nullptr, stream()), "optimizer_step"); }
at::Tensor clip_grad_norm(int64_t h, int64_t which, double max_norm, const at::Tensor& like) {
  auto out = at::empty({1}, like.options().dtype(at::kFloat));
  ok(recnet_clip_grad_norm(H(h), (int32_t)which, (float)max_norm, out.data_ptr<float>(), stream()), "clip_grad_norm");
  return out;
}

// ---------------------------------------------------------------- per-step API
std::tuple<at::Tensor, at::Tensor, at::Tensor> decoder_step(int64_t h, const at::Tensor& tokens, const c10::optional<at::Tensor>& h_in,
                                                            const c10::optional<at::Tensor>& c_in, const c10::optional<at::Tensor>& enc,
                                                            bool train, int64_t seed, int64_t t) {
  chk(tokens, at::kLong, "input");
  const int64_t B = dim(h, RECNET_DIM_B), Hd = dim(h, RECNET_DIM_H);
  shape(tokens, {1, B}, "input");
  if (h_in.has_value() && h_in->defined()) shape(*h_in, {B, Hd}, "hidden h");
  if (c_in.has_value() && c_in->defined()) shape(*c_in, {B, Hd}, "hidden c");
  if (enc.has_value() && enc->defined()) chk_enc(h, *enc);
  auto opt = tokens.options().dtype(at::kFloat);
  auto logits = at::empty({B, recnet_dim(H(h), RECNET_DIM_V)}, opt);
  auto ho = at::empty({B, recnet_dim(H(h), RECNET_DIM_H)}, opt), co = at::empty_like(ho);
  ok(recnet_decoder_step(H(h), tokens.data_ptr<int64_t>(), fptr(h_in, "hidden h"), fptr(c_in, "hidden c"), fptr(enc, "encoder_outputs"),
                         logits.data_ptr<float>(), ho.data_ptr<float>(), co.data_ptr<float>(), train, (uint32_t)seed, (int32_t)t, stream()),
     "decoder_step");
  return {logits, ho, co};
}
std::tuple<at::Tensor, at::Tensor, at::Tensor> reconstructor_step(int64_t h, const c10::optional<at::Tensor>& input, const at::Tensor& hr_in,
                                                                  const c10::optional<at::Tensor>& cr_in,
                                                                  const c10::optional<at::Tensor>& decoder_hiddens, int64_t T, bool train,
                                                                  int64_t seed, int64_t t) {
  chk(hr_in, at::kFloat, "hidden hr");
  const int64_t B = dim(h, RECNET_DIM_B), R = dim(h, RECNET_DIM_R), Hd = dim(h, RECNET_DIM_H);
  chk_T(h, T);
  shape(hr_in, {B, R}, "hidden hr");
  if (cr_in.has_value() && cr_in->defined()) shape(*cr_in, {B, R}, "hidden cr");
  if (input.has_value() && input->defined()) shape(*input, {B, Hd}, "input");
  if (decoder_hiddens.has_value() && decoder_hiddens->defined()) shape(*decoder_hiddens, {T, 1, B, Hd}, "decoder_hiddens");
  auto out = at::empty({B, R}, hr_in.options()), ho = at::empty({B, R}, hr_in.options()), co = at::empty({B, R}, hr_in.options());
  ok(recnet_reconstructor_step(H(h), fptr(input, "input"), hr_in.data_ptr<float>(), fptr(cr_in, "hidden cr"),
                               fptr(decoder_hiddens, "decoder_hiddens"), (int32_t)T, out.data_ptr<float>(), ho.data_ptr<float>(),
                               co.data_ptr<float>(), train, (uint32_t)seed, (int32_t)t, stream()), "reconstructor_step");
  return {out, ho, co};
}

// ---------------------------------------------------------------- search (eval.py:19-120)
std::tuple<at::Tensor, at::Tensor> greedy_search(int64_t h, const at::Tensor& enc) {
  chk_enc(h, enc);
  auto toks = at::zeros({recnet_dim(H(h), RECNET_DIM_TM), enc.size(0)}, enc.options().dtype(at::kLong));
  auto n = at::zeros({1}, enc.options().dtype(at::kInt));
  ok(recnet_greedy_search(H(h), enc.data_ptr<float>(), toks.data_ptr<int64_t>(), n.data_ptr<int32_t>(), stream()), "greedy_search");
  return {toks, n};
}
std::tuple<at::Tensor, at::Tensor> beam_search(int64_t h, const at::Tensor& enc, int64_t beam_width) {
  chk_enc(h, enc);
  auto best = at::zeros({recnet_dim(H(h), RECNET_DIM_TM), enc.size(0)}, enc.options().dtype(at::kLong));
  auto n = at::zeros({1}, enc.options().dtype(at::kInt));
  ok(recnet_beam_search(H(h), enc.data_ptr<float>(), (int32_t)beam_width, best.data_ptr<int64_t>(), n.data_ptr<int32_t>(), stream()), "beam_search");
  return {best, n};
}

}  // namespace

TORCH_LIBRARY(recnet, m) {
  m.def("forward_decoder(int handle, Tensor encoder_outputs, Tensor targets, int T, Tensor step_weight, bool train, int seed) -> (Tensor loss, Tensor hiddens, Tensor scalars)");
  m.def("forward_decoder_free(int handle, Tensor encoder_outputs, Tensor targets, int T, Tensor step_weight, bool train, int seed) -> (Tensor loss, Tensor hiddens, Tensor output_indices, Tensor scalars)");
  m.def("backward_decoder(int handle, Tensor encoder_outputs, Tensor targets, Tensor? dhiddens, float grad_scale) -> ()");
  m.def("forward_reconstructor(int handle, Tensor encoder_outputs, Tensor? decoder_hiddens, int T, bool train, int seed) -> (Tensor loss, Tensor scalars)");
  m.def("backward_reconstructor(int handle, Tensor encoder_outputs, int T, float grad_scale) -> Tensor dhiddens");
  m.def("add_reg_grad(int handle, int which, float grad_scale) -> ()");
  m.def("train_step_fwd_bwd(int handle, Tensor encoder_outputs, Tensor targets, int T, Tensor step_weight, int seed) -> Tensor scalars");
  m.def("train_step(int handle, Tensor encoder_outputs, Tensor targets, int T, Tensor step_weight, int seed, int step) -> Tensor scalars");
  m.def("optimizer_step(int handle, int step, int flags) -> ()");
  m.def("clip_grad_norm(int handle, int which, float max_norm, Tensor like) -> Tensor total_norm");
  m.def("decoder_step(int handle, Tensor input, Tensor? h, Tensor? c, Tensor? encoder_outputs, bool train, int seed, int t) -> (Tensor logits, Tensor h, Tensor c)");
  m.def("reconstructor_step(int handle, Tensor? input, Tensor hr, Tensor? cr, Tensor? decoder_hiddens, int T, bool train, int seed, int t) -> (Tensor output, Tensor hr, Tensor cr)");
  m.def("greedy_search(int handle, Tensor encoder_outputs) -> (Tensor tokens, Tensor n_steps)");
  m.def("beam_search(int handle, Tensor encoder_outputs, int beam_width) -> (Tensor tokens, Tensor n_steps)");
}

// The handle is an int, so dispatch cannot key on a tensor for every op: ops with tensor arguments are registered for the
// CUDA (= HIP on ROCm) key — a CPU tensor then fails with "no kernel for CPU backend", loudly — and the tensor-free ones
// as CompositeExplicitAutograd.
TORCH_LIBRARY_IMPL(recnet, CUDA, m) {
  m.impl("forward_decoder", forward_decoder);
  m.impl("forward_decoder_free", forward_decoder_free);
  m.impl("backward_decoder", backward_decoder);
  m.impl("forward_reconstructor", forward_reconstructor);
  m.impl("backward_reconstructor", backward_reconstructor);
  m.impl("train_step_fwd_bwd", train_step_fwd_bwd);
  m.impl("train_step", train_step);
  m.impl("clip_grad_norm", clip_grad_norm);
  m.impl("decoder_step", decoder_step);
  m.impl("reconstructor_step", reconstructor_step);
  m.impl("greedy_search", greedy_search);
  m.impl("beam_search", beam_search);
}
TORCH_LIBRARY_IMPL(recnet, CompositeExplicitAutograd, m) {
  m.impl("add_reg_grad", add_reg_grad);
  m.impl("optimizer_step", optimizer_step);
}
