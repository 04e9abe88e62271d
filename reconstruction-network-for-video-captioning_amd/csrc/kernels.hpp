// Hand-written gfx950 kernels of the RecNet train step other than the MFMA GEMM (gemm.hpp):
// embedding gather, the fused per-caption recurrent-step kernels (LSTM gate pointwise + Bahdanau
// attention with the caption's encoder states streamed once per step), masked cross-entropy with
// logits dropout, reconstruction MSE, their backward mirrors, norms and the multi-tensor Adam.
// All arithmetic here is fp32; only GEMM operands are ever rounded to bf16.
#pragma once
#include "common.hpp"
#include "launch.hpp"

#include "kernels_util.hpp"
#include "kernels_decoder.hpp"
#include "kernels_reconstructor.hpp"
#include "kernels_search.hpp"
#include "kernels_optim.hpp"
