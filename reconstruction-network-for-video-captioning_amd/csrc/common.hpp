// Common device helpers for the RecNet gfx950 kernels (wave64, MFMA, LDS).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __bf16 bf16_t;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4_t __attribute__((ext_vector_type(4)));

#define RN_WAVE 64

// ---------------------------------------------------------------- dropout (counter-based)
// keep(seed, site, t, b, j): see oracle/dropmask.py (the numpy restatement used by the tests).
#define RN_SITE_DEC_EMBED 0u
#define RN_SITE_DEC_LOGIT 1u
#define RN_SITE_REC_INPUT 2u

__host__ __device__ __forceinline__ uint32_t rn_fmix32(uint32_t x) {
  x ^= x >> 16; x *= 0x85EBCA6Bu; x ^= x >> 13; x *= 0xC2B2AE35u; x ^= x >> 16;
  return x;
}
__host__ __device__ __forceinline__ uint32_t rn_site_key(uint32_t seed, uint32_t site) {
  return seed * 0x632BE5ABu + site * 0x7F4A7C15u + 0x1234567u;
}
// returns the multiplicative mask: 0 (dropped) or 1/(1-p) (kept).  thr = floor(p*2^32), thr==0 -> no dropout.
__device__ __forceinline__ float rn_drop_scale(uint32_t key, uint32_t thr, float inv_keep, uint32_t idx) {
  if (thr == 0u) return 1.0f;
  uint32_t h = rn_fmix32(idx * 0x9E3779B1u + key);
  return h >= thr ? inv_keep : 0.0f;
}
__host__ __forceinline__ uint32_t rn_drop_thr(float p) {
  if (p <= 0.f) return 0u;
  double v = (double)p * 4294967296.0;
  return v >= 4294967295.0 ? 0xFFFFFFFFu : (uint32_t)v;
}

// ---------------------------------------------------------------- reductions
// Wave-wide reductions (all 64 lanes active; every lane gets the result).  Four DPP steps (quad swaps, half-row and row
// mirrors: ~10 clocks each) leave the sum of each 16-lane row in all of its lanes, four v_readlane fetch the row sums.
// (The __shfl_xor butterfly costs six dependent ds_bpermute round trips, ~700 clocks — measured: seven of them in a row
// were 2.5 us of a 16 us decoder step.)
template <int CTRL> __device__ __forceinline__ float rn_dpp(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float rn_readlane(float v, int l) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l));
}
__device__ __forceinline__ float wave_sum(float v) {
  v += rn_dpp<0xB1>(v);     // quad_perm [1,0,3,2]
  v += rn_dpp<0x4E>(v);     // quad_perm [2,3,0,1]
  v += rn_dpp<0x141>(v);    // row_half_mirror
  v += rn_dpp<0x140>(v);    // row_mirror
  return (rn_readlane(v, 0) + rn_readlane(v, 16)) + (rn_readlane(v, 32) + rn_readlane(v, 48));
}
__device__ __forceinline__ float wave_max(float v) {
  v = fmaxf(v, rn_dpp<0xB1>(v));
  v = fmaxf(v, rn_dpp<0x4E>(v));
  v = fmaxf(v, rn_dpp<0x141>(v));
  v = fmaxf(v, rn_dpp<0x140>(v));
  return fmaxf(fmaxf(rn_readlane(v, 0), rn_readlane(v, 16)), fmaxf(rn_readlane(v, 32), rn_readlane(v, 48)));
}
// block-wide sum for blockDim.x == 256 (4 waves); sm: >= 4 floats of LDS scratch. All threads get the result.
__device__ __forceinline__ float block_sum256(float v, float* sm) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
  __syncthreads();
  return sm[0] + sm[1] + sm[2] + sm[3];
}
__device__ __forceinline__ float block_max256(float v, float* sm) {
  v = wave_max(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
  __syncthreads();
  return fmaxf(fmaxf(sm[0], sm[1]), fmaxf(sm[2], sm[3]));
}

// Transcendentals.  The recurrent-step kernels are VALU-issue bound on tanh / sigmoid (libm tanhf is ~80
// instructions); these forms are one v_exp_f32 + one v_rcp_f32 (~12 instructions) with absolute error
// <= 3e-7, far inside the fp32-path parity bar (2e-5 on hidden states).
__device__ __forceinline__ float rn_exp(float x) { return __expf(x); }
__device__ __forceinline__ float rn_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float rn_tanh(float x) {
  const float t = __expf(-2.0f * fabsf(x));
  const float r = (1.0f - t) * __builtin_amdgcn_rcpf(1.0f + t);
  return copysignf(r, x);
}

// ---------------------------------------------------------------- Adam (shared by adam_chunk_kernel and the GEMM epilogue)
// torch.optim.Adam (single-tensor form of torch 2.10): g' = clip * (g + reg) + wd * p ; m <- lerp(m, g', 1-b1) ;
// v <- b2 v + (1-b2) g'^2 ; [vmax <- max(vmax, v)] ; p <- p - (lr / bc1) * m / (sqrt(v̂) / sqrt(bc2) + eps).
// One definition, contraction pinned per expression, so that the optimiser kernel and the weight-gradient product that applies
// the update in its epilogue (gemm_lds.hpp) round identically.
struct AdamHyper { double lr, beta1, beta2; float eps, wd, one_m_b1, beta2f, one_m_b2; int amsgrad; float reg_coef; };
// what the products of one grouped launch share when they apply Adam in their epilogue (gemm_lds.hpp)
struct AdamShared { AdamHyper hp; const int32_t* step_ptr; const float* poison; const uint32_t* pending; int step_off; int pad; };
__device__ __forceinline__ float rn_adam_update(float p, float gr, float& m, float& v, float& vmx, float k, float cl, const AdamHyper& hp,
                                                float step_size, float bc2s) {
#pragma clang fp contract(on)
  float g = (gr + k * p) * cl;
  g = g + hp.wd * p;
  m = m + hp.one_m_b1 * (g - m);
  v = v * hp.beta2f + hp.one_m_b2 * g * g;
  float vh = v;
  if (hp.amsgrad) { vh = fmaxf(vmx, v); vmx = vh; }
  const float denom = sqrtf(vh) / bc2s + hp.eps;
  return p - step_size * (m / denom);
}
