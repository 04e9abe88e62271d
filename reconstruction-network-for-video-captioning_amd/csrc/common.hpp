// Common device helpers for the RecNet gfx950 kernels (wave64, MFMA, LDS).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __bf16 bf16_t;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

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
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
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
