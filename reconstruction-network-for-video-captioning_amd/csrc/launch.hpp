// Host-side declarations shared by the translation units of librecnet_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define RN_PREC_F32 0
#define RN_PREC_BF16 1

// kernel-symbol tags of the recurrent-step GEMM launches (see gemm.hip)
#define RN_TAG_DEC_FWD 1
#define RN_TAG_DEC_BWD 2
#define RN_TAG_REC_FWD 3
#define RN_TAG_REC_BWD 4
#define RN_TAG_REC_ATT 5
#define RN_TAG_REC_ATT_BWD 6
// profile sites of the persistent reconstructor chains (rec_chain.hpp), one launch each per step
#define RN_SITE_REC_CHAIN_FWD 7
#define RN_SITE_REC_CHAIN_BWD 8
#define RN_SITE_DEC_CHAIN_FWD 9
#define RN_SITE_DEC_CHAIN_BWD 10

// ---- gemm.hip
int rn_gemm_bk(int prec);
int rn_pick_splitk(int prec, int M, int N, int K, int max_split, int chain);
int rn_effective_splitk(int prec, int K, int splitk);
// a_bf16 / b_bf16: operand memory holds bf16 (pre-packed) instead of fp32.  reduce_after: when splitk > 1,
// run the slab reduction (otherwise the caller's fused consumer sums ws[z][M][N] itself).
// MSE epilogue of the bf16 DMA kernel (GemmArgs::mse_*): row = s * B + b against ref[b * bstride + s * sstride + col]
struct RnMse { const float* ref; float* part; size_t bstride, sstride; int B; float gcoef, lp; };
void rn_launch_gemm(int prec, const void* A, int a_bf16, int a_col, int lda, const void* B, int b_bf16, int b_col,
                    int ldb, float* C, int ldc, const float* bias, int M, int N, int K, float alpha,
                    int accumulate, int splitk, float* ws, int reduce_after, hipStream_t st, int tag = 0, int c_bf16 = 0,
                    void* c2 = nullptr, int ldc2 = 0, const RnMse* mse = nullptr);

// Grouped launch: up to 8 products of ONE operand layout (a_col, b_col) with bf16 operands in one grid (gemm_lds.hpp:
// gemm_group_kernel).  Split factors are chosen here from a list-scheduling estimate over 2 workgroups per CU; a split
// product is summed inside the launch by its last-arriving slice (ws: fp32 slabs, cnt: zero-initialised tile counters that the
// kernel leaves zeroed).  Returns 0 when it launched; 1 when the group does not qualify (the caller runs the products one by one).
#ifndef GG_MAX
#define GG_MAX 8
#endif
struct RnGemmDesc {
  const void* A; int lda; const void* B; int ldb; float* C; int ldc; const float* bias;
  int M, N, K; float alpha; int accumulate; int c_bf16; void* c2; int ldc2;
  // Adam in the epilogue (gemm.hpp: GemmArgs::ad_*): null ad_p = plain product
  float* ad_p; float* ad_m; float* ad_v; float* ad_vmax; void* ad_img; void* ad_imgt; const float* ad_pnorm; int ad_ld_img, ad_ld_imgt;
};
// slots: workgroup slots the launch can expect to get (0: the whole chip, 2 per CU) — launches that run beside a persistent chain
// kernel or beside another grouped launch pass what is left to them, so that the split factors are not chosen for an empty chip.
int rn_launch_gemm_group(int a_col, int b_col, const RnGemmDesc* d, int n, float* ws, size_t ws_floats, unsigned* cnt, int cnt_words,
                         hipStream_t st, int slots = 0, const struct AdamShared* adam = nullptr, unsigned long long* stamp = nullptr);

// ---- dropout descriptor handed to kernels: seed lives in device memory so a captured graph can be
// replayed with a new seed.
struct DropDesc {
  const uint32_t* seed;   // device
  uint32_t site;
  uint32_t thr;           // floor(p * 2^32); 0 = disabled
  float inv_keep;         // 1 / (1 - p)
  int Bg;                 // global batch size
  int boff;               // global index of local caption 0
};
