// MFMA GEMM for gfx950:  C[M,N] (+)= alpha * sum_k A(m,k) * B(n,k)  (+ bias[n])
//
// One kernel template covers every contraction of the RecNet train step:
//   * operand layouts: "row" = k contiguous in memory (x and W of x.W^T), "col" = the m / n index
//     contiguous (the operands of dY^T.X and of dY.W).  Row operands are staged in LDS as [m][k]
//     and read with ds_read_b128; col operands are staged as [k][m] — a straight coalesced copy —
//     and read with gfx950's transposing ds_read_b64_tr_b16, so no transposed copies of
//     activations or weights are ever materialised in HBM.
//   * compute type: bf16 MFMA (v_mfma_f32_16x16x32_bf16, fp32 accumulate) or exact fp32 MFMA
//     (v_mfma_f32_16x16x4_f32).  Sources may be fp32 (converted while staging) or pre-packed bf16.
//   * split-K over gridDim.z into fp32 slabs (consumed either by splitk_reduce or directly by the
//     fused recurrent-step kernels).
// Tile 128x128, 256 threads = 4 waves (2x2), each wave 64x64 = 4x4 MFMA tiles; register-prefetched
// single LDS buffer (global loads of tile k+1 are in flight while tile k is multiplied).
#pragma once
#include "common.hpp"

struct GemmArgs {
  const void* A; const void* B; float* C; const float* bias;
  int M, N, K; int lda, ldb, ldc;
  float alpha; int accumulate;
  int splitk; int kchunk;     // kchunk: multiple of BK; K range of slice z = [z*kchunk, min(K,(z+1)*kchunk))
  float* ws;                  // fp32 slabs [splitk][M][N] when splitk > 1
  int a_vec, b_vec;           // 16-byte vector loads are legal for this operand (alignment checked on host)
  int c_bf16;                 // direct epilogue stores bf16 into C (reinterpreted); requires splitk == 1
  void* C2; int ldc2;         // optional second, bf16 copy of the fp32 result (direct epilogue of gemm_lds only)
  // Adam in the epilogue (grouped launches of dY^T . X products, gemm_lds.hpp): the product IS the gradient of a parameter
  // block with the layout of C — p / m / v / vmax point at that block's first row, img / imgT at its packed operand image
  // ([rows][ld_img] bf16) and the transposed image ([cols][ld_imgt] bf16, may be null); null p: plain epilogue
  float* ad_p; float* ad_m; float* ad_v; float* ad_vmax; void* ad_img; void* ad_imgt; const float* ad_pnorm; int ad_ld_img, ad_ld_imgt;
  unsigned* cnt;              // grouped launches (gemm_group_kernel) with splitk > 1: one arrival counter per output tile —
                              // the slice that arrives last sums the slabs and runs the epilogue (no reduction launch)
  // MSE in the epilogue (gemm_lds.hpp, direct fp32 epilogue, splitk == 1; the local reconstructor's output layer, train.py:126-128):
  // row = s * mse_B + b is compared with ref[b * bstride + s * sstride + col]; C <- gcoef * diff (d loss / d out), C2 <- bf16
  // (lp * gcoef * diff), mse_part[tile] <- sum diff^2 of the tile (summed in tile order by rec_loss_finalize_kernel).  null: off
  const float* mse_ref; float* mse_part; size_t mse_bstride, mse_sstride; int mse_B; float mse_gcoef, mse_lp;
};

template <typename CT> struct GemmCfg;
template <> struct GemmCfg<bf16_t> {
  static constexpr int BK = 64, VW = 8, LDR = 72 /*[128][BK+8]*/, LDC = 136 /*[BK][128+8]*/;
  typedef bf16x8 vec;
};
template <> struct GemmCfg<float> {
  static constexpr int BK = 32, VW = 4, LDR = 36, LDC = 144;
  typedef f32x4 vec;
};
#define GEMM_TILE 128
#define GEMM_SMEM_BYTES 18432   // per operand, both compute types, both layouts

// ---- global -> register vector (converted to the compute type)
template <typename CT, typename ST> struct VecLoad;
template <> struct VecLoad<bf16_t, float> {
  static __device__ __forceinline__ bf16x8 full(const float* p) {
    const float4 a = *reinterpret_cast<const float4*>(p);
    const float4 b = *reinterpret_cast<const float4*>(p + 4);
    bf16x8 r;
    r[0] = (bf16_t)a.x; r[1] = (bf16_t)a.y; r[2] = (bf16_t)a.z; r[3] = (bf16_t)a.w;
    r[4] = (bf16_t)b.x; r[5] = (bf16_t)b.y; r[6] = (bf16_t)b.z; r[7] = (bf16_t)b.w;
    return r;
  }
};
template <> struct VecLoad<bf16_t, bf16_t> {
  static __device__ __forceinline__ bf16x8 full(const bf16_t* p) { return *reinterpret_cast<const bf16x8*>(p); }
};
template <> struct VecLoad<float, float> {
  static __device__ __forceinline__ f32x4 full(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
};

template <typename CT, typename ST>
__device__ __forceinline__ typename GemmCfg<CT>::vec gemm_load_vec(const ST* p, int nvalid, int vec_ok) {
  constexpr int VW = GemmCfg<CT>::VW;
  if (nvalid >= VW && vec_ok) return VecLoad<CT, ST>::full(p);
  typename GemmCfg<CT>::vec r;
#pragma unroll
  for (int j = 0; j < VW; ++j) r[j] = (j < nvalid) ? (CT)p[j] : (CT)0.0f;
  return r;
}

// Stage one operand tile (4 vectors per thread).  COL=false: tile [128 rows][BK], rows bounded by rlim,
// k bounded by kend.  COL=true: tile [BK][128], the k index selects the memory row.
template <typename CT, typename ST, bool COL>
__device__ __forceinline__ void gemm_tile_load(typename GemmCfg<CT>::vec (&reg)[4], const ST* base, int ld,
                                               int row0, int rlim, int k0, int kend, int vec_ok) {
  typedef GemmCfg<CT> G;
  const int tid = threadIdx.x;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int v = tid + i * 256;
    if (!COL) {
      constexpr int VPR = G::BK / G::VW;
      const int r = v / VPR, c = (v % VPR) * G::VW;
      const int gr = row0 + r, gk = k0 + c;
      int nv = (gr < rlim) ? (kend - gk) : 0;
      nv = nv < 0 ? 0 : nv;
      const ST* p = base + (size_t)(gr < rlim ? gr : 0) * ld + (nv > 0 ? gk : 0);
      reg[i] = gemm_load_vec<CT, ST>(p, nv, vec_ok);
    } else {
      constexpr int VPR = GEMM_TILE / G::VW;
      const int r = v / VPR, c = (v % VPR) * G::VW;
      const int gk = k0 + r, gm = row0 + c;
      int nv = (gk < kend) ? (rlim - gm) : 0;
      nv = nv < 0 ? 0 : nv;
      const ST* p = base + (size_t)(gk < kend ? gk : 0) * ld + (nv > 0 ? gm : 0);
      reg[i] = gemm_load_vec<CT, ST>(p, nv, vec_ok);
    }
  }
}

template <typename CT, bool COL>
__device__ __forceinline__ void gemm_tile_store(CT* s, const typename GemmCfg<CT>::vec (&reg)[4]) {
  typedef GemmCfg<CT> G;
  const int tid = threadIdx.x;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int v = tid + i * 256;
    if (!COL) {
      constexpr int VPR = G::BK / G::VW;
      const int r = v / VPR, c = (v % VPR) * G::VW;
      *reinterpret_cast<typename G::vec*>(s + r * G::LDR + c) = reg[i];
    } else {
      constexpr int VPR = GEMM_TILE / G::VW;
      const int r = v / VPR, c = (v % VPR) * G::VW;
      *reinterpret_cast<typename G::vec*>(s + r * G::LDC + c) = reg[i];
    }
  }
}

// ---- fragment reads
template <bool COL>
__device__ __forceinline__ bf16x8 gemm_frag(const bf16_t* s, int row /*tile row of lane 0 of the 16-row block*/,
                                            int ks /*k offset within the tile, multiple of 32*/, int lane) {
  typedef GemmCfg<bf16_t> G;
  if (!COL) {
    return *reinterpret_cast<const bf16x8*>(s + (row + (lane & 15)) * G::LDR + ks + 8 * (lane >> 4));
  } else {
    // [k][m] image: transposing read.  In each 16-lane group, lane 4q+p supplies the address of
    // k-row q, m-columns 4p..4p+3 and receives, for m-column (lane&15), the 4 k-rows.
    const int li = lane & 15, q = li >> 2, p = li & 3, g = lane >> 4;
    const bf16_t* a = s + (ks + 8 * g + q) * G::LDC + row + 4 * p;
    typedef __attribute__((address_space(3))) bf16x4 lds_b4;
    bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_b4*)(a));
    bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_b4*)(a + 4 * G::LDC));
    bf16x8 r;
    r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
    r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
    return r;
  }
}
template <bool COL>
__device__ __forceinline__ float gemm_frag(const float* s, int row, int ks /*multiple of 4*/, int lane) {
  typedef GemmCfg<float> G;
  if (!COL) return s[(row + (lane & 15)) * G::LDR + ks + (lane >> 4)];
  return s[(ks + (lane >> 4)) * G::LDC + row + (lane & 15)];
}

__device__ __forceinline__ f32x4 gemm_mma(bf16x8 a, bf16x8 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 gemm_mma(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}
template <typename CT> struct FragT;
template <> struct FragT<bf16_t> { typedef bf16x8 type; static constexpr int KSTEP = 32; };
template <> struct FragT<float> { typedef float type; static constexpr int KSTEP = 4; };

// TAG only changes the symbol name: 0 = batched GEMM, 1..5 = the recurrent-step GEMM sites (launch.hpp), so that
// rocprofv3 --stats reports each dependent-chain launch site separately from the big batched GEMMs.
template <typename CT, typename TA, typename TB, bool ACOL, bool BCOL, int TAG>
__global__ __launch_bounds__(256) void gemm_kernel(const GemmArgs p) {
  typedef GemmCfg<CT> G;
  typedef typename FragT<CT>::type frag_t;
  constexpr int KSTEP = FragT<CT>::KSTEP;
  __shared__ __attribute__((aligned(16))) char smem[2 * GEMM_SMEM_BYTES];
  CT* sA = reinterpret_cast<CT*>(smem);
  CT* sB = reinterpret_cast<CT*>(smem + GEMM_SMEM_BYTES);

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
  const int m0 = blockIdx.y * GEMM_TILE, n0 = blockIdx.x * GEMM_TILE;
  const int z = blockIdx.z;
  const int kbeg = z * p.kchunk;
  int kend = kbeg + p.kchunk;
  if (kend > p.K) kend = p.K;
  const TA* A = reinterpret_cast<const TA*>(p.A);
  const TB* B = reinterpret_cast<const TB*>(p.B);

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  typename G::vec ra[4], rb[4];
  const int nkt = (kend > kbeg) ? (kend - kbeg + G::BK - 1) / G::BK : 0;
  if (nkt > 0) {
    gemm_tile_load<CT, TA, ACOL>(ra, A, p.lda, m0, p.M, kbeg, kend, p.a_vec);
    gemm_tile_load<CT, TB, BCOL>(rb, B, p.ldb, n0, p.N, kbeg, kend, p.b_vec);
  }
  for (int kt = 0; kt < nkt; ++kt) {
    __syncthreads();                       // previous tile's fragment reads are done
    gemm_tile_store<CT, ACOL>(sA, ra);
    gemm_tile_store<CT, BCOL>(sB, rb);
    __syncthreads();
    if (kt + 1 < nkt) {                    // prefetch the next tile into registers (in flight during the MFMAs)
      const int k0 = kbeg + (kt + 1) * G::BK;
      gemm_tile_load<CT, TA, ACOL>(ra, A, p.lda, m0, p.M, k0, kend, p.a_vec);
      gemm_tile_load<CT, TB, BCOL>(rb, B, p.ldb, n0, p.N, k0, kend, p.b_vec);
    }
#pragma unroll
    for (int ks = 0; ks < G::BK; ks += KSTEP) {
      frag_t fa[4], fb[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) fa[i] = gemm_frag<ACOL>(sA, wm + i * 16, ks, lane);
#pragma unroll
      for (int j = 0; j < 4; ++j) fb[j] = gemm_frag<BCOL>(sB, wn + j * 16, ks, lane);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = gemm_mma(fa[i], fb[j], acc[i][j]);
    }
  }

  // epilogue: C/D fragment map of the 16x16 MFMA: col = lane & 15, row = (lane >> 4) * 4 + r
  const int cr = (lane >> 4) * 4, cc = lane & 15;
  if (p.splitk > 1) {
    float* W = p.ws + (size_t)z * p.M * p.N;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int col = n0 + wn + j * 16 + cc;
        if (col < p.N) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int row = m0 + wm + i * 16 + cr + r;
            if (row < p.M) W[(size_t)row * p.N + col] = acc[i][j][r];
          }
        }
      }
  } else {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int col = n0 + wn + j * 16 + cc;
        if (col < p.N) {
          const float bv = p.bias ? p.bias[col] : 0.f;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int row = m0 + wm + i * 16 + cr + r;
            if (row < p.M) {
              float v = p.alpha * acc[i][j][r] + bv;
              if (p.c_bf16) {
                reinterpret_cast<bf16_t*>(p.C)[(size_t)row * p.ldc + col] = (bf16_t)v;
              } else {
                float* dst = p.C + (size_t)row * p.ldc + col;
                if (p.accumulate) v += *dst;
                *dst = v;
              }
            }
          }
        }
      }
  }
}

// out = alpha * sum_z ws[z] + bias (+ out)
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ ws, int S, int M, int N,
                                                            float* __restrict__ C, int ldc,
                                                            const float* __restrict__ bias, float alpha,
                                                            int accumulate) {
  const size_t total = (size_t)M * N;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int row = (int)(i / N), col = (int)(i % N);
    float s = 0.f;
    for (int z = 0; z < S; ++z) s += ws[(size_t)z * total + i];
    float v = alpha * s + (bias ? bias[col] : 0.f);
    float* dst = C + (size_t)row * ldc + col;
    if (accumulate) v += *dst;
    *dst = v;
  }
}
