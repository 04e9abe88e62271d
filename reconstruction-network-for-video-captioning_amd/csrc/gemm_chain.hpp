// Recurrent-step GEMM (forward form) for the dependent chains:  slab[z][M,N] = A[M, Kz] . B[N, Kz]^T,  M <= 128.
//
// One launch of a chain streams the packed recurrent weights (B, up to 19 MB) once against a <= 128-row activation
// block, then the next kernel of the chain needs the result: what matters is the time from launch to last store, not
// throughput.  The ring kernel (gemm_lds.hpp) keeps NS-1 k-tiles in flight per workgroup and stages the weights through
// LDS although no two waves ever share a weight element.  Here
//   * the weights go straight from global memory into the MFMA B-operand registers (lane = weight row n, 16 bytes of k:
//     exactly the 16x16x32 fragment), every load of the workgroup's whole K slice is issued before the first wait;
//   * the activation slice [128][Kz] (<= 2 k-tiles: the decoder's K = 512 / 4 slices, the local attention's product) goes
//     registers -> LDS once and is shared by the four waves, which split the tile's columns (BN / 4 each);
//   * one memory round trip per workgroup, then the MFMAs, then 16-byte slab stores.
// Same operand requirements as gemm_lds.hpp (16-byte aligned bases, leading dimensions multiples of 8, zero padding).
#pragma once
#include "gemm_lds.hpp"

#define GC_MAX_KT 4          // k-tiles (64 deep) per K slice

// NG = 16-column groups per wave (BN = 64 NG columns per workgroup); NKT = k-tiles the kernel is compiled for: the
// loads of all NKT tiles are issued unconditionally, straight-line (slices with fewer tiles re-read a valid address and
// skip the arithmetic).  Plain loads only (no LDS-DMA here: with DMA and register loads in flight together the
// compiler's wait-count pass degrades every wait to vmcnt(0)); the activation tile goes registers -> LDS.
template <int NG, int NKT, int TAG>
__global__ __launch_bounds__(256) void gemm_chain_kernel(const GemmArgs p) {
  extern __shared__ __attribute__((aligned(16))) char gc_smem[];
  constexpr int BN = 64 * NG;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n0 = blockIdx.x * BN + wave * (16 * NG);
  const int z = blockIdx.z;
  const int kbeg = z * p.kchunk;
  int kend = kbeg + p.kchunk;
  if (kend > p.K) kend = p.K;
  const int nkt = (kend > kbeg) ? (kend - kbeg + 63) / 64 : 0;     // <= NKT (host)
  const bf16_t* A = reinterpret_cast<const bf16_t*>(p.A);
  const bf16_t* B = reinterpret_cast<const bf16_t*>(p.B);

  // ---- issue everything: per k-tile 2 NG weight fragments of this wave + this thread's 4 x 16 bytes of the activations
  bf16x8 wb[NKT][2][NG], ar[NKT][4];
  const bf16_t* brow[NG];
#pragma unroll
  for (int g = 0; g < NG; ++g) {
    int n = n0 + g * 16 + (lane & 15);
    n = n < p.N ? n : p.N - 1;
    brow[g] = B + (size_t)n * p.ldb + (lane >> 4) * 8;
  }
#pragma unroll
  for (int kt = 0; kt < NKT; ++kt) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        const int k = kbeg + kt * 64 + ks * 32;
        // chunks that start at or beyond the slice end are not part of the product (zeroed below); the buffers are
        // zero padded up to their leading dimension, so a chunk that straddles K reads zeros for its tail
        const bf16_t* src = (k + (lane >> 4) * 8 < kend) ? brow[g] + k : brow[g] + kbeg;
        wb[kt][ks][g] = *reinterpret_cast<const bf16x8*>(src);
      }
#pragma unroll
    for (int i = 0; i < 4; ++i)
      ar[kt][i] = *reinterpret_cast<const bf16x8*>(
          gl_piece_src<false>(A, p.lda, 0, p.M, kbeg + (kt < nkt ? kt : 0) * 64, kend, wave * 4 + i, lane));
  }

  __builtin_amdgcn_sched_barrier(0);             // all loads are issued before anything waits on one of them

  f32x4 acc[8][NG];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int g = 0; g < NG; ++g) acc[i][g] = f32x4{0.f, 0.f, 0.f, 0.f};

#pragma unroll
  for (int kt = 0; kt < NKT; ++kt) {
    if (kt < nkt) {                              // block-uniform
      char* cur = gc_smem + kt * 16384;
      const int k0 = kbeg + kt * 64;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        bf16x8 v = ar[kt][i];
        if (k0 + 64 > kend) {                    // partial last tile of K: chunks beyond K are zero
          const int piece = wave * 4 + i, r = piece * 8 + (lane >> 3), cp = lane & 7;
          if (k0 + ((cp ^ (r & 7)) << 3) >= kend) v = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
        }
        *reinterpret_cast<bf16x8*>(cur + (wave * 4 + i) * 1024 + lane * 16) = v;
      }
      __syncthreads();                           // the whole activation tile kt is in LDS
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        bf16x8 fb[NG], fa[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) fa[i] = gl_frag<false>(cur, i * 16, ks * 32, lane);   // rows >= M: copies of row M-1
#pragma unroll
        for (int g = 0; g < NG; ++g) {
          fb[g] = wb[kt][ks][g];
          if (k0 + ks * 32 + (lane >> 4) * 8 >= kend) fb[g] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
        }
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
          for (int g = 0; g < NG; ++g) acc[i][g] = gemm_mma(fa[i], fb[g], acc[i][g]);
      }
    }
  }
  const int mblocks = (p.M + 15) >> 4;           // 16-row blocks that hold real rows (<= 8)
  // ---- epilogue: 128 x (16 NG) per wave -> wave-private LDS block -> 16-byte stores of whole row segments
  __syncthreads();                             // all waves are done with the activation tiles
  constexpr int WC = 16 * NG, LDS_LD = WC + 4;
  float* stg = reinterpret_cast<float*>(gc_smem) + wave * (32 * LDS_LD);
  const int cr = (lane >> 4) * 4, cc = lane & 15;
  const bool to_slab = p.splitk > 1;
  float* Cb = to_slab ? p.ws + (size_t)z * p.M * p.N : p.C;
  const int ldc = to_slab ? p.N : p.ldc;
  const bool vec4 = ((ldc & 3) == 0) && ((((uintptr_t)Cb) & 15) == 0);
  constexpr int LPR = WC / 4;                  // lanes per row (4 floats each)
  constexpr int RPI = 64 / LPR;                // rows per store instruction
#pragma unroll
  for (int q4 = 0; q4 < 4; ++q4) {             // 32 rows at a time
    if (q4 * 2 < mblocks) {
#pragma unroll
      for (int ii = 0; ii < 2; ++ii)
#pragma unroll
        for (int g = 0; g < NG; ++g)
#pragma unroll
          for (int r = 0; r < 4; ++r) stg[(ii * 16 + cr + r) * LDS_LD + g * 16 + cc] = acc[q4 * 2 + ii][g][r];
#pragma unroll
      for (int it = 0; it < 32 / RPI; ++it) {
        const int rl = it * RPI + lane / LPR, c4 = (lane % LPR) * 4;
        const f32x4 v = *reinterpret_cast<const f32x4*>(stg + rl * LDS_LD + c4);
        const int row = q4 * 32 + rl, col = n0 + c4;
#ifdef GC_PROBE_SKIP_STORE
        if (row < p.M && col < p.N && v[0] == 12345.678f) {
#else
        if (row < p.M && col < p.N) {
#endif
          float* dst = Cb + (size_t)row * ldc + col;
          if (to_slab) {
            if (vec4 && col + 3 < p.N) *reinterpret_cast<f32x4*>(dst) = v;
            else for (int q = 0; q < 4 && col + q < p.N; ++q) dst[q] = v[q];
          } else {
            for (int q = 0; q < 4 && col + q < p.N; ++q) {
              float w = p.alpha * v[q] + (p.bias ? p.bias[col + q] : 0.f);
              if (p.accumulate) w += dst[q];
              dst[q] = w;
            }
          }
        }
      }
    }
  }
}
