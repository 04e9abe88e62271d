// Hand-written gfx950 kernels of the RecNet train step other than the MFMA GEMM (gemm.hpp):
// embedding gather, the fused per-caption recurrent-step kernels (LSTM gate pointwise + Bahdanau
// attention with the caption's encoder states streamed once per step), masked cross-entropy with
// logits dropout, reconstruction MSE, their backward mirrors, norms and the multi-tensor Adam.
// All arithmetic here is fp32; only GEMM operands are ever rounded to bf16.
#pragma once
#include "common.hpp"
#include "launch.hpp"

// =============================================================================================
// small utilities
// =============================================================================================
__global__ void set_u32_kernel(uint32_t* p, uint32_t v) { *p = v; }
__global__ void set_f32_kernel(float* p, float v) { *p = v; }
// step counter += 1; seed slot = seed_base + step (graph-replay friendly train step)
__global__ void advance_step_kernel(int32_t* step, uint32_t* seed_slot, uint32_t seed_base) {
  int s = *step + 1; *step = s; *seed_slot = seed_base + (uint32_t)s;
}

// sum_{z<n} p[z*stride] over the split-K slabs.  All (<= 16) loads are issued back to back and reduced as a
// tree: a dependent round trip to L2 / memory costs ~1-3 us in these low-occupancy chain kernels, so the
// number of serialized load rounds, not bytes, sets their run time (PMC: SQ_WAIT_ANY ~75 % of wave cycles).
__device__ __forceinline__ float sum_strided(const float* __restrict__ p, size_t stride, int n) {
  if (n <= 16) {
    float v[16];
#pragma unroll
    for (int z = 0; z < 16; ++z) {
      const int zz = z < n ? z : n - 1;                 // clamped: branch-free, the duplicates hit L1
      v[z] = p[(size_t)zz * stride];
    }
#pragma unroll
    for (int z = 0; z < 16; ++z) v[z] = z < n ? v[z] : 0.f;
#pragma unroll
    for (int w = 8; w >= 1; w >>= 1)
#pragma unroll
      for (int z = 0; z < w; ++z) v[z] += v[z + w];
    return v[0];
  }
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  int z = 0;
  for (; z + 4 <= n; z += 4) {
    const float a = p[(size_t)z * stride], b = p[(size_t)(z + 1) * stride], c = p[(size_t)(z + 2) * stride], d = p[(size_t)(z + 3) * stride];
    s0 += a; s1 += b; s2 += c; s3 += d;
  }
  for (; z < n; ++z) s0 += p[(size_t)z * stride];
  return (s0 + s1) + (s2 + s3);
}
// 8 consecutive operand elements as floats (16-byte aligned for bf16, 32-byte for float)
__device__ __forceinline__ void load8(const bf16_t* p, float (&v)[8]) {
  const bf16x8 x = *reinterpret_cast<const bf16x8*>(p);
#pragma unroll
  for (int j = 0; j < 8; ++j) v[j] = (float)x[j];
}
__device__ __forceinline__ void load8(const float* p, float (&v)[8]) {
  const f32x4 a = *reinterpret_cast<const f32x4*>(p), b = *reinterpret_cast<const f32x4*>(p + 4);
  v[0] = a[0]; v[1] = a[1]; v[2] = a[2]; v[3] = a[3]; v[4] = b[0]; v[5] = b[1]; v[6] = b[2]; v[7] = b[3];
}

template <typename AT> struct Raw8;
template <> struct Raw8<bf16_t> {
  bf16x8 v;
  __device__ __forceinline__ void load(const bf16_t* p) { v = *reinterpret_cast<const bf16x8*>(p); }
  __device__ __forceinline__ void zero() { for (int j = 0; j < 8; ++j) v[j] = (bf16_t)0.f; }
  __device__ __forceinline__ float at(int j) const { return (float)v[j]; }
};
template <> struct Raw8<float> {
  f32x4 a, b;
  __device__ __forceinline__ void load(const float* p) { a = *reinterpret_cast<const f32x4*>(p); b = *reinterpret_cast<const f32x4*>(p + 4); }
  __device__ __forceinline__ void zero() { a = f32x4{0.f, 0.f, 0.f, 0.f}; b = a; }
  __device__ __forceinline__ float at(int j) const { return j < 4 ? a[j] : b[j - 4]; }
};

// out[c] += sum_r X[r*ld + c], 16-byte loads: a wave covers 512 columns of one row per instruction, the four waves of a
// workgroup take rows r0 + w, r0 + w + 4, ...; grid (ceil(cols / 512), row splits); `out` pre-zeroed (float atomics,
// one per column per workgroup).  Needs ld % 8 == 0 and a 16-byte aligned base (true for every operand buffer).
template <typename ST>
__global__ __launch_bounds__(256) void colsum_vec_kernel(const ST* __restrict__ X, int rows, int cols, int ld,
                                                         float* __restrict__ out) {
  __shared__ float sm[4][64][9];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c0 = blockIdx.x * 512 + lane * 8;
  const int rs = gridDim.y, per = (rows + rs - 1) / rs;
  const int r0 = blockIdx.y * per, r1 = min(rows, r0 + per);
  float acc[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) acc[j] = 0.f;
  if (c0 < cols) {
    int r = r0 + wave;
    for (; r + 4 < r1; r += 8) {          // two rows in flight
      Raw8<ST> a, b;
      a.load(X + (size_t)r * ld + c0); b.load(X + (size_t)(r + 4) * ld + c0);
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] += a.at(j) + b.at(j);
    }
    for (; r < r1; r += 4) {
      Raw8<ST> a; a.load(X + (size_t)r * ld + c0);
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] += a.at(j);
    }
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) sm[wave][lane][j] = acc[j];
  __syncthreads();
  if (wave == 0 && c0 < cols) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float v = sm[0][lane][j] + sm[1][lane][j] + sm[2][lane][j] + sm[3][lane][j];
      if (c0 + j < cols) atomicAdd(out + c0 + j, v);
    }
  }
}

__device__ __forceinline__ uint32_t drop_key(const DropDesc& dd) { return rn_site_key(*dd.seed, dd.site); }
__device__ __forceinline__ float drop_at(const DropDesc& dd, uint32_t key, int t, int b, int N, int j) {
  const uint32_t idx = ((uint32_t)t * (uint32_t)dd.Bg + (uint32_t)(dd.boff + b)) * (uint32_t)N + (uint32_t)j;
  return rn_drop_scale(key, dd.thr, dd.inv_keep, idx);
}

// out[i] = scale * sum_j x[j]   (single block; deterministic order)
__global__ __launch_bounds__(256) void reduce_sum_kernel(const float* __restrict__ x, int n, float* out, float scale) {
  __shared__ float sm[4];
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) s += x[i];
  s = block_sum256(s, sm);
  if (threadIdx.x == 0) *out = s * scale;
}

// out[c] (+)= sum_r X[r*ld + c].  grid (ceil(cols/64), RS); with RS > 1 `out` must be pre-zeroed (atomics).
template <typename ST>
__global__ __launch_bounds__(256) void colsum_kernel(const ST* __restrict__ X, int rows, int cols, int ld,
                                                     float* __restrict__ out, int use_atomic) {
  __shared__ float sm[4][64];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63), rg = threadIdx.x >> 6;
  const int rs = gridDim.y, per = (rows + rs - 1) / rs;
  const int r0 = blockIdx.y * per, r1 = min(rows, r0 + per);
  float s = 0.f;
  if (c < cols)
    for (int r = r0 + rg; r < r1; r += 4) s += (float)X[(size_t)r * ld + c];
  sm[rg][threadIdx.x & 63] = s;
  __syncthreads();
  if (rg == 0 && c < cols) {
    s = sm[0][threadIdx.x] + sm[1][threadIdx.x] + sm[2][threadIdx.x] + sm[3][threadIdx.x];
    if (use_atomic) atomicAdd(out + c, s); else out[c] = s;
  }
}

// out[i] = a[i] + b[i]  (biases b_ih + b_hh)
__global__ void add2_kernel(const float* a, const float* b, float* out, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = a[i] + b[i];
}
template <typename T>
__global__ void scale_kernel(T* x, size_t n, float s) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    x[i] = (T)((float)x[i] * s);
}
// operand copy: dst[r][c] = (AT)(scale * src[r*ld_src + c]) for c < cols, 0 for cols <= c < ld_dst (the zero
// padding the DMA-staged GEMM relies on).  Used for enc, the packed weight images and dout.
template <typename AT>
__global__ void pack_block_kernel(AT* __restrict__ dst, int ld_dst, const float* __restrict__ src, int ld_src, int rows,
                                  int cols, float scale) {
  const size_t total = (size_t)rows * ld_dst;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int r = (int)(i / ld_dst), c = (int)(i % ld_dst);
    dst[i] = (AT)(c < cols ? scale * src[(size_t)r * ld_src + c] : 0.f);
  }
}
// dst[r] = [src1[r, 0:c1) | src2[r, 0:c2) | 0 ...] with leading dimension ld_dst  (concatenated weight image)
template <typename DT>
__global__ void pack2_kernel(DT* __restrict__ dst, int ld_dst, const float* __restrict__ src1, int ld1, int c1,
                             const float* __restrict__ src2, int ld2, int c2, int rows) {
  const size_t total = (size_t)rows * ld_dst;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int r = (int)(i / ld_dst), c = (int)(i % ld_dst);
    float v = 0.f;
    if (c < c1) v = src1[(size_t)r * ld1 + c];
    else if (c < c1 + c2) v = src2[(size_t)r * ld2 + (c - c1)];
    dst[i] = (DT)v;
  }
}
// Recurrent weights into the 4-block gate layout: packed row (q * Hd + u) takes master row (map[q] * Hd + u) of
// src1 (columns [0,c1)) and of src2 (columns [c1, c1+c2)), zeros where map[q] < 0 and in the padding.
// LSTM: map = {0,1,2,3}.  GRU: W_ih map {0,1,2,-1}, W_hh map {0,1,-1,2}  (see gru_point).
struct GateMap { int m[4]; };
template <typename DT>
__global__ void pack_gates_kernel(DT* __restrict__ dst, int ld_dst, int Hd, const float* __restrict__ src1, int ld1, int c1,
                                  GateMap map1, const float* __restrict__ src2, int ld2, int c2, GateMap map2) {
  const size_t total = (size_t)4 * Hd * ld_dst;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int r = (int)(i / ld_dst), c = (int)(i % ld_dst), q = r / Hd, u = r - q * Hd;
    float v = 0.f;
    if (c < c1) { if (map1.m[q] >= 0) v = src1[(size_t)(map1.m[q] * Hd + u) * ld1 + c]; }
    else if (c < c1 + c2) { if (map2.m[q] >= 0) v = src2[(size_t)(map2.m[q] * Hd + u) * ld2 + (c - c1)]; }
    dst[i] = (DT)v;
  }
}
// dst[c][r] = src[r][c]  (32 x 32 tiles through LDS): the K-contiguous image of a weight that a backward chain GEMM
// uses as its "col" operand, so that it can load fragments straight into MFMA registers (gemm_chain.hpp)
template <typename AT>
__global__ __launch_bounds__(256) void transpose_at_kernel(const AT* __restrict__ src, int ld_src, int rows, int cols,
                                                           AT* __restrict__ dst, int ld_dst) {
  __shared__ AT tile[32][33];
  const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32, tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int i = ty; i < 32; i += 8) {
    const int r = r0 + i, c = c0 + tx;
    tile[i][tx] = (r < rows && c < cols) ? src[(size_t)r * ld_src + c] : (AT)0.f;
  }
  __syncthreads();
  for (int i = ty; i < 32; i += 8) {
    const int c = c0 + i, r = r0 + tx;                 // dst row = source column; pad columns [rows, ld_dst) get zeros
    if (c < cols && r < ld_dst) dst[(size_t)c * ld_dst + r] = tile[tx][i];
  }
}
// gate-interleaved image of a [4 Hd][cols] recurrent weight (rec_step.hpp): destination row (u/8)*32 + gate*8 + u%8
template <typename DT>
__global__ void pack_interleave_kernel(DT* __restrict__ dst, int ld_dst, int Hd, const float* __restrict__ src, int ld_src, int cols) {
  const size_t total = (size_t)4 * Hd * ld_dst;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int d = (int)(i / ld_dst), c = (int)(i % ld_dst);
    const int j = d >> 5, gate = (d >> 3) & 3, ul = d & 7;
    dst[i] = (DT)(c < cols ? src[(size_t)(gate * Hd + j * 8 + ul) * ld_src + c] : 0.f);
  }
}
// dst[r][c] = sum_j src[r*ld_src + j*cols + c]  (sum of NCH side-by-side partial blocks), zero padded to ld_dst
template <typename AT>
__global__ void sum_chunks_kernel(AT* __restrict__ dst, int ld_dst, const AT* __restrict__ src, int ld_src, int rows,
                                  int cols, int nch) {
  const size_t total = (size_t)rows * ld_dst;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int r = (int)(i / ld_dst), c = (int)(i % ld_dst);
    float v = 0.f;
    if (c < cols) for (int j = 0; j < nch; ++j) v += (float)src[(size_t)r * ld_src + j * cols + c];
    dst[i] = (AT)v;
  }
}
__global__ void copy_kernel(const float* __restrict__ x, float* __restrict__ y, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) y[i] = x[i];
}

// =============================================================================================
// LSTM gate math (torch.nn.LSTM order i, f, g, o)
// =============================================================================================
struct LstmOut { float i, f, g, o, c, h; };
__device__ __forceinline__ LstmOut lstm_point(float gi, float gf, float gg, float go, float c_prev) {
  LstmOut r;
  r.i = rn_sigmoid(gi);
  r.f = rn_sigmoid(gf);
  r.g = rn_tanh(gg);
  r.o = rn_sigmoid(go);
  r.c = r.f * c_prev + r.i * r.g;
  r.h = r.o * rn_tanh(r.c);
  return r;
}
struct LstmGrad { float di, df, dg, d_o, dc_prev; };
__device__ __forceinline__ LstmGrad lstm_point_bwd(float dh, float dc_in, float i, float f, float g, float o,
                                                   float c, float c_prev) {
  const float tc = rn_tanh(c);
  const float dc = dc_in + dh * o * (1.f - tc * tc);
  LstmGrad r;
  r.d_o = dh * tc * o * (1.f - o);
  r.di = dc * g * i * (1.f - i);
  r.df = dc * c_prev * f * (1.f - f);
  r.dg = dc * i * (1.f - g * g);
  r.dc_prev = dc * f;
  return r;
}

// =============================================================================================
// GRU gate math (torch.nn.GRU order r, z, n) in the library's 4-block gate layout
//   block 0 = r, block 1 = z (input + hidden parts summed), block 2 = W_in x + b_in, block 3 = W_hn h + b_hn :
// the packed weights hold W_ih as blocks (r, z, n, 0) and W_hh as blocks (r, z, 0, n), so every GEMM, slab and
// gate-gradient row of the LSTM path is reused as is and only this pointwise part differs.
//   n = tanh(g2 + r * g3) ; h = (1 - z) n + z h_prev.  Saved activations: (r, z, n, g3).
// =============================================================================================
struct GruOut { float r, z, n, hn, h; };
__device__ __forceinline__ GruOut gru_point(float g0, float g1, float g2, float g3, float h_prev) {
  GruOut o;
  o.r = rn_sigmoid(g0);
  o.z = rn_sigmoid(g1);
  o.hn = g3;
  o.n = rn_tanh(g2 + o.r * g3);
  o.h = (1.f - o.z) * o.n + o.z * h_prev;
  return o;
}
// returns the gate-block gradients in (di, df, dg, d_o) = (d g0, d g1, d g2, d g3) and dc_prev = the direct part
// of d h_prev (dh * z); the part through W_hh comes from the next GEMM like the LSTM's.
__device__ __forceinline__ LstmGrad gru_point_bwd(float dh, float r, float z, float n, float hn, float h_prev) {
  LstmGrad g;
  const float dn = dh * (1.f - z) * (1.f - n * n);
  g.dg = dn;
  g.d_o = dn * r;
  g.di = dn * hn * r * (1.f - r);
  g.df = dh * (h_prev - n) * z * (1.f - z);
  g.dc_prev = dh * z;
  return g;
}
// bias of the 4-block gate layout: LSTM b_ih + b_hh ; GRU (b_ir + b_hr, b_iz + b_hz, b_in, b_hn)
__global__ void gate_bias_kernel(const float* __restrict__ bih, const float* __restrict__ bhh, float* __restrict__ out,
                                 int Hd, int gru) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= 4 * Hd) return;
  if (!gru) { out[i] = bih[i] + bhh[i]; return; }
  const int blk = i / Hd;
  out[i] = blk < 2 ? bih[i] + bhh[i] : (blk == 2 ? bih[i] : bhh[i - Hd]);
}
// zeroes up to 8 buffers in one launch (the targets of the step's atomic column sums / scatter-add)
struct ZeroList { float* p[8]; size_t n[8]; int cnt; };
__global__ __launch_bounds__(256) void zero_list_kernel(const ZeroList z) {
  for (int k = 0; k < z.cnt; ++k) {
    float* p = z.p[k];
    const size_t n = z.n[k];
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = 0.f;
  }
}
// gradients of the two bias vectors from the column sums of the 4-block gate gradients
__global__ void gate_bias_grad_kernel(const float* __restrict__ sum4, float* __restrict__ dbih, float* __restrict__ dbhh,
                                      int Hd, int gru) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= 4 * Hd) return;
  if (!gru) { dbih[i] = sum4[i]; dbhh[i] = sum4[i]; return; }
  const int blk = i / Hd;
  if (blk < 2) { dbih[i] = sum4[i]; dbhh[i] = sum4[i]; }
  else if (blk == 2) dbih[i] = sum4[i];
  else dbhh[i - Hd] = sum4[i];
}

// =============================================================================================
// embedding  (decoder.py:46-48)
// =============================================================================================
// emb[row, :] = scale * Emb[token(row), :] * dropmask ; row = (t - t0) * B + b
template <typename AT>
__global__ __launch_bounds__(128) void embed_fwd_kernel(const float* __restrict__ Emb, const int64_t* __restrict__ targets,
                                                        const int64_t* __restrict__ tokens, AT* __restrict__ emb, int ld,
                                                        int B, int E, int V, float scale, DropDesc dd, int t0) {
  const int row = blockIdx.x, t = t0 + row / B, b = row % B;
  long tok = tokens ? tokens[b] : (t == 0 ? 1 : targets[(size_t)(t - 1) * B + b]);
  tok = tok < 0 ? 0 : (tok >= V ? V - 1 : tok);
  const uint32_t key = drop_key(dd);
  const float* src = Emb + (size_t)tok * E;
  AT* dst = emb + (size_t)row * ld;
  for (int j = threadIdx.x; j < ld; j += 128) dst[j] = (AT)(j < E ? src[j] * scale * drop_at(dd, key, t, b, E, j) : 0.f);
}
// input token of decoder step t for caption b: the tokens that were actually fed when a free-running forward recorded
// them (in_tok [T][B], train.py:47-51), else teacher forcing: <SOS> at t = 0, targets[t-1] after (train.py:25,45)
__device__ __forceinline__ long rn_input_token(const int64_t* __restrict__ in_tok, const int64_t* __restrict__ targets, int t, int b,
                                               int B, int V) {
  long tok = in_tok ? in_tok[(size_t)t * B + b] : (t == 0 ? 1 : targets[(size_t)(t - 1) * B + b]);
  return tok < 0 ? 0 : (tok >= V ? V - 1 : tok);
}
// dEmb[token(row), :] += scale * dropmask * demb[row, :]   (dEmb pre-zeroed)
__global__ __launch_bounds__(128) void embed_bwd_kernel(float* __restrict__ dEmb, const int64_t* __restrict__ targets,
                                                        const float* __restrict__ demb, int B, int E, int V,
                                                        float scale, DropDesc dd, int row0, const int64_t* __restrict__ in_tok) {
  const int row = row0 + blockIdx.x, t = row / B, b = row % B;
  const long tok = rn_input_token(in_tok, targets, t, b, B, V);
  if (tok < 3) return;        // <PAD> / <SOS> / <EOS> rows are summed by embed_bwd_hot_kernel (a third of all rows hit them)
  const uint32_t key = drop_key(dd);
  const float* src = demb + (size_t)row * E;
  float* dst = dEmb + (size_t)tok * E;
  for (int j = threadIdx.x; j < E; j += 128) {
    const float m = drop_at(dd, key, t, b, E, j);
    if (m != 0.f) atomicAdd(dst + j, src[j] * scale * m);
  }
}

// The three special tokens feed ~40 % of all (t, b) positions (every position after a caption's <EOS> is <PAD>, step 0
// is <SOS>): through the generic kernel that is >1000 atomics on each of the same E addresses.  Here a block (column
// chunk, 32-row slice) sums the matching rows of its slice in registers and issues one atomic per token and column.
#define RN_HOT_ROWS 32
__global__ __launch_bounds__(128) void embed_bwd_hot_kernel(float* __restrict__ dEmb, const int64_t* __restrict__ targets,
                                                            const float* __restrict__ demb, int B, int E, int V,
                                                            float scale, DropDesc dd, int row0, int nrow,
                                                            const int64_t* __restrict__ in_tok) {
  __shared__ int stok[RN_HOT_ROWS];
  const int j = blockIdx.x * 128 + threadIdx.x, i0 = blockIdx.y * RN_HOT_ROWS;
  if (threadIdx.x < RN_HOT_ROWS) {   // tokens of this block's rows (so the row loop below has no dependent global load)
    const int i = i0 + threadIdx.x;
    int tk = -1;
    if (i < nrow) {
      const int row = row0 + i, t = row / B, b = row - t * B;
      tk = (int)rn_input_token(in_tok, targets, t, b, B, V);
    }
    stok[threadIdx.x] = tk;
  }
  __syncthreads();
  if (j >= E) return;
  const uint32_t key = drop_key(dd);
  float acc[3] = {0.f, 0.f, 0.f};
  // branch-free in groups of 8 rows: the 8 loads are in flight together (rows of other tokens are read and discarded)
#pragma unroll
  for (int g = 0; g < RN_HOT_ROWS; g += 8) {
    float v[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int i = i0 + g + q;
      v[q] = (i < nrow && stok[g + q] >= 0 && stok[g + q] <= 2) ? demb[(size_t)(row0 + i) * E + j] : 0.f;
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int tk = stok[g + q];
      if (tk < 0 || tk > 2) continue;
      const int row = row0 + i0 + g + q, t = row / B, b = row - t * B;
      const float x = v[q] * scale * drop_at(dd, key, t, b, E, j);
      acc[0] += tk == 0 ? x : 0.f; acc[1] += tk == 1 ? x : 0.f; acc[2] += tk == 2 ? x : 0.f;
    }
  }
#pragma unroll
  for (int q = 0; q < 3; ++q)
    if (acc[q] != 0.f) atomicAdd(dEmb + (size_t)q * E + j, acc[q]);
}

// =============================================================================================
// decoder recurrent step (decoder.py:50-66), forward.
// Exact-math restructuring: ctx_t . W_c^T = (1/F) sum_f a_t[f] (enc[b,f] . W_c^T) = (1/F) sum_f a_t[f] P[b,f,:]
// with P = enc . W_ih[:, E:]^T hoisted out of the time loop, so the only GEMM left in the chain is
// h_{t-1} . [W_hh ; attn_W]^T (K = H).  One workgroup per (caption, 64-hidden-unit chunk):
//   Wh  = slab columns [4H, 4H+A)                           (attn_W h_{t-1})
//   a[f] = w . tanh(Wh + Uv[b,f] + b)                        one wave per frame, wavefront reduction
//   gates[col] = Xe[t,b,col] + h.W_hh^T (slabs) + (1/F) sum_f a[f] P[b,f,col]   for the chunk's 4 x 64 columns
//   LSTM pointwise -> h_t (fp32 + operand copy), c_t, saved activations
// Workgroup = 4 gates x UC units (UC = blockDim.x / 4, 64..256): big workgroups keep the per-caption score work and
// the Uv / slab re-reads (the kernel is bound by bytes pulled from the memory side) to 2 per caption.
// AT = operand type of the GEMM inputs this kernel reads / writes (bf16 in the bf16 path, float in the exact path).
// =============================================================================================
#define RN_UC_MAX 256       // hidden units per workgroup = blockDim.x / 4 (one thread per gate per unit)
struct DecCellArgs {
  int t, B, F, H, A, S;
  int gru;                // 1: GRU pointwise (c_prev = h_{t-1}, c_out unused), see gru_point
  const float* slab;      // [S][B][4H+A] split-K partials of h_{t-1} . [W_hh ; W]^T, nullptr when h_{t-1} = 0
  const float* Xe;        // [B][4H] of step t (emb . W_e^T + b_ih + b_hh)
  const void* P;          // [B*F][ldp] AT
  int ldp;
  const float* Uv;        // [B][F][A]
  const float* ab; const float* w;
  const float* c_prev;    // [B][H] or nullptr (zeros)
  float* h_out; float* c_out;   // [B][H] of step t
  void* h_lp; int ld_hlp;       // [B][ld_hlp] AT copy of h_t (next step's GEMM operand; zero padded) or nullptr
  float* acts;            // [B][4H] post-activation gates or nullptr
  float* Wh_out;          // [B][A] or nullptr
  float* att_out;         // [B][F] or nullptr
};

template <typename AT>
__global__ __launch_bounds__(1024) void dec_cell_kernel(const DecCellArgs p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* swh = smem;            // [A]
  float* sa = swh + p.A;        // [F]
  float* spre = sa + p.F;       // [4 * UC]
  const int NT = blockDim.x, UC = NT >> 2, NW = NT >> 6;
  const int b = blockIdx.x, u0 = blockIdx.y * UC, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int H = p.H, A = p.A, F = p.F, W4 = 4 * H, WS = 4 * H + A;
  const size_t zs = (size_t)p.B * WS;
  // ---- every global load of the kernel is issued up front (the kernel is one link of a dependent chain and
  // runs at ~3 waves per SIMD, so exposed memory latency, not bandwidth, is what it costs)
  const int g = tid / UC, ul = tid % UC, u = u0 + ul;   // gate phase: thread = (gate, hidden unit of the chunk)
  float pre = 0.f;
  float pv[32];                               // P[b, f, col] for f < min(F, 32)
#pragma unroll
  for (int f = 0; f < 32; ++f) pv[f] = 0.f;
  const AT* pp = nullptr;
  if (u < H) {
    const int col = g * H + u;
    pp = reinterpret_cast<const AT*>(p.P) + (size_t)b * F * p.ldp + col;
#pragma unroll
    for (int f = 0; f < 32; ++f) if (f < F) pv[f] = (float)pp[(size_t)f * p.ldp];
    pre = p.Xe[(size_t)b * W4 + col];
    if (p.slab) pre += sum_strided(p.slab + (size_t)b * WS + col, zs, p.S);
  }
  float cprev = 0.f;
  if (tid < UC && u0 + tid < H && p.c_prev) cprev = p.c_prev[(size_t)b * H + u0 + tid];
  // score phase operands: wave w handles frames w, w+NW, ...; lane handles k = lane, lane + 64 (A <= 128 fast path)
  float uvr[8][2];
  const bool fastA = (A <= 128) && (F <= 8 * NW);
  if (fastA) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int f = wave + NW * i;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int k = lane + 64 * j;
        uvr[i][j] = (f < F && k < A) ? p.Uv[((size_t)b * F + f) * A + k] : 0.f;
      }
    }
  }
  for (int k = tid; k < A; k += NT) {
    const float v = p.slab ? sum_strided(p.slab + (size_t)b * WS + W4 + k, zs, p.S) : 0.f;
    swh[k] = v;
    if (p.Wh_out && blockIdx.y == 0) p.Wh_out[(size_t)b * A + k] = v;
  }
  __syncthreads();
  if (fastA) {
    float wk[2], bk[2], hk[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int k = lane + 64 * j;
      wk[j] = k < A ? p.w[k] : 0.f; bk[j] = k < A ? p.ab[k] : 0.f; hk[j] = k < A ? swh[k] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int f = wave + NW * i;
      if (f < F) {
        float s = wk[0] * rn_tanh(hk[0] + uvr[i][0] + bk[0]);
        if (A > 64) s += wk[1] * rn_tanh(hk[1] + uvr[i][1] + bk[1]);
        s = wave_sum(s);
        if (lane == 0) { sa[f] = s; if (p.att_out && blockIdx.y == 0) p.att_out[(size_t)b * F + f] = s; }
      }
    }
  } else {
    for (int f = wave; f < F; f += NW) {
      const float* uv = p.Uv + ((size_t)b * F + f) * A;
      float s = 0.f;
      for (int k = lane; k < A; k += 64) s += p.w[k] * rn_tanh(swh[k] + uv[k] + p.ab[k]);
      s = wave_sum(s);
      if (lane == 0) { sa[f] = s; if (p.att_out && blockIdx.y == 0) p.att_out[(size_t)b * F + f] = s; }
    }
  }
  __syncthreads();
  if (u < H) {
    float c0 = 0.f, c1 = 0.f;
#pragma unroll
    for (int f = 0; f < 32; f += 2) {
      if (f < F) c0 += sa[f] * pv[f];
      if (f + 1 < F) c1 += sa[f + 1] * pv[f + 1];
    }
    for (int f = 32; f < F; ++f) c0 += sa[f] * (float)pp[(size_t)f * p.ldp];
    pre += (c0 + c1) * (1.0f / (float)F);
  }
  spre[g * UC + ul] = pre;
  __syncthreads();
  if (tid < UC && u0 + tid < H) {
    const int uu = u0 + tid;
    const size_t o = (size_t)b * H + uu;
    float hv, a0, a1, a2, a3;
    if (p.gru) {
      const GruOut r = gru_point(spre[tid], spre[UC + tid], spre[2 * UC + tid], spre[3 * UC + tid], cprev);
      hv = r.h; a0 = r.r; a1 = r.z; a2 = r.n; a3 = r.hn;
    } else {
      const LstmOut r = lstm_point(spre[tid], spre[UC + tid], spre[2 * UC + tid], spre[3 * UC + tid], cprev);
      hv = r.h; a0 = r.i; a1 = r.f; a2 = r.g; a3 = r.o;
      p.c_out[o] = r.c;
    }
    p.h_out[o] = hv;
    if (p.h_lp) reinterpret_cast<AT*>(p.h_lp)[(size_t)b * p.ld_hlp + uu] = (AT)hv;
    if (p.acts) {
      float* a = p.acts + (size_t)b * W4 + uu;
      a[0] = a0; a[H] = a1; a[2 * H] = a2; a[3 * H] = a3;
    }
  }
  // zero padding of the operand copy (columns [H, ld_hlp)), once per row
  if (p.h_lp && blockIdx.y == 0)
    for (int j = H + tid; j < p.ld_hlp; j += NT) reinterpret_cast<AT*>(p.h_lp)[(size_t)b * p.ld_hlp + j] = (AT)0.f;
}

// Vector form of dec_cell_kernel for the bf16 path (H % 8 == 0, F <= 32, A <= 128): wave = gate, each lane owns 8
// consecutive hidden units, so every P / Xe / slab access is a 16-byte load and a wave-instruction covers 1 KiB of
// one row (the generic kernel reads P with 2-byte loads, 128 B per instruction).  One workgroup covers 512 units
// x 4 gates of one caption; all loads are issued before the scores are computed.
template <typename AT>
__global__ __launch_bounds__(256) void dec_cell_vec_kernel(const DecCellArgs p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* swh = smem;            // [A]
  float* sa = swh + p.A;        // [F] (+ pad to 16 B)
  float* spre = sa + ((p.F + 3) & ~3);   // [4][512]
  const int b = blockIdx.x, u0 = blockIdx.y * 512, tid = threadIdx.x, lane = tid & 63, g = tid >> 6;
  const int H = p.H, A = p.A, F = p.F, W4 = 4 * H, WS = 4 * H + A;
  const size_t zs = (size_t)p.B * WS;
  const int u = u0 + lane * 8;
  const bool live = u < H;                    // H % 8 == 0: a lane's 8 units are all inside or all outside
  const int col = g * H + u;
  Raw8<AT> pv[32];
  float pre[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) pre[j] = 0.f;
  if (live) {
    const AT* pp = reinterpret_cast<const AT*>(p.P) + (size_t)b * F * p.ldp + col;
#pragma unroll
    for (int f = 0; f < 32; ++f) { if (f < F) pv[f].load(pp + (size_t)f * p.ldp); else pv[f].zero(); }
    const f32x4 x0 = *reinterpret_cast<const f32x4*>(p.Xe + (size_t)b * W4 + col);
    const f32x4 x1 = *reinterpret_cast<const f32x4*>(p.Xe + (size_t)b * W4 + col + 4);
    pre[0] = x0[0]; pre[1] = x0[1]; pre[2] = x0[2]; pre[3] = x0[3]; pre[4] = x1[0]; pre[5] = x1[1]; pre[6] = x1[2]; pre[7] = x1[3];
    if (p.slab) {
      for (int z = 0; z < p.S; ++z) {
        const float* sp = p.slab + z * zs + (size_t)b * WS + col;
        const f32x4 s0 = *reinterpret_cast<const f32x4*>(sp), s1 = *reinterpret_cast<const f32x4*>(sp + 4);
        pre[0] += s0[0]; pre[1] += s0[1]; pre[2] += s0[2]; pre[3] += s0[3];
        pre[4] += s1[0]; pre[5] += s1[1]; pre[6] += s1[2]; pre[7] += s1[3];
      }
    }
  }
  // score operands: wave g handles frames g, g + 4, ...; lane handles k = lane, lane + 64
  float uvr[8][2];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int f = g + 4 * i;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int k = lane + 64 * j;
      uvr[i][j] = (f < F && k < A) ? p.Uv[((size_t)b * F + f) * A + k] : 0.f;
    }
  }
  // the rest of what the later phases read from memory (attention vectors, c_{t-1} of the pointwise phase): requested
  // now, so that no phase after a barrier starts with a memory round trip
  float wk[2], bk[2], cpre[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int k = lane + 64 * j;
    wk[j] = k < A ? p.w[k] : 0.f; bk[j] = k < A ? p.ab[k] : 0.f;
    const int uu = u0 + tid + 256 * j;
    cpre[j] = (p.c_prev && uu < H) ? p.c_prev[(size_t)b * H + uu] : 0.f;
  }
  for (int k = tid; k < A; k += 256) {
    const float v = p.slab ? sum_strided(p.slab + (size_t)b * WS + W4 + k, zs, p.S) : 0.f;
    swh[k] = v;
    if (p.Wh_out && blockIdx.y == 0) p.Wh_out[(size_t)b * A + k] = v;
  }
  __syncthreads();
  {
    float hk[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int k = lane + 64 * j;
      hk[j] = k < A ? swh[k] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int f = g + 4 * i;
      if (f < F) {
        float s = wk[0] * rn_tanh(hk[0] + uvr[i][0] + bk[0]);
        if (A > 64) s += wk[1] * rn_tanh(hk[1] + uvr[i][1] + bk[1]);
        s = wave_sum(s);
        if (lane == 0) { sa[f] = s; if (p.att_out && blockIdx.y == 0) p.att_out[(size_t)b * F + f] = s; }
      }
    }
  }
  __syncthreads();
  if (live) {
    float c[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) c[j] = 0.f;
#pragma unroll
    for (int f = 0; f < 32; ++f)
      if (f < F) {
        const float a = sa[f];
#pragma unroll
        for (int j = 0; j < 8; ++j) c[j] += a * pv[f].at(j);
      }
    const float invF = 1.0f / (float)F;
    float* dst = spre + g * 512 + lane * 8;
    *reinterpret_cast<f32x4*>(dst) = f32x4{pre[0] + c[0] * invF, pre[1] + c[1] * invF, pre[2] + c[2] * invF, pre[3] + c[3] * invF};
    *reinterpret_cast<f32x4*>(dst + 4) = f32x4{pre[4] + c[4] * invF, pre[5] + c[5] * invF, pre[6] + c[6] * invF, pre[7] + c[7] * invF};
  }
  __syncthreads();
#pragma unroll
  for (int jj = 0; jj < 2; ++jj) {
    const int ul = tid + 256 * jj;
    const int uu = u0 + ul;
    if (uu >= H) break;
    const size_t o = (size_t)b * H + uu;
    const float cprev = cpre[jj];
    float hv, a0, a1, a2, a3;
    if (p.gru) {
      const GruOut r = gru_point(spre[ul], spre[512 + ul], spre[1024 + ul], spre[1536 + ul], cprev);
      hv = r.h; a0 = r.r; a1 = r.z; a2 = r.n; a3 = r.hn;
    } else {
      const LstmOut r = lstm_point(spre[ul], spre[512 + ul], spre[1024 + ul], spre[1536 + ul], cprev);
      hv = r.h; a0 = r.i; a1 = r.f; a2 = r.g; a3 = r.o;
      p.c_out[o] = r.c;
    }
    p.h_out[o] = hv;
    if (p.h_lp) reinterpret_cast<AT*>(p.h_lp)[(size_t)b * p.ld_hlp + uu] = (AT)hv;
    if (p.acts) {
      float* a = p.acts + (size_t)b * W4 + uu;
      a[0] = a0; a[H] = a1; a[2 * H] = a2; a[3 * H] = a3;
    }
  }
  if (p.h_lp && blockIdx.y == 0)
    for (int j = H + tid; j < p.ld_hlp; j += 256) reinterpret_cast<AT*>(p.h_lp)[(size_t)b * p.ld_hlp + j] = (AT)0.f;
}

// =============================================================================================
// decoder recurrent step, backward: one workgroup per (caption, frame chunk), RN_FCH chunks.
//   dh_t = dHs[t] + (dgates_{t+1} | dWh_{t+1}) . [W_hh ; W ; .. ; W] (split-K slabs) -> LSTM pointwise backward
//   da[f] = (1/F) dgates_t . P[b,f,:]   ;   dz = da[f] w (1 - tanh^2)   ;  dWh (per chunk), dUv, dw accumulate
// Every chunk recomputes the (cheap) pointwise backward of the whole row, chunk 0 stores it.  The row
// written is [dgates (4H) | dWh chunk 0 (A) | .. | dWh chunk RN_FCH-1 (A)]: the A operand of the next step's
// GEMM (against the packed [W_hh ; W x RN_FCH]) and of the deferred weight-gradient GEMMs — the partial dWh
// are summed by the GEMM's K loop, deterministically, instead of by atomics.
// dc_carry is double-buffered by step parity because the chunks of one caption run concurrently.
// =============================================================================================
#define RN_FCH 4
struct DecCellBwdArgs {
  int t, B, F, H, A, S;
  int gru;               // 1: GRU (c unused, c_prev = h_{t-1}, the carry holds dh * z instead of dc * f)
  const float* slab;     // [S][B][H] or nullptr (t == T-1)
  const float* dHs;      // [B][H] direct gradient of h_t from the vocabulary projection
  const float* dHs2;     // [B][H] direct gradient of h_t from the reconstructor, or nullptr
  const float* acts; const float* c; const float* c_prev;
  const float* dc_in; float* dc_out; int first;
  void* dGx; int ld_dgx;   // [B][ld_dgx] AT
  const void* P; int ldp; const float* Uv; const float* ab; const float* w;
  const float* Wh;       // [B][A] of step t
  float* dUv;            // [B][F][A] accumulated over t
  float* dwacc;          // [RN_FCH][B][A] accumulated over t
  void* dUv_lp; int ld_dUv; int last;   // at the last executed step (t == 0) also emit the AT copy of dUv
};

template <typename AT>
__global__ __launch_bounds__(256) void dec_cell_bwd_kernel(const DecCellBwdArgs p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* sdg = smem;            // [4H]
  float* sda = sdg + 4 * p.H;   // [F]
  float* spart = sda + p.F;     // [2][G][A] partial sums
  const int b = blockIdx.x, ch = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int H = p.H, A = p.A, F = p.F, W4 = 4 * H;
  const size_t zs = (size_t)p.B * H;
  AT* dgx = reinterpret_cast<AT*>(p.dGx) + (size_t)b * p.ld_dgx;
  const AT* Pb = reinterpret_cast<const AT*>(p.P) + (size_t)b * F * p.ldp;
  const int nf = (F - ch + RN_FCH - 1) / RN_FCH;          // frames of this chunk: f = ch, ch + RN_FCH, ...
  const int G = (A <= 256) ? 256 / A : 1;
  // ---- loads that do not depend on this kernel's own results are issued first (P rows of the wave's frames,
  // Uv / dUv of the thread's (f, k) cells); fast path: 4H <= 2048 (multiple of 8), <= 8 frames per chunk, A <= 256
  const bool fast = ((W4 & 7) == 0) && W4 <= 2048 && nf <= 8 && A <= 256 && nf <= 4 * G;
  Raw8<AT> pr[2][4];
  float uvr[4], duvr[4], whk_pre = 0.f, wk_pre = 0.f;
  const int kk = (A <= 256) ? tid % A : 0, gi = (A <= 256) ? tid / A : 0;
  if (fast) {
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int i = wave + 4 * q;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int n = (lane + 64 * j) * 8;
        if (i < nf && n < W4) pr[q][j].load(Pb + (size_t)(ch + i * RN_FCH) * p.ldp + n); else pr[q][j].zero();
      }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int i = gi + q * G;
      uvr[q] = 0.f; duvr[q] = 0.f;
      if (gi < G && i < nf) {
        const size_t o = ((size_t)b * F + ch + i * RN_FCH) * A + kk;
        uvr[q] = p.Uv[o];
        if (!p.first) duvr[q] = p.dUv[o];
      }
    }
    if (gi < G) { whk_pre = p.Wh[(size_t)b * A + kk] + p.ab[kk]; wk_pre = p.w[kk]; }
  }
  for (int u = tid; u < H; u += 256) {
    const size_t o = (size_t)b * H + u;
    float dh = p.dHs[o];
    if (p.dHs2) dh += p.dHs2[o];
    if (p.slab) dh += sum_strided(p.slab + o, zs, p.S);
    const float* a = p.acts + (size_t)b * W4 + u;
    const float carry = p.first ? 0.f : p.dc_in[o], cpv = p.c_prev ? p.c_prev[o] : 0.f;
    const LstmGrad g = p.gru ? gru_point_bwd(dh + carry, a[0], a[H], a[2 * H], a[3 * H], cpv)
                             : lstm_point_bwd(dh, carry, a[0], a[H], a[2 * H], a[3 * H], p.c[o], cpv);
    sdg[u] = g.di; sdg[H + u] = g.df; sdg[2 * H + u] = g.dg; sdg[3 * H + u] = g.d_o;
    if (ch == 0) {
      dgx[u] = (AT)g.di; dgx[H + u] = (AT)g.df; dgx[2 * H + u] = (AT)g.dg; dgx[3 * H + u] = (AT)g.d_o;
      p.dc_out[o] = g.dc_prev;
    }
  }
  if (ch == 0) for (int j = W4 + RN_FCH * A + tid; j < p.ld_dgx; j += 256) dgx[j] = (AT)0.f;   // pad
  __syncthreads();
  const float invF = 1.0f / (float)F;
  if (fast) {
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int i = wave + 4 * q;
      if (i < nf) {
        float s0 = 0.f, s1 = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int n = (lane + 64 * j) * 8;
          if (n < W4) {
            const f32x4 g0 = *reinterpret_cast<const f32x4*>(sdg + n), g1 = *reinterpret_cast<const f32x4*>(sdg + n + 4);
            s0 += g0[0] * pr[q][j].at(0) + g0[1] * pr[q][j].at(1) + g0[2] * pr[q][j].at(2) + g0[3] * pr[q][j].at(3);
            s1 += g1[0] * pr[q][j].at(4) + g1[1] * pr[q][j].at(5) + g1[2] * pr[q][j].at(6) + g1[3] * pr[q][j].at(7);
          }
        }
        const float s = wave_sum(s0 + s1);
        if (lane == 0) sda[ch + i * RN_FCH] = s * invF;
      }
    }
  } else {
    for (int i = wave; i < nf; i += 4) {
      const int f = ch + i * RN_FCH;
      const AT* pp = Pb + (size_t)f * p.ldp;
      float s = 0.f;
      for (int n = lane; n < W4; n += 64) s += sdg[n] * (float)pp[n];
      s = wave_sum(s);
      if (lane == 0) sda[f] = s * invF;
    }
  }
  __syncthreads();
  // (f, k) plane: thread -> k = tid % A, frame group gi = tid / A (A <= 256), else one thread per k
  if (fast) {
    if (gi < G) {
      const float whk = whk_pre, wk = wk_pre;
      float dwh = 0.f, dw = 0.f;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int i = gi + q * G;
        if (i < nf) {
          const int f = ch + i * RN_FCH;
          const size_t o = ((size_t)b * F + f) * A + kk;
          const float tz = rn_tanh(whk + uvr[q]);
          const float ds = sda[f] * wk * (1.f - tz * tz);
          dw += sda[f] * tz;
          dwh += ds;
          const float nv = duvr[q] + ds;
          p.dUv[o] = nv;
          if (p.last) reinterpret_cast<AT*>(p.dUv_lp)[((size_t)b * F + f) * p.ld_dUv + kk] = (AT)nv;
        }
      }
      spart[gi * A + kk] = dwh;
      spart[(G + gi) * A + kk] = dw;
    }
  } else {
    auto fk = [&](int k2, int g2) {
      const float whk = p.Wh[(size_t)b * A + k2] + p.ab[k2];
      const float wk = p.w[k2];
      float dwh = 0.f, dw = 0.f;
      for (int i = g2; i < nf; i += G) {
        const int f = ch + i * RN_FCH;
        const size_t o = ((size_t)b * F + f) * A + k2;
        const float tz = rn_tanh(whk + p.Uv[o]);
        const float ds = sda[f] * wk * (1.f - tz * tz);
        dw += sda[f] * tz;
        dwh += ds;
        const float nv = p.first ? ds : p.dUv[o] + ds;
        p.dUv[o] = nv;
        if (p.last) reinterpret_cast<AT*>(p.dUv_lp)[((size_t)b * F + f) * p.ld_dUv + k2] = (AT)nv;
      }
      spart[g2 * A + k2] = dwh;
      spart[(G + g2) * A + k2] = dw;
    };
    if (A <= 256) {
      if (tid < G * A) fk(tid % A, tid / A);
    } else {
      for (int k2 = tid; k2 < A; k2 += 256) fk(k2, 0);
    }
  }
  __syncthreads();
  for (int k2 = tid; k2 < A; k2 += 256) {
    float a = 0.f, c = 0.f;
    for (int j = 0; j < G; ++j) { a += spart[j * A + k2]; c += spart[(G + j) * A + k2]; }
    dgx[W4 + ch * A + k2] = (AT)a;
    const size_t o2 = ((size_t)ch * p.B + b) * A + k2;
    p.dwacc[o2] = p.first ? c : p.dwacc[o2] + c;
  }
  if (p.last)   // zero padding of the dUv operand copy
    for (int i = wave; i < nf; i += 4) {
      const int f = ch + i * RN_FCH;
      for (int j = A + lane; j < p.ld_dUv; j += 64) reinterpret_cast<AT*>(p.dUv_lp)[((size_t)b * F + f) * p.ld_dUv + j] = (AT)0.f;
    }
}

// ctx[t,b,d] = (1/F) sum_f att[t,b,f] enc[b,f,d] for all t at once (the attended features of every step,
// needed only by the deferred dW_ih[:, E:] = dgates^T . ctx GEMM).  grid (B, ceil(ld/256)); T <= 32.
template <typename AT>
__global__ __launch_bounds__(256) void ctx_all_kernel(const float* __restrict__ att, const float* __restrict__ enc,
                                                      AT* __restrict__ ctx, int ld, int T, int B, int F, int D) {
  extern __shared__ __attribute__((aligned(16))) float smem[];   // [32][F], zero padded beyond T
  const int b = blockIdx.x, d = blockIdx.y * 256 + threadIdx.x;
  for (int i = threadIdx.x; i < 32 * F; i += 256) {
    const int t = i / F, f = i % F;
    smem[i] = t < T ? att[((size_t)t * B + b) * F + f] : 0.f;
  }
  __syncthreads();
  if (d >= ld) return;
  float acc[32];
#pragma unroll
  for (int t = 0; t < 32; ++t) acc[t] = 0.f;
  if (d < D)
    for (int f = 0; f < F; ++f) {
      const float e = enc[((size_t)b * F + f) * D + d];
#pragma unroll
      for (int t = 0; t < 32; ++t) acc[t] += smem[t * F + f] * e;
    }
  const float invF = 1.0f / (float)F;
#pragma unroll
  for (int t = 0; t < 32; ++t)
    if (t < T) ctx[((size_t)t * B + b) * ld + d] = (AT)(acc[t] * invF);
}
// general-T fallback (caption_max_len + 1 > 32)
template <typename AT>
__global__ __launch_bounds__(256) void ctx_all_slow_kernel(const float* __restrict__ att, const float* __restrict__ enc,
                                                           AT* __restrict__ ctx, int ld, int T, int B, int F, int D) {
  const int b = blockIdx.x, d = blockIdx.y * 256 + threadIdx.x;
  if (d >= ld) return;
  for (int t = 0; t < T; ++t) {
    float s = 0.f;
    if (d < D) for (int f = 0; f < F; ++f) s += att[((size_t)t * B + b) * F + f] * enc[((size_t)b * F + f) * D + d];
    ctx[((size_t)t * B + b) * ld + d] = (AT)(s / (float)F);
  }
}

// =============================================================================================
// generic element-wise LSTM step (reconstructors: hidden size R)
// =============================================================================================
struct LstmPwArgs {
  int B, Hd, S;
  int gru;                                               // 1: GRU (c_prev = h_prev, c_out unused)
  const float* slab; size_t slab_stride; int slab_ld;   // [S] x [B][slab_ld], gate columns at [0, 4Hd)
  const float* X; int x_ld;                              // optional pre-computed input part [B][x_ld]
  const float* b1; const float* b2;                      // optional bias vectors [4Hd]
  const float* c_prev;                                   // [B][Hd] or nullptr
  float* h_out; int h_ld;                                // [B][h_ld] fp32
  void* h_lp; int hlp_ld; int hlp_pad_from;              // AT copy [B][hlp_ld]; pad columns [hlp_pad_from, hlp_ld) zeroed
  void* h_lp2; int hlp2_ld;                              // optional second AT destination (next step's GEMM input row)
  float* c_out; float* acts;
};
template <typename AT>
__global__ __launch_bounds__(256) void lstm_pw_kernel(const LstmPwArgs p) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= p.B * p.Hd) return;
  const int b = i / p.Hd, u = i % p.Hd, Hd = p.Hd;
  float g[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int col = q * Hd + u;
    float v = p.X ? p.X[(size_t)b * p.x_ld + col] : 0.f;
    if (p.b1) v += p.b1[col];
    if (p.b2) v += p.b2[col];
    if (p.S) v += sum_strided(p.slab + (size_t)b * p.slab_ld + col, p.slab_stride, p.S);
    g[q] = v;
  }
  const float cp = p.c_prev ? p.c_prev[(size_t)b * Hd + u] : 0.f;
  float hv, a0, a1, a2, a3;
  if (p.gru) {
    const GruOut r = gru_point(g[0], g[1], g[2], g[3], cp);
    hv = r.h; a0 = r.r; a1 = r.z; a2 = r.n; a3 = r.hn;
  } else {
    const LstmOut r = lstm_point(g[0], g[1], g[2], g[3], cp);
    hv = r.h; a0 = r.i; a1 = r.f; a2 = r.g; a3 = r.o;
    p.c_out[(size_t)b * Hd + u] = r.c;
  }
  p.h_out[(size_t)b * p.h_ld + u] = hv;
  if (p.h_lp) {
    AT* d = reinterpret_cast<AT*>(p.h_lp) + (size_t)b * p.hlp_ld;
    d[u] = (AT)hv;
    if (u < p.hlp_ld - p.hlp_pad_from) d[p.hlp_pad_from + u] = (AT)0.f;
  }
  if (p.h_lp2) reinterpret_cast<AT*>(p.h_lp2)[(size_t)b * p.hlp2_ld + u] = (AT)hv;
  float* a = p.acts + (size_t)b * 4 * Hd + u;
  a[0] = a0; a[Hd] = a1; a[2 * Hd] = a2; a[3 * Hd] = a3;
}

struct LstmBwdArgs {
  int B, Hd, S;
  int gru;
  const float* dh_direct; int dhd_ld; float dh_scale;    // optional [B][dhd_ld]
  const float* slab; size_t slab_stride; int slab_ld; int slab_col0;   // recurrent part: sum_z slab[z][b][col0+u]
  const float* slab2; size_t slab2_stride; int S2;                     // optional second product [S2][B][Hd]
  const float* acts; const float* c; const float* c_prev;
  float* dc_carry; int first;
  void* dG; int ld_dg;                                    // [B][ld_dg] AT, gate columns [0,4Hd), zero padded
};
template <typename AT>
__global__ __launch_bounds__(256) void lstm_bwd_kernel(const LstmBwdArgs p) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= p.B * p.Hd) return;
  const int b = i / p.Hd, u = i % p.Hd, Hd = p.Hd;
  float dh = p.dh_direct ? p.dh_scale * p.dh_direct[(size_t)b * p.dhd_ld + u] : 0.f;
  if (p.S) dh += sum_strided(p.slab + (size_t)b * p.slab_ld + p.slab_col0 + u, p.slab_stride, p.S);
  if (p.S2) dh += sum_strided(p.slab2 + (size_t)b * Hd + u, p.slab2_stride, p.S2);
  const size_t o = (size_t)b * Hd + u;
  const float* a = p.acts + (size_t)b * 4 * Hd + u;
  const float carry = p.first ? 0.f : p.dc_carry[o], cpv = p.c_prev ? p.c_prev[o] : 0.f;
  const LstmGrad g = p.gru ? gru_point_bwd(dh + carry, a[0], a[Hd], a[2 * Hd], a[3 * Hd], cpv)
                           : lstm_point_bwd(dh, carry, a[0], a[Hd], a[2 * Hd], a[3 * Hd], p.c[o], cpv);
  AT* dg = reinterpret_cast<AT*>(p.dG) + (size_t)b * p.ld_dg;
  dg[u] = (AT)g.di; dg[Hd + u] = (AT)g.df; dg[2 * Hd + u] = (AT)g.dg; dg[3 * Hd + u] = (AT)g.d_o;
  if (u < p.ld_dg - 4 * Hd) dg[4 * Hd + u] = (AT)0.f;
  p.dc_carry[o] = g.dc_prev;
}

// =============================================================================================
// masked cross-entropy with logits dropout (decoder.py:69, train.py:54-56,68) — forward + dlogits
//   rowloss[t,b] = [tgt>0] * cw[t] * CE(drop(logits[t,b,:]), tgt) ;
//   dlog[t,b,:]  = [tgt>0] * cw[t] * (softmax - onehot) * dropmask     (AT operand copy, zero padded to ld)
// =============================================================================================
template <typename AT>
__global__ __launch_bounds__(256) void ce_kernel(const float* __restrict__ logits, const int64_t* __restrict__ targets,
                                                 const float* __restrict__ cw, float* __restrict__ rowloss,
                                                 AT* __restrict__ dlog, int ld, int B, int V, DropDesc dd) {
  __shared__ float sm[4];
  const int row = blockIdx.x, t = row / B, b = row % B, tid = threadIdx.x;
  const float* x = logits + (size_t)row * V;
  AT* dx = dlog + (size_t)row * ld;
  const long tgt = targets[(size_t)t * B + b];
  if (tgt <= 0 || tgt >= V) {
    if (tid == 0) rowloss[row] = 0.f;
    for (int v = tid; v < ld; v += 256) dx[v] = (AT)0.f;
    return;
  }
  const uint32_t key = drop_key(dd);
  float mx = -3.0e38f;
  for (int v = tid; v < V; v += 256) mx = fmaxf(mx, x[v] * drop_at(dd, key, t, b, V, v));
  mx = block_max256(mx, sm);
  float s = 0.f;
  for (int v = tid; v < V; v += 256) s += expf(x[v] * drop_at(dd, key, t, b, V, v) - mx);
  s = block_sum256(s, sm);
  const float lse = mx + logf(s);
  const float wgt = cw[t];
  if (tid == 0) rowloss[row] = wgt * (lse - x[tgt] * drop_at(dd, key, t, b, V, (int)tgt));
  for (int v = tid; v < ld; v += 256) {
    float gv = 0.f;
    if (v < V) {
      const float m = drop_at(dd, key, t, b, V, v);
      gv = wgt * (expf(x[v] * m - lse) - (v == tgt ? 1.f : 0.f)) * m;
    }
    dx[v] = (AT)gv;
  }
}
// logits *= dropmask (step API, train mode)
__global__ void logits_drop_kernel(float* __restrict__ logits, int B, int V, DropDesc dd, int t) {
  const uint32_t key = drop_key(dd);
  const size_t total = (size_t)B * V;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int b = (int)(i / V), v = (int)(i % V);
    logits[i] *= drop_at(dd, key, t, b, V, v);
  }
}

// =============================================================================================
// global reconstructor helpers (global_reconstructor.py:33-41, train.py:96-102)
// =============================================================================================
// out[b,c] = scale * sum_t X[t,b,c]  (+ AT operand copy with zero padding)
template <typename AT>
__global__ void mean_over_t_kernel(const float* __restrict__ X, int T, int Bn, int Cn, float scale, float* __restrict__ out,
                                   AT* __restrict__ out_lp, int ld_lp) {
  const int ldx = out_lp ? ld_lp : Cn;
  const size_t n = (size_t)Bn * ldx;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const int b = (int)(i / ldx), c = (int)(i % ldx);
    float s = 0.f;
    if (c < Cn) {
      for (int t = 0; t < T; ++t) s += X[((size_t)t * Bn + b) * Cn + c];
      s *= scale;
      out[(size_t)b * Cn + c] = s;
    }
    if (out_lp) out_lp[i] = (AT)s;
  }
}
// The global reconstructor's LSTM input x_t = [h_t ; drop_t(mp)] (global_reconstructor.py:38-41) as one GEMM operand:
//   xcat[t,b, 0:H) = h_t (copy of the decoder's operand copy),  xcat[t,b, H:2H) = mp[b] * dropmask(t,b,.),  zero padded
template <typename AT>
__global__ void xcat_global_kernel(const AT* __restrict__ hs, int ld_hs, const float* __restrict__ mp, AT* __restrict__ xcat, int ld,
                                   int T, int B, int H, DropDesc dd) {
  const uint32_t key = drop_key(dd);
  const size_t total = (size_t)T * B * ld;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % ld), b = (int)((i / ld) % B), t = (int)(i / ((size_t)ld * B));
    AT v = (AT)0.f;
    if (c < H) v = hs[((size_t)t * B + b) * ld_hs + c];
    else if (c < 2 * H) v = (AT)(mp[(size_t)b * H + (c - H)] * drop_at(dd, key, t, b, H, c - H));
    xcat[i] = v;
  }
}
// dmp[b,h] = sum_t dmpd[t,b,h] * dropmask(t,b,h)
__global__ void bcast_drop_bwd_kernel(const float* __restrict__ dmpd, float* __restrict__ dmp, int T, int B, int H, DropDesc dd) {
  const uint32_t key = drop_key(dd);
  const size_t n = (size_t)B * H;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const int h = (int)(i % H), b = (int)(i / H);
    float s = 0.f;
    for (int t = 0; t < T; ++t) s += dmpd[(size_t)t * n + i] * drop_at(dd, key, t, b, H, h);
    dmp[i] = s;
  }
}
// Y[t*n + i] (+)= c * x[i]
__global__ void add_bcast_kernel(float* __restrict__ Y, const float* __restrict__ x, int T, size_t n, float c, int accumulate) {
  const size_t total = (size_t)T * n;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const float v = c * x[i % n];
    Y[i] = accumulate ? Y[i] + v : v;
  }
}
// encmean[b,d] = (1/F) sum_f enc[b,f,d]
__global__ void mean_over_f_kernel(const float* __restrict__ enc, int B, int F, int D, float* __restrict__ out) {
  const size_t n = (size_t)B * D;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const int b = (int)(i / D), d = (int)(i % D);
    float s = 0.f;
    for (int f = 0; f < F; ++f) s += enc[((size_t)b * F + f) * D + d];
    out[i] = s / (float)F;
  }
}
// diff = out - ref(b, s, :);  partial[block] = sum diff^2 ; out <- gcoef * diff   (d loss / d out, fp32)
//   ref indexing: ref[b*ref_bstride + s*ref_sstride + r], out rows ordered (s, b)
__global__ __launch_bounds__(256) void mse_kernel(float* __restrict__ out, const float* __restrict__ ref, int Sn, int B,
                                                  int R, size_t ref_bstride, size_t ref_sstride,
                                                  float gcoef, float* __restrict__ partial) {
  __shared__ float sm[4];
  const size_t total = (size_t)Sn * B * R;
  float acc = 0.f;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int r = (int)(i % R), b = (int)((i / R) % B), s = (int)(i / ((size_t)R * B));
    const float d = out[i] - ref[(size_t)b * ref_bstride + (size_t)s * ref_sstride + r];
    acc += d * d;
    out[i] = gcoef * d;
  }
  acc = block_sum256(acc, sm);
  if (threadIdx.x == 0) partial[blockIdx.x] = acc;
}

// =============================================================================================
// local reconstructor attention (local_reconstructor.py:38-50), one workgroup per caption
//   beta[t'] = w . tanh(W hr + U h_t' + b)  (no softmax);  x = drop((1/T) sum_t' beta[t'] h_t')
// =============================================================================================
struct LocAttnArgs {
  int s, B, T, H, A, S;
  const float* slab;      // [S][B][A] split-K partials of hr_{s-1} . W_r^T  (nullptr at s = 0: zeros)
  const float* Ud;        // [T][B][A]
  const float* ab; const float* w;
  const float* Hs;        // [T][B][H] decoder hidden states
  float* Whr_out;         // [B][A]
  float* beta_out;        // [B][T]
  void* xcat; int xcat_ld;   // AT row [x (H) | hr (R) | pad]: x -> [0,H)
  DropDesc dd;
};
// grid (B, ceil(H / 256)): every workgroup recomputes the (cheap) scores beta, then each thread owns one column h
template <typename AT>
__global__ __launch_bounds__(256) void loc_attn_fwd_kernel(const LocAttnArgs p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* swh = smem;            // [A]
  float* sbeta = swh + p.A;     // [T]
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int h = blockIdx.y * 256 + tid;
  const size_t zs = (size_t)p.B * p.A;
  // Hs[t', b, h] for this thread's column: issued before anything that depends on the scores (T <= 32 fast path)
  float hv[32];
  const bool fastT = p.T <= 32;
  if (fastT) {
#pragma unroll
    for (int t = 0; t < 32; ++t) hv[t] = (t < p.T && h < p.H) ? p.Hs[((size_t)t * p.B + b) * p.H + h] : 0.f;
  }
  // Ud[t, b, k] of this wave's time steps (t = wave, wave + 4, ...; lane -> k = lane, lane + 64), also issued up front:
  // the score loop below then has no load in it (A <= 128, T <= 32 fast path)
  const bool fastA = fastT && p.A <= 128;
  float udr[8][2], wk[2] = {0.f, 0.f}, bk[2] = {0.f, 0.f};
  if (fastA) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int k = lane + 64 * j;
      wk[j] = k < p.A ? p.w[k] : 0.f; bk[j] = k < p.A ? p.ab[k] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int t = wave + 4 * i;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int k = lane + 64 * j;
        udr[i][j] = (t < p.T && k < p.A) ? p.Ud[((size_t)t * p.B + b) * p.A + k] : 0.f;
      }
    }
  }
  for (int k = tid; k < p.A; k += 256) {
    const float v = p.slab ? sum_strided(p.slab + (size_t)b * p.A + k, zs, p.S) : 0.f;
    swh[k] = v;
    if (blockIdx.y == 0) p.Whr_out[(size_t)b * p.A + k] = v;
  }
  __syncthreads();
  if (fastA) {
    float hk[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int k = lane + 64 * j;
      hk[j] = k < p.A ? swh[k] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int t = wave + 4 * i;
      if (t < p.T) {
        float s = wk[0] * rn_tanh(hk[0] + udr[i][0] + bk[0]);
        if (p.A > 64) s += wk[1] * rn_tanh(hk[1] + udr[i][1] + bk[1]);
        s = wave_sum(s);
        if (lane == 0) { sbeta[t] = s; if (blockIdx.y == 0) p.beta_out[(size_t)b * p.T + t] = s; }
      }
    }
  } else {
    for (int t = wave; t < p.T; t += 4) {
      const float* ud = p.Ud + ((size_t)t * p.B + b) * p.A;
      float s = 0.f;
      for (int k = lane; k < p.A; k += 64) s += p.w[k] * rn_tanh(swh[k] + ud[k] + p.ab[k]);
      s = wave_sum(s);
      if (lane == 0) { sbeta[t] = s; if (blockIdx.y == 0) p.beta_out[(size_t)b * p.T + t] = s; }
    }
  }
  __syncthreads();
  if (h >= p.H) return;
  const uint32_t key = drop_key(p.dd);
  const float invT = 1.0f / (float)p.T;
  AT* xr = reinterpret_cast<AT*>(p.xcat) + (size_t)b * p.xcat_ld;
  float s0 = 0.f, s1 = 0.f;
  if (fastT) {
#pragma unroll
    for (int t = 0; t < 32; t += 2) {
      if (t < p.T) s0 += sbeta[t] * hv[t];
      if (t + 1 < p.T) s1 += sbeta[t + 1] * hv[t + 1];
    }
  } else {
    for (int t = 0; t < p.T; ++t) s0 += sbeta[t] * p.Hs[((size_t)t * p.B + b) * p.H + h];
  }
  xr[h] = (AT)((s0 + s1) * invT * drop_at(p.dd, key, p.s, b, p.H, h));
}

// Attention backward of reconstructor step s, one workgroup per (caption, chunk of decoder steps t'), RN_TCH
// chunks (t' = ch, ch + RN_TCH, ...).  dx_s comes from the x columns of dGr_s . [W_ih | W_hh] (split-K slabs);
// outputs: dHs[t'] += (1/T) beta dx and dUd[t'] += dz for the chunk's own t' (no conflicts between chunks), the
// chunk's partial dWhr_s written side by side [chunk 0 | .. | chunk RN_TCH-1] (summed by the K loop of the next
// GEMM against [W_r ; .. ; W_r], like the decoder's dWh), and the dw accumulator per chunk.
#define RN_TCH 4
struct LocBwdArgs {
  int s, B, T, H, R, A, S;
  const float* slab;      // [S][B][H+R]
  const float* Hs; const float* Ud; const float* ab; const float* w;
  const float* Whr;       // [B][A] of step s
  const float* beta;      // [B][T] of step s
  float* dHs;             // [T][B][H] accumulated over s
  float* dUd;             // [T][B][A] accumulated over s
  void* dWhr; int ld_dwhr;   // AT [B][ld] of step s: RN_TCH partial blocks of A columns
  float* dwacc;           // [RN_TCH][B][A]
  int first;
  void* dUd_lp; int ld_dUd; int last;   // at s == 0 also emit the AT copy of dUd
  DropDesc dd;
};
template <typename AT>
__global__ __launch_bounds__(256) void loc_attn_bwd_kernel(const LocBwdArgs p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* sdx = smem;            // [H]
  float* sdb = sdx + p.H;       // [T]
  float* sbt = sdb + p.T;       // [T] beta / T
  float* spart = sbt + p.T;     // [2][G][A]
  const int b = blockIdx.x, ch = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int H = p.H, R = p.R, A = p.A, T = p.T;
  const int W2 = H + R;
  const size_t zs = (size_t)p.B * W2;
  const uint32_t key = drop_key(p.dd);
  const float invT = 1.0f / (float)T;
  const int nt = (T - ch + RN_TCH - 1) / RN_TCH;        // decoder steps of this chunk
  const size_t st = (size_t)p.B * H * RN_TCH;
  const int G = (A <= 256) ? 256 / A : 1;
  // ---- fast path (T <= 32, H <= 512, A <= 128): every global load of the kernel that does not depend on its own
  // results is issued here, before the first barrier — the kernel is one link of a dependent chain and otherwise pays
  // one memory latency per phase (hidden-state rows of the wave's dot products, the chunk's old dHs values, Ud / dUd of
  // the thread's (t', k) cells)
  const bool fast = T <= 32 && H <= 512 && A <= 128;
  float hsr[2][8], dhv[2][8], udv[4], dudv[4], whk_pre = 0.f, wk_pre = 0.f;
  const int kk = tid % A, gi = tid / A;
  if (fast) {
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int i = wave + 4 * q;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int h = lane + 64 * j;
        hsr[q][j] = (i < nt && h < H) ? p.Hs[((size_t)(ch + i * RN_TCH) * p.B + b) * H + h] : 0.f;
      }
    }
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int h = tid + 256 * q;
#pragma unroll
      for (int i = 0; i < 8; ++i)
        dhv[q][i] = (!p.first && h < H && i < nt) ? p.dHs[((size_t)ch * p.B + b) * H + h + (size_t)i * st] : 0.f;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int i = gi + q * G;
      udv[q] = 0.f; dudv[q] = 0.f;
      if (gi < G && i < nt) {
        const size_t o = ((size_t)(ch + i * RN_TCH) * p.B + b) * A + kk;
        udv[q] = p.Ud[o];
        if (!p.first) dudv[q] = p.dUd[o];
      }
    }
    if (gi < G) { whk_pre = p.Whr[(size_t)b * A + kk] + p.ab[kk]; wk_pre = p.w[kk]; }
  }
  for (int j = tid; j < H; j += 256)
    sdx[j] = sum_strided(p.slab + (size_t)b * W2 + j, zs, p.S) * drop_at(p.dd, key, p.s, b, H, j);
  for (int t = tid; t < T; t += 256) sbt[t] = p.beta[(size_t)b * T + t] * invT;
  __syncthreads();
  if (fast) {
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int i = wave + 4 * q;
      if (i < nt) {
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) { const int h = lane + 64 * j; if (h < H) s += sdx[h] * hsr[q][j]; }
        s = wave_sum(s);
        if (lane == 0) sdb[ch + i * RN_TCH] = s * invT;
      }
    }
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int h = tid + 256 * q;
      if (h < H) {
        const float dx = sdx[h];
#pragma unroll
        for (int i = 0; i < 8; ++i)
          if (i < nt) p.dHs[((size_t)ch * p.B + b) * H + h + (size_t)i * st] = dhv[q][i] + sbt[ch + i * RN_TCH] * dx;
      }
    }
    __syncthreads();
    if (gi < G) {
      const float whk = whk_pre, wk = wk_pre;
      float dwh = 0.f, dw = 0.f;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int i = gi + q * G;
        if (i < nt) {
          const int t = ch + i * RN_TCH;
          const size_t o = ((size_t)t * p.B + b) * A + kk;
          const float tz = rn_tanh(whk + udv[q]);
          const float dz = sdb[t] * wk * (1.f - tz * tz);
          dw += sdb[t] * tz;
          dwh += dz;
          const float nv = dudv[q] + dz;
          p.dUd[o] = nv;
          if (p.last) reinterpret_cast<AT*>(p.dUd_lp)[((size_t)t * p.B + b) * p.ld_dUd + kk] = (AT)nv;
        }
      }
      spart[gi * A + kk] = dwh;
      spart[(G + gi) * A + kk] = dw;
    }
  } else {
  for (int i = wave; i < nt; i += 4) {
    const int t = ch + i * RN_TCH;
    const float* hs = p.Hs + ((size_t)t * p.B + b) * H;
    float s = 0.f;
    for (int h = lane; h < H; h += 64) s += sdx[h] * hs[h];
    s = wave_sum(s);
    if (lane == 0) sdb[t] = s * invT;
  }
  // dHs[t',b,:] += (1/T) beta[t'] dx for the chunk's t' (independent read-modify-writes, four in flight)
  for (int h = tid; h < H; h += 256) {
    const float dx = sdx[h];
    float* d0 = p.dHs + ((size_t)ch * p.B + b) * H + h;
    int i = 0;
    for (; i + 4 <= nt; i += 4) {
      float* d = d0 + (size_t)i * st;
      float v0 = 0.f, v1 = 0.f, v2 = 0.f, v3 = 0.f;
      if (!p.first) { v0 = d[0]; v1 = d[st]; v2 = d[2 * st]; v3 = d[3 * st]; }
      const int t = ch + i * RN_TCH;
      d[0] = v0 + sbt[t] * dx; d[st] = v1 + sbt[t + RN_TCH] * dx; d[2 * st] = v2 + sbt[t + 2 * RN_TCH] * dx;
      d[3 * st] = v3 + sbt[t + 3 * RN_TCH] * dx;
    }
    for (; i < nt; ++i) {
      float* d = d0 + (size_t)i * st;
      d[0] = (p.first ? 0.f : d[0]) + sbt[ch + i * RN_TCH] * dx;
    }
  }
  __syncthreads();
  // (t', k) plane: thread -> k = tid % A, group gi = tid / A
  auto tk = [&](int k2, int g2) {
    const float whk = p.Whr[(size_t)b * A + k2] + p.ab[k2];
    const float wk = p.w[k2];
    float dwh = 0.f, dw = 0.f;
    for (int i = g2; i < nt; i += G) {
      const int t = ch + i * RN_TCH;
      const size_t o = ((size_t)t * p.B + b) * A + k2;
      const float tz = rn_tanh(whk + p.Ud[o]);
      const float dz = sdb[t] * wk * (1.f - tz * tz);
      dw += sdb[t] * tz;
      dwh += dz;
      const float nv = p.first ? dz : p.dUd[o] + dz;
      p.dUd[o] = nv;
      if (p.last) reinterpret_cast<AT*>(p.dUd_lp)[((size_t)t * p.B + b) * p.ld_dUd + k2] = (AT)nv;
    }
    spart[g2 * A + k2] = dwh;
    spart[(G + g2) * A + k2] = dw;
  };
  if (A <= 256) { if (tid < G * A) tk(tid % A, tid / A); }
  else for (int k2 = tid; k2 < A; k2 += 256) tk(k2, 0);
  }   // !fast
  __syncthreads();
  AT* dwr = reinterpret_cast<AT*>(p.dWhr) + (size_t)b * p.ld_dwhr;
  for (int k2 = tid; k2 < A; k2 += 256) {
    float a = 0.f, c = 0.f;
    for (int j = 0; j < G; ++j) { a += spart[j * A + k2]; c += spart[(G + j) * A + k2]; }
    dwr[ch * A + k2] = (AT)a;
    const size_t o2 = ((size_t)ch * p.B + b) * A + k2;
    p.dwacc[o2] = p.first ? c : p.dwacc[o2] + c;
  }
  if (ch == 0) for (int j = RN_TCH * A + tid; j < p.ld_dwhr; j += 256) dwr[j] = (AT)0.f;
  if (p.last)
    for (int i = 0; i < nt; ++i) {
      const int t = ch + i * RN_TCH;
      for (int j = A + tid; j < p.ld_dUd; j += 256) reinterpret_cast<AT*>(p.dUd_lp)[((size_t)t * p.B + b) * p.ld_dUd + j] = (AT)0.f;
    }
}

// =============================================================================================
// inference search on the device (eval.py:19-120): the per-sample Python loops of the reference become kernels
// =============================================================================================
// out[row] = argmax_v x[row, v]  (lowest index among equal maxima), one workgroup per row
__global__ __launch_bounds__(256) void argmax_rows_kernel(const float* __restrict__ x, int ld, int cols, int64_t* __restrict__ out) {
  __shared__ float sv[256]; __shared__ int si[256];
  const int row = blockIdx.x, tid = threadIdx.x;
  float best = -3.0e38f; int bi = 0x7fffffff;
  for (int v = tid; v < cols; v += 256) {
    const float y = x[(size_t)row * ld + v];
    if (y > best || (y == best && v < bi)) { best = y; bi = v; }
  }
  sv[tid] = best; si[tid] = bi;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if (tid < w) {
      const float y = sv[tid + w]; const int j = si[tid + w];
      if (y > sv[tid] || (y == sv[tid] && j < si[tid])) { sv[tid] = y; si[tid] = j; }
    }
    __syncthreads();
  }
  if (tid == 0) out[row] = si[0];
}
// record the step's tokens; the reference stops after the first step whose tokens are all <PAD> (eval.py:30,116)
__global__ void search_stop_kernel(const int64_t* __restrict__ tokens, int n, int t, int32_t* __restrict__ n_steps) {
  __shared__ int any;
  if (threadIdx.x == 0) any = 0;
  __syncthreads();
  int a = 0;
  for (int i = threadIdx.x; i < n; i += blockDim.x) a |= (tokens[i] != 0);
  if (a) atomicOr(&any, 1);
  __syncthreads();
  if (threadIdx.x == 0 && !any && *n_steps == 0) *n_steps = t + 1;
}
__global__ void search_finish_kernel(int32_t* n_steps, int tm) { if (*n_steps == 0) *n_steps = tm; }
// scores[b, i*V + v] = log(sigmoid(logits_i[b, v])) + cum[i, b] / len(i, b)^0.7   (eval.py:51-62)
//   len = position of the last <EOS> in hypothesis i of caption b (+1), or t + 1 when it has none
__global__ __launch_bounds__(256) void beam_score_kernel(const float* __restrict__ logits, const float* __restrict__ cum,
                                                         const int32_t* __restrict__ last_eos, float* __restrict__ scores,
                                                         int B, int V, int nb, int i, int t) {
  const int b = blockIdx.x;
  const int le = last_eos[i * B + b];
  const double len = le >= 0 ? (double)(le + 1) : (double)(t + 1);
  const float norm = cum[i * B + b] / (float)pow(len, 0.7);
  const float* x = logits + (size_t)b * V;
  float* o = scores + (size_t)b * nb * V + (size_t)i * V;
  for (int v = threadIdx.x; v < V; v += 256) o[v] = logf(1.0f / (1.0f + expf(-x[v]))) + norm;
}
// top-k (k <= 8) of each row of scores [B][n], descending, lowest index first among equals; destroys scores
__global__ __launch_bounds__(256) void topk_rows_kernel(float* __restrict__ scores, int n, int k, float* __restrict__ vals,
                                                        int32_t* __restrict__ idx) {
  __shared__ float sv[256]; __shared__ int si[256];
  const int b = blockIdx.x, tid = threadIdx.x;
  float* x = scores + (size_t)b * n;
  for (int j = 0; j < k; ++j) {
    float best = -INFINITY; int bi = 0x7fffffff;
    for (int v = tid; v < n; v += 256) {
      const float y = x[v];
      if (y > best || (y == best && v < bi)) { best = y; bi = v; }
    }
    sv[tid] = best; si[tid] = bi;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
      if (tid < w) {
        const float y = sv[tid + w]; const int q = si[tid + w];
        if (y > sv[tid] || (y == sv[tid] && q < si[tid])) { sv[tid] = y; si[tid] = q; }
      }
      __syncthreads();
    }
    if (tid == 0) {
      vals[b * k + j] = sv[0]; idx[b * k + j] = si[0];
      if (si[0] < n) x[si[0]] = -INFINITY;
    }
    __syncthreads();
  }
}
// regather the hypotheses: new beam k of caption b continues old beam src = idx / V with token idx % V (eval.py:66-114)
struct BeamUpdArgs {
  int B, H, V, Tm, bw, t;
  const float* vals; const int32_t* idx;                 // [B][bw]
  const float* h_next; const float* c_next;              // [nb_old][B][H] states after this step
  const int32_t* last_eos_old; const int64_t* hist_old;  // [nb_old][B], [nb_old][B][Tm]
  float* h_new; float* c_new; float* cum_new; int32_t* last_eos_new; int64_t* hist_new; int64_t* tok_new;
  const int32_t* n_steps;                                // != 0: the search already stopped (eval.py:116)
};
__global__ __launch_bounds__(128) void beam_update_kernel(const BeamUpdArgs p) {
  const int k = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
  int flat = p.idx[b * p.bw + k];
  // The reference leaves its loop at the first step whose tokens are all <PAD>; the device loop has a fixed trip
  // count, so from then on the hypotheses are carried over unchanged (same beam order, <PAD> appended).
  if (*p.n_steps != 0) flat = k * p.V;
  const int src = flat / p.V, tok = flat % p.V;
  const size_t so = ((size_t)src * p.B + b), dn = ((size_t)k * p.B + b);
  for (int j = tid; j < p.H; j += 128) {
    p.h_new[dn * p.H + j] = p.h_next[so * p.H + j];
    p.c_new[dn * p.H + j] = p.c_next[so * p.H + j];
  }
  for (int j = tid; j < p.Tm; j += 128) p.hist_new[dn * p.Tm + j] = j < p.t ? p.hist_old[so * p.Tm + j] : (j == p.t ? (int64_t)tok : 0);
  if (tid == 0) {
    p.cum_new[dn] = p.vals[b * p.bw + k];
    p.last_eos_new[dn] = tok == 2 ? p.t : p.last_eos_old[so];
    p.tok_new[dn] = tok;
  }
}
// best[t][b] = hist[beam 0][b][t]
__global__ void beam_best_kernel(const int64_t* __restrict__ hist, int64_t* __restrict__ best, int B, int Tm) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * Tm) return;
  const int t = i / B, b = i % B;
  best[i] = hist[(size_t)b * Tm + t];
}

// =============================================================================================
// norms, clipping and multi-tensor Adam (train.py:69,103,129,149,186,270-273)
// =============================================================================================
struct TensorDesc { float* p; float* g; float* m; float* v; float* vmax; int n; int chunk0; int nchunks; int pad; };
// Where the packed operand image(s) of a parameter tensor live: element (r, c) of a [rows][cols] tensor goes to
// dst[r * ld + (c - c0)] for every destination whose column window [c0, c0 + nc) contains c.  The Adam kernel
// writes them directly, so the weights are re-packed (bf16) in the same pass that updates them.
// Rows: only source rows [r0, r0 + nr) are written, to destination row (r - r0) (dst is pre-offset) — the GRU's
// 3-block weights land in the 4-block packed layout this way.
// mode 1 (gate interleave, rec_step.hpp): r = gate * nr + u goes to destination row (u / 8) * 32 + gate * 8 + u % 8.
struct PackDst { void* dst; int ld; int c0; int nc; int r0; int nr; int mode; int pad; };
struct PackDesc { int ndst; int cols; PackDst d[6]; };
#define RN_CHUNK 8192

// partial[chunk] = sum over the chunk of p^2 (mode 0) or (g + coef * p / ||p||)^2 (mode 1)
__global__ __launch_bounds__(256) void sumsq_chunk_kernel(const TensorDesc* __restrict__ tab, const int2* __restrict__ chunks,
                                                          int mode, const float* __restrict__ pnorm, float coef,
                                                          float* __restrict__ partial) {
  __shared__ float sm[4];
  const int2 ch = chunks[blockIdx.x];
  const TensorDesc td = tab[ch.x];
  const int end = min(td.n, ch.y + RN_CHUNK);
  float k = 0.f;
  if (mode == 1) { const float nrm = pnorm[ch.x]; k = nrm > 0.f ? coef / nrm : 0.f; }
  float s = 0.f;
  // chunks start at multiples of RN_CHUNK elements of a 16-byte aligned tensor: float4 loads for the whole quads
  const bool al = ((((uintptr_t)td.p) | ((uintptr_t)(mode ? td.g : td.p))) & 15) == 0;
  int i0 = ch.y;
  if (al) {
    const int nq = (end - ch.y) >> 2;
    const f32x4* p4 = reinterpret_cast<const f32x4*>(td.p + ch.y);
    const f32x4* g4 = reinterpret_cast<const f32x4*>((mode ? td.g : td.p) + ch.y);
    float s1 = 0.f, s2 = 0.f, s3 = 0.f;
    for (int q = threadIdx.x; q < nq; q += 256) {
      const f32x4 pv = p4[q];
      f32x4 x = pv;
      if (mode) { const f32x4 gv = g4[q]; x = gv + k * pv; }
      s += x[0] * x[0]; s1 += x[1] * x[1]; s2 += x[2] * x[2]; s3 += x[3] * x[3];
    }
    s = (s + s1) + (s2 + s3);
    i0 = ch.y + (nq << 2);
  }
  for (int i = i0 + threadIdx.x; i < end; i += 256) {
    const float x = mode == 0 ? td.p[i] : td.g[i] + k * td.p[i];
    s += x * x;
  }
  s = block_sum256(s, sm);
  if (threadIdx.x == 0) partial[blockIdx.x] = s;
}
// one block per tensor: out[tensor] = sqrt(sum of its chunk partials)   (deterministic order)
__global__ __launch_bounds__(256) void tensor_norm_kernel(const TensorDesc* __restrict__ tab, const float* __restrict__ partial,
                                                          float* __restrict__ out_norm) {
  __shared__ float sm[4];
  const TensorDesc td = tab[blockIdx.x];
  float s = 0.f;
  for (int c = threadIdx.x; c < td.nchunks; c += 256) s += partial[td.chunk0 + c];
  s = block_sum256(s, sm);
  if (threadIdx.x == 0) out_norm[blockIdx.x] = sqrtf(s);
}
// total = sqrt(sum_i norms[i]^2), clip coefficient of torch.nn.utils.clip_grad_norm_; also sum of norms.
__global__ void norm_finalize_kernel(const float* __restrict__ norms, int n, float max_norm, float* total_out,
                                     float* clip_out, float* sum_out) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  float ss = 0.f, sn = 0.f;
  for (int i = 0; i < n; ++i) { ss += norms[i] * norms[i]; sn += norms[i]; }
  const float tot = sqrtf(ss);
  if (total_out) *total_out = tot;
  if (sum_out) *sum_out = sn;
  if (clip_out) {
    float c = 1.f;
    if (max_norm > 0.f) { c = max_norm / (tot + 1e-6f); if (c > 1.f) c = 1.f; }
    *clip_out = c;
  }
}
// g += coef * p / ||p||   (autograd-compatible path: the regulariser's gradient, train.py:69-70)
__global__ __launch_bounds__(256) void add_reg_grad_kernel(const TensorDesc* __restrict__ tab, const int2* __restrict__ chunks,
                                                           const float* __restrict__ pnorm, float coef) {
  const int2 ch = chunks[blockIdx.x];
  const TensorDesc td = tab[ch.x];
  const int end = min(td.n, ch.y + RN_CHUNK);
  const float nrm = pnorm[ch.x];
  const float k = nrm > 0.f ? coef / nrm : 0.f;
  for (int i = ch.y + threadIdx.x; i < end; i += 256) td.g[i] += k * td.p[i];
}

// g *= *clip   (clip_grad_norm_ in place)
__global__ __launch_bounds__(256) void scale_grads_kernel(const TensorDesc* __restrict__ tab, const int2* __restrict__ chunks,
                                                          const float* __restrict__ clip) {
  const int2 ch = chunks[blockIdx.x];
  const TensorDesc td = tab[ch.x];
  const int end = min(td.n, ch.y + RN_CHUNK);
  const float c = *clip;
  if (c == 1.f) return;
  for (int i = ch.y + threadIdx.x; i < end; i += 256) td.g[i] *= c;
}

struct AdamHyper { double lr, beta1, beta2; float eps, wd, one_m_b1, beta2f, one_m_b2; int amsgrad; float reg_coef; };
// torch.optim.Adam (single-tensor form of torch 2.10): g' = clip * (g + reg) + wd * p ;
// m <- lerp(m, g', 1-b1) ; v <- b2 v + (1-b2) g'^2 ; [vmax <- max(vmax, v)] ;
// p <- p - (lr / bc1) * m / (sqrt(v̂) / sqrt(bc2) + eps)
__global__ __launch_bounds__(256) void adam_chunk_kernel(const TensorDesc* __restrict__ tab, const int2* __restrict__ chunks,
                                                         AdamHyper hp, const float* __restrict__ pnorm,
                                                         const float* __restrict__ clip, const int32_t* __restrict__ step_ptr,
                                                         const PackDesc* __restrict__ pack, int lp) {
  __shared__ float sc[2];
  if (threadIdx.x == 0) {
    const double st = (double)(*step_ptr);
    const double bc1 = 1.0 - pow(hp.beta1, st);
    const double bc2 = 1.0 - pow(hp.beta2, st);
    sc[0] = (float)(hp.lr / bc1);
    sc[1] = (float)sqrt(bc2);
  }
  __syncthreads();
  const float step_size = sc[0], bc2s = sc[1];
  const int2 ch = chunks[blockIdx.x];
  const TensorDesc td = tab[ch.x];
  const int end = min(td.n, ch.y + RN_CHUNK);
  const float nrm = pnorm ? pnorm[ch.x] : 0.f;
  const float k = (nrm > 0.f) ? hp.reg_coef / nrm : 0.f;
  const float cl = clip ? *clip : 1.f;
  PackDesc pk; pk.ndst = 0; pk.cols = 1;
  if (pack) pk = pack[ch.x];
  for (int i = ch.y + threadIdx.x; i < end; i += 256) {
    const float p = td.p[i];
    float g = (td.g[i] + k * p) * cl;
    g = g + hp.wd * p;
    float m = td.m[i];
    m = m + hp.one_m_b1 * (g - m);
    float v = td.v[i] * hp.beta2f + hp.one_m_b2 * g * g;
    td.m[i] = m; td.v[i] = v;
    float vh = v;
    if (hp.amsgrad) { vh = fmaxf(td.vmax[i], v); td.vmax[i] = vh; }
    const float denom = sqrtf(vh) / bc2s + hp.eps;
    const float pn = p - step_size * (m / denom);
    td.p[i] = pn;
    if (pk.ndst) {
      const int r = i / pk.cols, c = i - r * pk.cols;
#pragma unroll
      for (int d = 0; d < 6; ++d)
        if (d < pk.ndst && c >= pk.d[d].c0 && c < pk.d[d].c0 + pk.d[d].nc &&
            (pk.d[d].mode || (r >= pk.d[d].r0 && r < pk.d[d].r0 + pk.d[d].nr))) {
          int dr = r - pk.d[d].r0;
          if (pk.d[d].mode) { const int gate = r / pk.d[d].nr, u = r - gate * pk.d[d].nr; dr = (u >> 3) * 32 + gate * 8 + (u & 7); }
          const size_t o = (size_t)dr * pk.d[d].ld + (c - pk.d[d].c0);
          if (lp) reinterpret_cast<bf16_t*>(pk.d[d].dst)[o] = (bf16_t)pn; else reinterpret_cast<float*>(pk.d[d].dst)[o] = pn;
        }
    }
  }
}
