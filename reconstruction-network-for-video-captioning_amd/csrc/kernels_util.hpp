// RecNet gfx950 kernels: small utilities, reductions, packing, column sums, gate math (LSTM / GRU), embedding.
// Included through kernels.hpp.
#pragma once
// =============================================================================================
// small utilities
// =============================================================================================
// one 100 MHz wall-clock stamp (recnet_read_stamps: start / end of a phase that is not one grouped launch)
__global__ void stamp_u64_kernel(unsigned long long* slot, int take_max) {
  const unsigned long long t = wall_clock64();
  if (take_max) __hip_atomic_fetch_max(slot, t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else __hip_atomic_store(slot, t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__global__ void set_u32_kernel(uint32_t* p, uint32_t v) { *p = v; }
// Data parallel: a rank whose chain kernel gave up (rec_chain.hpp: the poison word is NaN, else 0) marks one element of the
// gradient bucket that is all-reduced LAST; after the SUM every rank sees the NaN and raises its own poison word, so all
// ranks skip the optimiser update together (adam_chunk_kernel tests the word) instead of the healthy ones applying garbage.
__global__ void poison_mark_kernel(float* g0, const float* poison) { if (*poison != 0.f) *g0 = __int_as_float(0x7fc00000); }
__global__ void poison_collect_kernel(const float* g0, float* poison) { if (*g0 != *g0) *poison = __int_as_float(0x7fc00000); }
__global__ void set_f32_kernel(float* p, float v) { *p = v; }
// step counter += 1; seed slot = seed_base + step (graph-replay friendly train step)
// A ring of the last seven stamps of a one-thread kernel (ring[0] = count): the step's first and last kernels keep one each, so that
// the idle time BETWEEN back-to-back replayed steps can be read directly (recnet_read_step_ring) instead of as a difference of totals
__device__ __forceinline__ void rn_ring_push(unsigned long long* ring, unsigned long long now) {
  const unsigned long long c = ring[0];
  ring[1 + (c % 7ull)] = now; ring[0] = c + 1ull;
}
__global__ void advance_step_kernel(int32_t* step, uint32_t* seed_slot, uint32_t seed_base) {
  int s = *step + 1; *step = s; *seed_slot = seed_base + (uint32_t)s;
  const unsigned long long now = (unsigned long long)wall_clock64();
  __hip_atomic_store(reinterpret_cast<unsigned long long*>(seed_slot + 32), now, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // phase stamp: step start (recnet_read_stamps, wait_chain_kernel)
  rn_ring_push(reinterpret_cast<unsigned long long*>(seed_slot + 34), now);      // ctrl[34 .. 49]
}

// step counter = step (host-numbered train step), with the step-start stamp of advance_step_kernel
__global__ void set_step_kernel(int32_t* step, int32_t v) {
  *step = v;
  __hip_atomic_store(reinterpret_cast<unsigned long long*>(step + 31), (unsigned long long)wall_clock64(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
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

// The reconstructor's loss scalars in one launch: mse = scale * sum_j x[j] ; rec_loss = mse + lam_reg * reg ;
// total = dec_loss + lam_recon * rec_loss   (scal: [2] dec_loss [3] mse [4] reg [5] rec_loss [6] total)
__global__ __launch_bounds__(256) void rec_loss_finalize_kernel(const float* __restrict__ x, int n, float scale, float* scal, float lam_reg,
                                                                float lam_recon) {
  __shared__ float sm[4];
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) s += x[i];
  s = block_sum256(s, sm);
  if (threadIdx.x == 0) {
    const float mse = s * scale, rl = mse + lam_reg * scal[4];
    scal[3] = mse; scal[5] = rl; scal[6] = scal[2] + lam_recon * rl;
  }
}
// out[i] = scale * sum_j x[j]   (single block; deterministic order)
__global__ __launch_bounds__(256) void reduce_sum_kernel(const float* __restrict__ x, int n, float* out, float scale) {
  __shared__ float sm[4];
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) s += x[i];
  s = block_sum256(s, sm);
  if (threadIdx.x == 0) *out = s * scale;
}

// ---- gradient transport of the data-parallel step, direct reduce-scatter form (dp.py: GradTransport): the
// staging around the two collectives as two kernels instead of W + 3 torch launches.
// dp_cast: dst[i] = (WT) src[i] (fp32 -> wire type), or fp32 <- wire type on the way back; 16 bytes of fp32 per lane and pass.
template <typename DT, typename ST>
__global__ __launch_bounds__(256) void dp_cast_kernel(const ST* __restrict__ src, DT* __restrict__ dst, long n) {
  const long n4 = n >> 2;
  for (long q = (long)blockIdx.x * 256 + threadIdx.x; q < n4; q += (long)gridDim.x * 256) {
    const long i = q << 2;
    float v[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = (float)src[i + e];
#pragma unroll
    for (int e = 0; e < 4; ++e) dst[i + e] = (DT)v[e];
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) dst[(n4 << 2) + threadIdx.x] = (DT)(float)src[(n4 << 2) + threadIdx.x];
}
// dp_reduce: red[j] = (WT) (sum over ranks r = 0 .. W-1, IN RANK ORDER, of (float) recv[r * chunk + j]) — fp32 accumulation at the
// destination, every contribution rounded to the wire type once by its sender and the sum once here: all ranks that reduce the
// same chunk get the same bytes.  All W loads of an element are in flight together (W <= 16).
template <typename WT>
__global__ __launch_bounds__(256) void dp_reduce_kernel(const WT* __restrict__ recv, int W, long chunk, WT* __restrict__ red) {
  for (long j = (long)blockIdx.x * 256 + threadIdx.x; j < chunk; j += (long)gridDim.x * 256) {
    float v[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) v[r] = r < W ? (float)recv[(long)r * chunk + j] : 0.f;
    float a = v[0];
#pragma unroll
    for (int r = 1; r < 16; ++r) if (r < W) a += v[r];
    red[j] = (WT)a;
  }
}

// The decoder's loss scalars in one launch (round 5: a row-sum kernel and two one-thread kernels before): scal[0] = CE = sum of the
// weighted row losses, scal[2] = dec_loss = CE + lambda_reg * reg (scal[1], from the parameter norms), scal[6] = total so far
__global__ __launch_bounds__(256) void dec_loss_finalize_kernel(const float* __restrict__ rowloss, int n, float* scal, float lambda_reg) {
  __shared__ float sm[4];
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) s += rowloss[i];
  s = block_sum256(s, sm);
  if (threadIdx.x == 0) { scal[0] = s; const float dl = s + lambda_reg * scal[1]; scal[2] = dl; scal[6] = dl + 0.f * dl; }
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
// dst[r][c] = (AT) src[r * ld_src + c] for c < cols only (a column window of wider operand rows; nothing else is touched)
template <typename AT>
__global__ void pack_cols_kernel(AT* __restrict__ dst, int ld_dst, const float* __restrict__ src, int ld_src, int rows, int cols) {
  const size_t total = (size_t)rows * cols;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int r = (int)(i / cols), c = (int)(i % cols);
    dst[(size_t)r * ld_dst + c] = (AT)src[(size_t)r * ld_src + c];
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
// Optional softmax over the frames of the attention energies (recnet_config.decoder_attn_normalize; the reference builds
// nn.Softmax(dim=1) at decoder.py:30 and never calls it, so the default is none).  sa[0..F) in LDS; wave 0 does the
// wavefront max / sum reductions; the caller brackets the call with __syncthreads().
__device__ __forceinline__ void attn_softmax_lds(float* sa, int F, float* att_out) {
  if (threadIdx.x < 64) {
    const int lane = threadIdx.x;
    float m = -3.0e38f;
    for (int f = lane; f < F; f += 64) m = fmaxf(m, sa[f]);
    m = wave_max(m);
    float s = 0.f;
    for (int f = lane; f < F; f += 64) s += __expf(sa[f] - m);
    s = wave_sum(s);
    const float inv = 1.0f / s;
    for (int f = lane; f < F; f += 64) {
      const float a = __expf(sa[f] - m) * inv;
      sa[f] = a;
      if (att_out) att_out[f] = a;
    }
  }
}
// its backward on the frame gradients: sda[f] <- a[f] (sda[f] - sum_g a[g] sda[g]),  a = the saved weights
__device__ __forceinline__ void attn_softmax_bwd_lds(float* sda, const float* att, int F) {
  if (threadIdx.x < 64) {
    const int lane = threadIdx.x;
    float d = 0.f;
    for (int f = lane; f < F; f += 64) d += att[f] * sda[f];
    d = wave_sum(d);
    for (int f = lane; f < F; f += 64) sda[f] = att[f] * (sda[f] - d);
  }
}

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
// The decoder prologue's three small kernels as ONE launch (bf16 path, fused step): blocks [0, nb_pack) cast the features to
// the bf16 operand copy (16-byte loads, 8-byte stores), the next ceil(rowsT / 2) blocks gather two embedding rows each, the last
// block forms the gate bias b_ih + b_hh — no fork / join of two streams in front of the grouped product that consumes all three.
struct ProPackArgs {
  bf16_t* enc_lp; int ld_enc; const float* enc; int rowsE, D;
  const float* Emb; const int64_t* targets; bf16_t* emb; int ld_emb, B, E, V; float scale; DropDesc dd; int rowsT;
  const float* bih; const float* bhh; float* bsum; int Hd, gru;
  int nb_pack;
};
__global__ __launch_bounds__(256) void prologue_pack_kernel(const ProPackArgs p) {
  const int blk = blockIdx.x, tid = threadIdx.x;
  if (blk < p.nb_pack) {
    if ((p.D & 3) == 0 && (p.ld_enc & 3) == 0 && ((((uintptr_t)p.enc) | ((uintptr_t)p.enc_lp)) & 15) == 0) {
      const int q4 = p.ld_enc >> 2;                       // quads per destination row
      const size_t total = (size_t)p.rowsE * q4;
      for (size_t i = (size_t)blk * 256 + tid; i < total; i += (size_t)p.nb_pack * 256) {
        const int r = (int)(i / q4), c = (int)(i % q4) * 4;
        bf16x4 o = {0, 0, 0, 0};
        if (c < p.D) { const f32x4 v = *reinterpret_cast<const f32x4*>(p.enc + (size_t)r * p.D + c); o[0] = (bf16_t)v[0]; o[1] = (bf16_t)v[1]; o[2] = (bf16_t)v[2]; o[3] = (bf16_t)v[3]; }
        *reinterpret_cast<bf16x4*>(p.enc_lp + (size_t)r * p.ld_enc + c) = o;
      }
    } else {
      const size_t total = (size_t)p.rowsE * p.ld_enc;
      for (size_t i = (size_t)blk * 256 + tid; i < total; i += (size_t)p.nb_pack * 256) {
        const int r = (int)(i / p.ld_enc), c = (int)(i % p.ld_enc);
        p.enc_lp[i] = (bf16_t)(c < p.D ? p.enc[(size_t)r * p.D + c] : 0.f);
      }
    }
    return;
  }
  const int eb = blk - p.nb_pack, neb = (p.rowsT + 1) >> 1;
  if (eb < neb) {
    const int row = eb * 2 + (tid >> 7), t128 = tid & 127;
    if (row >= p.rowsT) return;
    const int t = row / p.B, b = row % p.B;
    long tok = t == 0 ? 1 : p.targets[(size_t)(t - 1) * p.B + b];
    tok = tok < 0 ? 0 : (tok >= p.V ? p.V - 1 : tok);
    const uint32_t key = drop_key(p.dd);
    const float* src = p.Emb + (size_t)tok * p.E;
    bf16_t* dst = p.emb + (size_t)row * p.ld_emb;
    for (int j = t128; j < p.ld_emb; j += 128) dst[j] = (bf16_t)(j < p.E ? src[j] * p.scale * drop_at(p.dd, key, t, b, p.E, j) : 0.f);
    return;
  }
  for (int i = tid; i < 4 * p.Hd; i += 256) {
    if (!p.gru) { p.bsum[i] = p.bih[i] + p.bhh[i]; continue; }
    const int g = i / p.Hd;
    p.bsum[i] = g < 2 ? p.bih[i] + p.bhh[i] : (g == 2 ? p.bih[i] : p.bhh[i - p.Hd]);
  }
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

