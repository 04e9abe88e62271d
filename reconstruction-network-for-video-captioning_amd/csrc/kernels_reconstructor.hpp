// RecNet gfx950 kernels: reconstructors: element-wise LSTM / GRU step, global helpers, MSE, local attention forward / backward.
// Included through kernels.hpp.
#pragma once
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

// 4-wide form: a thread owns 4 consecutive hidden units, every slab / input / state access is a 16-byte load or store
// and exactly S slab loads per gate are issued (the scalar form reads one float at a time and pads its slab loop to 16
// loads).  Requires Hd, the leading dimensions and the column offsets to be multiples of 4 and 16-byte aligned bases.
template <typename AT> struct AT4;
template <> struct AT4<float> { typedef f32x4 type; static __device__ __forceinline__ type cvt(f32x4 v) { return v; } };
template <> struct AT4<bf16_t> {
  typedef bf16x4 type;
  static __device__ __forceinline__ type cvt(f32x4 v) { type r; r[0] = (bf16_t)v[0]; r[1] = (bf16_t)v[1]; r[2] = (bf16_t)v[2]; r[3] = (bf16_t)v[3]; return r; }
};
template <typename AT>
__global__ __launch_bounds__(256) void lstm_pw_vec_kernel(const LstmPwArgs p) {
  const int Hq = p.Hd >> 2, i = blockIdx.x * 256 + threadIdx.x;
  if (i >= p.B * Hq) return;
  const int b = i / Hq, u = (i - b * Hq) << 2, Hd = p.Hd;
  f32x4 g[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int col = q * Hd + u;
    f32x4 v = p.X ? *reinterpret_cast<const f32x4*>(p.X + (size_t)b * p.x_ld + col) : f32x4{0.f, 0.f, 0.f, 0.f};
    if (p.b1) v += *reinterpret_cast<const f32x4*>(p.b1 + col);
    if (p.b2) v += *reinterpret_cast<const f32x4*>(p.b2 + col);
    g[q] = v;
  }
  const f32x4 cp = p.c_prev ? *reinterpret_cast<const f32x4*>(p.c_prev + (size_t)b * Hd + u) : f32x4{0.f, 0.f, 0.f, 0.f};
  {
    const float* sp = p.slab + (size_t)b * p.slab_ld + u;
#pragma unroll 4
    for (int z = 0; z < p.S; ++z) {
#pragma unroll
      for (int q = 0; q < 4; ++q) g[q] += *reinterpret_cast<const f32x4*>(sp + (size_t)z * p.slab_stride + q * Hd);
    }
  }
  f32x4 hv, a0, a1, a2, a3, cv;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    if (p.gru) {
      const GruOut r = gru_point(g[0][j], g[1][j], g[2][j], g[3][j], cp[j]);
      hv[j] = r.h; a0[j] = r.r; a1[j] = r.z; a2[j] = r.n; a3[j] = r.hn; cv[j] = 0.f;
    } else {
      const LstmOut r = lstm_point(g[0][j], g[1][j], g[2][j], g[3][j], cp[j]);
      hv[j] = r.h; a0[j] = r.i; a1[j] = r.f; a2[j] = r.g; a3[j] = r.o; cv[j] = r.c;
    }
  }
  if (!p.gru) *reinterpret_cast<f32x4*>(p.c_out + (size_t)b * Hd + u) = cv;
  *reinterpret_cast<f32x4*>(p.h_out + (size_t)b * p.h_ld + u) = hv;
  typedef typename AT4<AT>::type at4;
  if (p.h_lp) {
    AT* d = reinterpret_cast<AT*>(p.h_lp) + (size_t)b * p.hlp_ld;
    *reinterpret_cast<at4*>(d + u) = AT4<AT>::cvt(hv);
    if (u < p.hlp_ld - p.hlp_pad_from) {          // pad columns [Hd, hlp_ld): fewer than 8, zeroed by the first threads
      for (int j = 0; j < 4; ++j) if (p.hlp_pad_from + u + j < p.hlp_ld) d[p.hlp_pad_from + u + j] = (AT)0.f;
    }
  }
  if (p.h_lp2) *reinterpret_cast<at4*>(reinterpret_cast<AT*>(p.h_lp2) + (size_t)b * p.hlp2_ld + u) = AT4<AT>::cvt(hv);
  float* a = p.acts + (size_t)b * 4 * Hd + u;
  *reinterpret_cast<f32x4*>(a) = a0; *reinterpret_cast<f32x4*>(a + Hd) = a1;
  *reinterpret_cast<f32x4*>(a + 2 * Hd) = a2; *reinterpret_cast<f32x4*>(a + 3 * Hd) = a3;
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

template <typename AT>
__global__ __launch_bounds__(256) void lstm_bwd_vec_kernel(const LstmBwdArgs p) {
  const int Hq = p.Hd >> 2, i = blockIdx.x * 256 + threadIdx.x;
  if (i >= p.B * Hq) return;
  const int b = i / Hq, u = (i - b * Hq) << 2, Hd = p.Hd;
  const size_t o = (size_t)b * Hd + u;
  f32x4 dh = p.dh_direct ? p.dh_scale * *reinterpret_cast<const f32x4*>(p.dh_direct + (size_t)b * p.dhd_ld + u) : f32x4{0.f, 0.f, 0.f, 0.f};
  const float* a = p.acts + (size_t)b * 4 * Hd + u;
  const f32x4 a0 = *reinterpret_cast<const f32x4*>(a), a1 = *reinterpret_cast<const f32x4*>(a + Hd),
              a2 = *reinterpret_cast<const f32x4*>(a + 2 * Hd), a3 = *reinterpret_cast<const f32x4*>(a + 3 * Hd);
  const f32x4 zero = f32x4{0.f, 0.f, 0.f, 0.f};
  const f32x4 carry = p.first ? zero : *reinterpret_cast<const f32x4*>(p.dc_carry + o);
  const f32x4 cpv = p.c_prev ? *reinterpret_cast<const f32x4*>(p.c_prev + o) : zero;
  const f32x4 cc = p.gru ? zero : *reinterpret_cast<const f32x4*>(p.c + o);
  {
    const float* sp = p.slab + (size_t)b * p.slab_ld + p.slab_col0 + u;
#pragma unroll 8
    for (int z = 0; z < p.S; ++z) dh += *reinterpret_cast<const f32x4*>(sp + (size_t)z * p.slab_stride);
    const float* s2 = p.slab2 + (size_t)b * Hd + u;
#pragma unroll 8
    for (int z = 0; z < p.S2; ++z) dh += *reinterpret_cast<const f32x4*>(s2 + (size_t)z * p.slab2_stride);
  }
  f32x4 di, df, dg, d_o, dcp;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const LstmGrad g = p.gru ? gru_point_bwd(dh[j] + carry[j], a0[j], a1[j], a2[j], a3[j], cpv[j])
                             : lstm_point_bwd(dh[j], carry[j], a0[j], a1[j], a2[j], a3[j], cc[j], cpv[j]);
    di[j] = g.di; df[j] = g.df; dg[j] = g.dg; d_o[j] = g.d_o; dcp[j] = g.dc_prev;
  }
  typedef typename AT4<AT>::type at4;
  AT* d = reinterpret_cast<AT*>(p.dG) + (size_t)b * p.ld_dg;
  *reinterpret_cast<at4*>(d + u) = AT4<AT>::cvt(di); *reinterpret_cast<at4*>(d + Hd + u) = AT4<AT>::cvt(df);
  *reinterpret_cast<at4*>(d + 2 * Hd + u) = AT4<AT>::cvt(dg); *reinterpret_cast<at4*>(d + 3 * Hd + u) = AT4<AT>::cvt(d_o);
  if (u < p.ld_dg - 4 * Hd) for (int j = 0; j < 4; ++j) if (4 * Hd + u + j < p.ld_dg) d[4 * Hd + u + j] = (AT)0.f;
  *reinterpret_cast<f32x4*>(p.dc_carry + o) = dcp;
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
                                   int T, int B, int H, DropDesc dd, int t0) {
  const uint32_t key = drop_key(dd);
  const size_t total = (size_t)T * B * ld;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % ld), b = (int)((i / ld) % B), t = (int)(i / ((size_t)ld * B));
    AT v = (AT)0.f;
    if (c < H) v = hs[((size_t)t * B + b) * ld_hs + c];
    else if (c < 2 * H) v = (AT)(mp[(size_t)b * H + (c - H)] * drop_at(dd, key, t0 + t, b, H, c - H));   // t0: the per-step API
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
// Global reconstructor, d loss / d decoder states from the merged input-side product tmp[t,b,0:2H) = dG_t . W_ih:
//   dmp[b,h] = sum_t tmp[t,b,H+h] * dropmask(t,b,h) ;  dhid[t,b,h] = tmp[t,b,h] + c * dmp[b,h]     (c = caption_max_len / T^2)
// Block = 64 consecutive (b, h) x 4 slices of t: a thread's loads are independent and issued eight at a time (the one-thread-per-(b, h)
// form walked its 2 T strided loads one by one on four waves per CU: 16.6 us in front of the decoder's BPTT; this one 5)
__global__ __launch_bounds__(256) void global_dhid_kernel(const float* __restrict__ tmp, float* __restrict__ dhid, int T, int B, int H, float c, DropDesc dd) {
  __shared__ float part[4][64];
  const uint32_t key = drop_key(dd);
  const size_t n = (size_t)B * H;
  const int q = threadIdx.x >> 6;
  const size_t i = (size_t)blockIdx.x * 64 + (threadIdx.x & 63);
  const bool on = i < n;
  const int h = on ? (int)(i % H) : 0, b = on ? (int)(i / H) : 0;
  const float* row = tmp + (size_t)b * 2 * H + h;
  const int tq = (T + 3) >> 2, t0 = q * tq, t1 = t0 + tq < T ? t0 + tq : T;
  float s = 0.f;
  if (on) {
#pragma unroll 8
    for (int t = t0; t < t1; ++t) s += row[(size_t)t * B * 2 * H + H] * drop_at(dd, key, t, b, H, h);
  }
  part[q][threadIdx.x & 63] = s;
  __syncthreads();
  // (slice order: the sum does not depend on the schedule)
  s = ((part[0][threadIdx.x & 63] + part[1][threadIdx.x & 63]) + (part[2][threadIdx.x & 63] + part[3][threadIdx.x & 63])) * c;
  if (on) {
#pragma unroll 8
    for (int t = t0; t < t1; ++t) dhid[(size_t)t * n + i] = row[(size_t)t * B * 2 * H] + s;
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

// 4-wide form (R % 4 == 0, 16-byte aligned rows): 16-byte accesses, and optionally the operand copy
// dlp[i] = (AT)(lp_scale * gcoef * diff) of d loss / d out that the backward's GEMMs read (ld_lp == R: no padding to zero)
template <typename AT>
__global__ __launch_bounds__(256) void mse_vec_kernel(float* __restrict__ out, const float* __restrict__ ref, int Sn, int B,
                                                      int R, size_t ref_bstride, size_t ref_sstride, float gcoef,
                                                      float* __restrict__ partial, AT* __restrict__ dlp, float lp_scale) {
  __shared__ float sm[4];
  const int Rq = R >> 2;
  const size_t total = (size_t)Sn * B * Rq;
  float acc = 0.f;
  typedef typename AT4<AT>::type at4;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int r = (int)(i % Rq) << 2, b = (int)((i / Rq) % B), s = (int)(i / ((size_t)Rq * B));
    const size_t o = (i / Rq) * R + r;
    const f32x4 v = *reinterpret_cast<const f32x4*>(out + o);
    const f32x4 e = *reinterpret_cast<const f32x4*>(ref + (size_t)b * ref_bstride + (size_t)s * ref_sstride + r);
    const f32x4 d = v - e;
    acc += (d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3]);
    const f32x4 g = gcoef * d;
    *reinterpret_cast<f32x4*>(out + o) = g;
    if (dlp) *reinterpret_cast<at4*>(dlp + o) = AT4<AT>::cvt(lp_scale * g);
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
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
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
  int slab_w;             // row width of the slabs: 0 = H + R (dx in columns [0, H) of the [dx | dhr] product), else the width given (H: dx only)
  const float* slab;      // [S][B][slab width]
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
  const int b = blockIdx.x, ch = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int H = p.H, R = p.R, A = p.A, T = p.T;
  const int W2 = p.slab_w ? p.slab_w : H + R;
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

