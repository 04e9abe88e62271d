// RecNet gfx950 kernels: decoder recurrent step: forward cell kernels, backward cell kernel, all-step context, masked CE.
// Included through kernels.hpp.
#pragma once
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
  int softmax;            // 1: softmax over the frames of the energies (attn_softmax_lds)
};

template <typename AT>
__global__ __launch_bounds__(1024) void dec_cell_kernel(const DecCellArgs p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* swh = smem;            // [A]
  float* sa = swh + p.A;        // [F]
  float* spre = sa + p.F;       // [4 * UC]
  const int NT = blockDim.x, UC = NT >> 2, NW = NT >> 6;
  const int b = blockIdx.x, u0 = blockIdx.y * UC, tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
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
        if (lane == 0) { sa[f] = s; if (p.att_out && blockIdx.y == 0 && !p.softmax) p.att_out[(size_t)b * F + f] = s; }
      }
    }
  } else {
    for (int f = wave; f < F; f += NW) {
      const float* uv = p.Uv + ((size_t)b * F + f) * A;
      float s = 0.f;
      for (int k = lane; k < A; k += 64) s += p.w[k] * rn_tanh(swh[k] + uv[k] + p.ab[k]);
      s = wave_sum(s);
      if (lane == 0) { sa[f] = s; if (p.att_out && blockIdx.y == 0 && !p.softmax) p.att_out[(size_t)b * F + f] = s; }
    }
  }
  __syncthreads();
  if (p.softmax) { attn_softmax_lds(sa, F, (p.att_out && blockIdx.y == 0) ? p.att_out + (size_t)b * F : nullptr); __syncthreads(); }
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
  const int b = blockIdx.x, u0 = blockIdx.y * 512, tid = threadIdx.x, lane = tid & 63, g = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int H = p.H, A = p.A, F = p.F, W4 = 4 * H, WS = 4 * H + A;
  const size_t zs = (size_t)p.B * WS;
  const int u = u0 + lane * 8;
  const bool live = u < H;                    // H % 8 == 0: a lane's 8 units are all inside or all outside
  const int col = g * H + u;
  // ---- every global load of the kernel is ISSUED here, before the first use of any loaded value: the kernel is one
  // link of a dependent chain at ~1 workgroup per CU, so each "load, wait, use, load" costs a full memory round trip
  // (the compiler keeps program order between a use and the loads that follow it; sched_barrier pins the block)
  Raw8<AT> pv[32];
  f32x4 x0 = {0.f, 0.f, 0.f, 0.f}, x1 = x0, sl[4][2];
#pragma unroll
  for (int z = 0; z < 4; ++z) { sl[z][0] = x0; sl[z][1] = x0; }
  const int S4 = p.slab ? (p.S < 4 ? p.S : 4) : 0;
  if (live) {
    const AT* pp = reinterpret_cast<const AT*>(p.P) + (size_t)b * F * p.ldp + col;
#pragma unroll
    for (int f = 0; f < 32; ++f) { if (f < F) pv[f].load(pp + (size_t)f * p.ldp); else pv[f].zero(); }
    x0 = *reinterpret_cast<const f32x4*>(p.Xe + (size_t)b * W4 + col);
    x1 = *reinterpret_cast<const f32x4*>(p.Xe + (size_t)b * W4 + col + 4);
    if (S4) {
      const float* sp0 = p.slab + (size_t)b * WS + col;
#pragma unroll
      for (int z = 0; z < 4; ++z) {
        const int zz = z < S4 ? z : S4 - 1;          // clamped duplicates instead of a branch
        sl[z][0] = *reinterpret_cast<const f32x4*>(sp0 + (size_t)zz * zs);
        sl[z][1] = *reinterpret_cast<const f32x4*>(sp0 + (size_t)zz * zs + 4);
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
  // attention vectors, c_{t-1} of the pointwise phase, and this thread's column of the attention pre-activation W h
  float wk[2], bk[2], cpre[2], whv[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int k = lane + 64 * j;
    wk[j] = k < A ? p.w[k] : 0.f; bk[j] = k < A ? p.ab[k] : 0.f;
    const int uu = u0 + tid + 256 * j;
    cpre[j] = (p.c_prev && uu < H) ? p.c_prev[(size_t)b * H + uu] : 0.f;
  }
  if (S4 && tid < A) {
#pragma unroll
    for (int z = 0; z < 4; ++z) whv[z] = p.slab[(size_t)(z < S4 ? z : S4 - 1) * zs + (size_t)b * WS + W4 + tid];
  }
  __builtin_amdgcn_sched_barrier(0);
  // ---- uses
  float pre[8];
  pre[0] = x0[0]; pre[1] = x0[1]; pre[2] = x0[2]; pre[3] = x0[3]; pre[4] = x1[0]; pre[5] = x1[1]; pre[6] = x1[2]; pre[7] = x1[3];
#pragma unroll
  for (int z = 0; z < 4; ++z)
    if (z < S4) {
      pre[0] += sl[z][0][0]; pre[1] += sl[z][0][1]; pre[2] += sl[z][0][2]; pre[3] += sl[z][0][3];
      pre[4] += sl[z][1][0]; pre[5] += sl[z][1][1]; pre[6] += sl[z][1][2]; pre[7] += sl[z][1][3];
    }
  if (live && p.slab)
    for (int z = 4; z < p.S; ++z) {               // split-K beyond 4 (not used by the chain's caps): plain loop
      const float* sp = p.slab + (size_t)b * WS + col + (size_t)z * zs;
      const f32x4 s0 = *reinterpret_cast<const f32x4*>(sp), s1 = *reinterpret_cast<const f32x4*>(sp + 4);
      pre[0] += s0[0]; pre[1] += s0[1]; pre[2] += s0[2]; pre[3] += s0[3];
      pre[4] += s1[0]; pre[5] += s1[1]; pre[6] += s1[2]; pre[7] += s1[3];
    }
  for (int k = tid; k < A; k += 256) {
    float v = 0.f;
    if (p.slab) {
      if (k == tid) {
#pragma unroll
        for (int z = 0; z < 4; ++z) v += z < S4 ? whv[z] : 0.f;
        for (int z = 4; z < p.S; ++z) v += p.slab[(size_t)z * zs + (size_t)b * WS + W4 + k];
      } else {
        v = sum_strided(p.slab + (size_t)b * WS + W4 + k, zs, p.S);
      }
    }
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
        if (lane == 0) { sa[f] = s; if (p.att_out && blockIdx.y == 0 && !p.softmax) p.att_out[(size_t)b * F + f] = s; }
      }
    }
  }
  __syncthreads();
  if (p.softmax) { attn_softmax_lds(sa, F, (p.att_out && blockIdx.y == 0) ? p.att_out + (size_t)b * F : nullptr); __syncthreads(); }
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
  const float* att; int softmax;        // softmax mode: the saved attention weights [B][F] of step t
};

template <typename AT>
__global__ __launch_bounds__(256) void dec_cell_bwd_kernel(const DecCellBwdArgs p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* sdg = smem;            // [4H]
  float* sda = sdg + 4 * p.H;   // [F]
  float* spart = sda + p.F;     // [2][G][A] partial sums
  const int b = blockIdx.x, ch = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
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
  float uvr[4], duvr[4], whk_pre = 0.f, abk_pre = 0.f, wk_pre = 0.f;
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
    // (kept as separate registers: adding them here would be a use, i.e. a wait for everything issued so far)
    if (gi < G) { whk_pre = p.Wh[(size_t)b * A + kk]; abk_pre = p.ab[kk]; wk_pre = p.w[kk]; }
  }
  // pointwise backward of the whole row.  H <= 512, S <= 16: the loads of both units a thread owns (u = tid, tid + 256)
  // are all issued before the first use — with the loads above that is the kernel's single memory round trip
  if (H <= 512 && p.S <= 16) {
    // Unconditional loads from always-valid addresses (absent inputs alias a present one and are masked in the
    // arithmetic): a select on a loaded value, like an add, is a use and would split the loads into several round trips.
    float d1[2], d2[2], sl[2][16], av[2][4], cr[2], cpv[2], cv[2];
    const int Sn = p.slab ? p.S : 0;
    const float* q2 = p.dHs2 ? p.dHs2 : p.dHs;
    const float* qs = p.slab ? p.slab : p.dHs;
    const float* qc = p.c_prev ? p.c_prev : p.c;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int u = tid + 256 * q;
      const size_t o = (size_t)b * H + (u < H ? u : 0);
      d1[q] = p.dHs[o];
      d2[q] = q2[o];
#pragma unroll
      for (int z = 0; z < 16; ++z) sl[q][z] = qs[o + (size_t)(z < Sn ? z : (Sn ? Sn - 1 : 0)) * (Sn ? zs : 0)];
      const float* a = p.acts + (size_t)b * W4 + (u < H ? u : 0);
#pragma unroll
      for (int j = 0; j < 4; ++j) av[q][j] = a[(size_t)j * H];
      cr[q] = p.dc_in[o];
      cpv[q] = qc[o];
      cv[q] = p.c[o];
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int u = tid + 256 * q;
      if (u < H) {
        if (!p.dHs2) d2[q] = 0.f;
        if (p.first) cr[q] = 0.f;
        if (!p.c_prev) cpv[q] = 0.f;
        float dh = d1[q] + d2[q];
        float s0 = 0.f, s1 = 0.f;
#pragma unroll
        for (int z = 0; z < 16; z += 2) { s0 += z < Sn ? sl[q][z] : 0.f; s1 += z + 1 < Sn ? sl[q][z + 1] : 0.f; }
        dh += s0 + s1;
        const LstmGrad g = p.gru ? gru_point_bwd(dh + cr[q], av[q][0], av[q][1], av[q][2], av[q][3], cpv[q])
                                 : lstm_point_bwd(dh, cr[q], av[q][0], av[q][1], av[q][2], av[q][3], cv[q], cpv[q]);
        sdg[u] = g.di; sdg[H + u] = g.df; sdg[2 * H + u] = g.dg; sdg[3 * H + u] = g.d_o;
        if (ch == 0) {
          dgx[u] = (AT)g.di; dgx[H + u] = (AT)g.df; dgx[2 * H + u] = (AT)g.dg; dgx[3 * H + u] = (AT)g.d_o;
          p.dc_out[(size_t)b * H + u] = g.dc_prev;
        }
      }
    }
  } else
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
  if (p.softmax) {
    // the softmax couples the frames: every chunk workgroup needs da of ALL frames before it can turn them into energy
    // gradients (opt-in mode: the 4x redundant frame products are a few microseconds)
    for (int f = wave; f < F; f += 4) {
      const AT* pp = Pb + (size_t)f * p.ldp;
      float s = 0.f;
      for (int n = lane; n < W4; n += 64) s += sdg[n] * (float)pp[n];
      s = wave_sum(s);
      if (lane == 0) sda[f] = s * invF;
    }
    __syncthreads();
    attn_softmax_bwd_lds(sda, p.att + (size_t)b * F, F);
  } else if (fast) {
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
      const float whk = whk_pre + abk_pre, wk = wk_pre;
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
  // Round 5: the weights of a frame as [f][32 t] in LDS, read as eight 16-byte broadcasts per frame, and TWO feature columns per
  // thread — 8 LDS instructions per frame and 64 multiply-adds instead of 32 four-byte LDS reads per 32 multiply-adds: the kernel was
  // bound by LDS instruction issue (84 us beside the reconstructor's chains for 27 MB of traffic).  grid (B, ceil(ld / 512)).
  extern __shared__ __attribute__((aligned(16))) float smem[];   // [F][32], zero padded beyond T
  const int b = blockIdx.x, d = (blockIdx.y * 256 + threadIdx.x) * 2;
  for (int i = threadIdx.x; i < 32 * F; i += 256) {
    const int t = i / F, f = i % F;
    smem[f * 32 + t] = t < T ? att[((size_t)t * B + b) * F + f] : 0.f;
  }
  __syncthreads();
  if (d >= ld) return;
  float a0[32], a1[32];
#pragma unroll
  for (int t = 0; t < 32; ++t) { a0[t] = 0.f; a1[t] = 0.f; }
  const bool two = (D & 1) == 0;      // (16-byte rows of an even width: the pair is one 8-byte load)
  if (d < D)
    for (int f = 0; f < F; ++f) {
      const float* er = enc + ((size_t)b * F + f) * D + d;
      float e0, e1;
      if (two) { const float2 e2 = *reinterpret_cast<const float2*>(er); e0 = e2.x; e1 = e2.y; }
      else { e0 = er[0]; e1 = d + 1 < D ? er[1] : 0.f; }
#pragma unroll
      for (int t4 = 0; t4 < 8; ++t4) {
        const f32x4 w = *reinterpret_cast<const f32x4*>(smem + f * 32 + t4 * 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) { a0[t4 * 4 + j] += w[j] * e0; a1[t4 * 4 + j] += w[j] * e1; }
      }
    }
  const float invF = 1.0f / (float)F;
#pragma unroll
  for (int t = 0; t < 32; ++t)
    if (t < T) {
      AT* dst = ctx + ((size_t)t * B + b) * ld + d;      // (ld is even: d + 1 < ld)
      dst[0] = (AT)(a0[t] * invF); dst[1] = (AT)((d + 1 < D ? a1[t] : 0.f) * invF);
    }
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
  constexpr int NV = 20;
  if (V <= NV * 256) {
    // the row stays in registers: one load and one dropout hash per element instead of three (the kernel is bound by the hashes and
    // the exponentials, not by its 78 MB; -6 us on the decoder-only step's gap between the chains); same formulas and summation order
    // as the general form below
    float xs[NV], ms[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int v = tid + 256 * i;
      const float m = v < V ? drop_at(dd, key, t, b, V, v) : 0.f;
      ms[i] = m; xs[i] = v < V ? x[v] * m : -3.0e38f;
    }
    float mx = -3.0e38f;
#pragma unroll
    for (int i = 0; i < NV; ++i) mx = fmaxf(mx, xs[i]);
    mx = block_max256(mx, sm);
    float s = 0.f;
    // (round 5: the exponentials are kept — softmax = e_i / s instead of a second expf per element: the kernel is bound by its
    // transcendentals and dropout hashes, not by its 78 MB)
#pragma unroll
    for (int i = 0; i < NV; ++i) { const float e = tid + 256 * i < V ? expf(xs[i] - mx) : 0.f; s += e; xs[i] = e; }
    s = block_sum256(s, sm);
    const float lse = mx + logf(s);
    const float wgt = cw[t], inv_s = 1.0f / s;
    if (tid == 0) rowloss[row] = wgt * (lse - x[tgt] * drop_at(dd, key, t, b, V, (int)tgt));
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int v = tid + 256 * i;
      if (v < ld) dx[v] = (AT)(v < V ? wgt * (xs[i] * inv_s - (v == tgt ? 1.f : 0.f)) * ms[i] : 0.f);
    }
    for (int v = NV * 256 + tid; v < ld; v += 256) dx[v] = (AT)0.f;
    return;
  }
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

