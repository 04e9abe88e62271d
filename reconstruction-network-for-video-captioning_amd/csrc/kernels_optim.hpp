// RecNet gfx950 kernels: norms, clipping, multi-tensor Adam with fused weight re-pack.
// Included through kernels.hpp.
#pragma once
// =============================================================================================
// norms, clipping and multi-tensor Adam (train.py:69,103,129,149,186,270-273)
// =============================================================================================
struct TensorDesc { float* p; float* g; float* m; float* v; float* vmax; int n; int chunk0; int nchunks; int pad; };
// Where the packed operand image(s) of a parameter tensor live: element (r, c) of a [rows][cols] tensor goes to
// dst[r * ld + (c - c0)] for every destination whose column window [c0, c0 + nc) contains c.  The Adam kernel
// writes them directly, so the weights are re-packed (bf16) in the same pass that updates them.
// Rows: only source rows [r0, r0 + nr) are written, to destination row (r - r0) (dst is pre-offset) — the GRU's
// 3-block weights land in the 4-block packed layout this way.
// (mode: reserved, 0)
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
// tensor_norm_kernel + norm_finalize_kernel in one launch of one workgroup (<= 16 tensors, a few hundred chunk partials each): the
// per-tensor norms out_norm[i] in the same summation order, then total / clip / sum — one launch less on the step's tail
__global__ __launch_bounds__(256) void norm_all_kernel(const TensorDesc* __restrict__ tab, const float* __restrict__ partial, int ntens,
                                                       float* __restrict__ out_norm, float max_norm, float* total_out, float* clip_out,
                                                       float* sum_out) {
  __shared__ float nrm[16];
  __shared__ int cn[16], c0[16];
  // every thread's partial sums of all tensors first — their loads are independent and in flight together (tensor by tensor, each
  // behind the block reduction of the one before, was sixteen dependent memory round trips: 8 us on the step's tail) —, then the block
  // reductions; the summation order is unchanged
  if (threadIdx.x < 16) { const bool ok = (int)threadIdx.x < ntens; cn[threadIdx.x] = ok ? tab[threadIdx.x].nchunks : 0; c0[threadIdx.x] = ok ? tab[threadIdx.x].chunk0 : 0; }
  __syncthreads();
  float sp[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    float s = 0.f;
    for (int c = threadIdx.x; c < cn[i]; c += 256) s += partial[c0[i] + c];
    sp[i] = s;
  }
  // the sixteen block sums with ONE barrier (round 6: block_sum256 per tensor was 32 barriers on the step's tail): every wave reduces its
  // sixteen values, lane 0 stores them, thread i adds the four wave sums of tensor i in block_sum256's order — the same numbers, bit for bit
  __shared__ float ws[4][16];
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const float w = wave_sum(sp[i]);
    if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6][i] = w;
  }
  __syncthreads();
  if ((int)threadIdx.x < ntens) {
    const int i = threadIdx.x;
    const float s = ws[0][i] + ws[1][i] + ws[2][i] + ws[3][i];
    nrm[i] = sqrtf(s); out_norm[i] = nrm[i];
  }
  __syncthreads();
  if (threadIdx.x != 0) return;
  float ss = 0.f, sn = 0.f;
  for (int i = 0; i < ntens; ++i) { ss += nrm[i] * nrm[i]; sn += nrm[i]; }
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

// (AdamHyper, rn_adam_update: common.hpp)
// torch.optim.Adam (single-tensor form of torch 2.10): g' = clip * (g + reg) + wd * p ;
// m <- lerp(m, g', 1-b1) ; v <- b2 v + (1-b2) g'^2 ; [vmax <- max(vmax, v)] ;
// p <- p - (lr / bc1) * m / (sqrt(v̂) / sqrt(bc2) + eps)
__global__ __launch_bounds__(256) void adam_chunk_kernel(const TensorDesc* __restrict__ tab, const int2* __restrict__ chunks,
                                                         AdamHyper hp, const float* __restrict__ pnorm,
                                                         const float* __restrict__ clip, const int32_t* __restrict__ step_ptr,
                                                         const PackDesc* __restrict__ pack, int lp,
                                                         const float* __restrict__ poison,
                                                         const uint32_t* __restrict__ pending = nullptr, int step_off = 0,
                                                         uint32_t skip_mask = 0u) {
  // A persistent chain kernel of this step gave up waiting (rec_chain.hpp: rc_give_up) and marked the step: its gradients
  // are garbage, so parameters, moments and the packed images stay as they are (the host sees the flag through
  // recnet_chain_status and the NaN total loss).
  if (poison && *poison != 0.f) return;
  // deferred reconstructor update (recnet_flush / the next fused step): nothing to do unless a step left one pending;
  // step_off = -1 when the step counter has already been advanced for the step this launch runs beside
  if (pending && *pending == 0u) return;
  // tensors another launch updates (the split reconstructor update: recurrent weights in the next step, the rest in this one)
  if ((skip_mask >> chunks[blockIdx.x].x) & 1u) return;
  __shared__ float sc[2];
  if (threadIdx.x == 0) {
    const double st = (double)(*step_ptr + step_off);
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
  // one element: the update and the packed image(s)
  auto pack_one = [&](int i, float pn) {
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
  };
  auto upd = [&](float p, float gr, float& m, float& v, float& vmx) -> float {
    return rn_adam_update(p, gr, m, v, vmx, k, cl, hp, step_size, bc2s);
  };
  // The update streams 28 (36 with AMSGrad) bytes per parameter: 16-byte accesses for the whole quads of the chunk (chunks
  // start at multiples of RN_CHUNK elements of 16-byte aligned tensors); a quad that lies inside one row and inside (or
  // outside) every packed window on a 4-column grid is packed with one 8-byte store per image.
  const bool al = (((uintptr_t)td.p | (uintptr_t)td.g | (uintptr_t)td.m | (uintptr_t)td.v | (uintptr_t)(hp.amsgrad ? td.vmax : td.p)) & 15) == 0;
  int i0 = ch.y;
  if (al) {
    const int nq = (end - ch.y) >> 2;
    bool grid4 = (pk.cols & 3) == 0;
    for (int d = 0; d < 6; ++d)
      if (d < pk.ndst && (((pk.d[d].c0 | pk.d[d].nc | pk.d[d].ld) & 3) || pk.d[d].mode)) grid4 = false;
    for (int q = threadIdx.x; q < nq; q += 256) {
      const int i = ch.y + (q << 2);
      const f32x4 p4 = *reinterpret_cast<const f32x4*>(td.p + i), g4 = *reinterpret_cast<const f32x4*>(td.g + i);
      f32x4 m4 = *reinterpret_cast<const f32x4*>(td.m + i), v4 = *reinterpret_cast<const f32x4*>(td.v + i);
      f32x4 x4 = hp.amsgrad ? *reinterpret_cast<const f32x4*>(td.vmax + i) : f32x4{0.f, 0.f, 0.f, 0.f};
      f32x4 n4;
#pragma unroll
      for (int e = 0; e < 4; ++e) { float m = m4[e], v = v4[e], x = x4[e]; n4[e] = upd(p4[e], g4[e], m, v, x); m4[e] = m; v4[e] = v; x4[e] = x; }
      *reinterpret_cast<f32x4*>(td.m + i) = m4; *reinterpret_cast<f32x4*>(td.v + i) = v4;
      if (hp.amsgrad) *reinterpret_cast<f32x4*>(td.vmax + i) = x4;
      *reinterpret_cast<f32x4*>(td.p + i) = n4;
      if (pk.ndst) {
        if (grid4) {
          const int r = i / pk.cols, c = i - r * pk.cols;
#pragma unroll
          for (int d = 0; d < 6; ++d)
            if (d < pk.ndst && c >= pk.d[d].c0 && c < pk.d[d].c0 + pk.d[d].nc && r >= pk.d[d].r0 && r < pk.d[d].r0 + pk.d[d].nr) {
              const size_t o = (size_t)(r - pk.d[d].r0) * pk.d[d].ld + (c - pk.d[d].c0);
              if (lp) {
                bf16x4 hb; hb[0] = (bf16_t)n4[0]; hb[1] = (bf16_t)n4[1]; hb[2] = (bf16_t)n4[2]; hb[3] = (bf16_t)n4[3];
                *reinterpret_cast<bf16x4*>(reinterpret_cast<bf16_t*>(pk.d[d].dst) + o) = hb;
              } else {
                *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(pk.d[d].dst) + o) = n4;
              }
            }
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) pack_one(i + e, n4[e]);
        }
      }
    }
    i0 = ch.y + (nq << 2);
  }
  for (int i = i0 + threadIdx.x; i < end; i += 256) {
    float m = td.m[i], v = td.v[i], x = hp.amsgrad ? td.vmax[i] : 0.f;
    const float pn = upd(td.p[i], td.g[i], m, v, x);
    td.m[i] = m; td.v[i] = v;
    if (hp.amsgrad) td.vmax[i] = x;
    td.p[i] = pn;
    if (pk.ndst) pack_one(i, pn);
  }
}
