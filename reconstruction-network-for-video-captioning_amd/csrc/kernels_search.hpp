// RecNet gfx950 kernels: device-side greedy / beam search.
// Included through kernels.hpp.
#pragma once
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

