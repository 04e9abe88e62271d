// Host-side dispatch of the MFMA GEMM template (gemm.hpp).
#include "gemm_chain.hpp"
#include "launch.hpp"
#include <stdlib.h>
#include <string.h>

namespace {
template <typename CT, typename TA, typename TB>
void launch_layout(const GemmArgs& a, int a_col, int b_col, dim3 grid, hipStream_t st, int tag) {
  // tag > 0: the recurrent-step launches (A = fp32 activations, row layout; the tag fixes B's layout).  Each
  // tag is its own kernel symbol so a rocprofv3 kernel trace separates the dependent-chain GEMMs by site.
  switch (tag) {
    case RN_TAG_DEC_FWD: hipLaunchKernelGGL((gemm_kernel<CT, TA, TB, false, false, RN_TAG_DEC_FWD>), grid, dim3(256), 0, st, a); return;
    case RN_TAG_DEC_BWD: hipLaunchKernelGGL((gemm_kernel<CT, TA, TB, false, true, RN_TAG_DEC_BWD>), grid, dim3(256), 0, st, a); return;
    case RN_TAG_REC_FWD: hipLaunchKernelGGL((gemm_kernel<CT, TA, TB, false, false, RN_TAG_REC_FWD>), grid, dim3(256), 0, st, a); return;
    case RN_TAG_REC_BWD: hipLaunchKernelGGL((gemm_kernel<CT, TA, TB, false, true, RN_TAG_REC_BWD>), grid, dim3(256), 0, st, a); return;
    case RN_TAG_REC_ATT: hipLaunchKernelGGL((gemm_kernel<CT, TA, TB, false, false, RN_TAG_REC_ATT>), grid, dim3(256), 0, st, a); return;
    case RN_TAG_REC_ATT_BWD: hipLaunchKernelGGL((gemm_kernel<CT, TA, TB, false, true, RN_TAG_REC_ATT_BWD>), grid, dim3(256), 0, st, a); return;
    default: break;
  }
  if (!a_col && !b_col) hipLaunchKernelGGL((gemm_kernel<CT, TA, TB, false, false, 0>), grid, dim3(256), 0, st, a);
  else if (!a_col && b_col) hipLaunchKernelGGL((gemm_kernel<CT, TA, TB, false, true, 0>), grid, dim3(256), 0, st, a);
  else if (a_col && !b_col) hipLaunchKernelGGL((gemm_kernel<CT, TA, TB, true, false, 0>), grid, dim3(256), 0, st, a);
  else hipLaunchKernelGGL((gemm_kernel<CT, TA, TB, true, true, 0>), grid, dim3(256), 0, st, a);
}
template <bool ACOL, bool BCOL, int NS, int TAG>
void launch_lds_one(const GemmArgs& a, dim3 grid, hipStream_t st) {
  static bool attr_done = false;
  auto fn = gemm_lds_kernel<ACOL, BCOL, NS, TAG>;
  if (!attr_done) { hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, NS * GL_STAGE_BYTES); attr_done = true; }
  hipLaunchKernelGGL(fn, grid, dim3(256), NS * GL_STAGE_BYTES, st, a);
}
// 96-column workgroups of the row/row form (see gemm_lds.hpp): reconstructor forward chain site
template <int NS, int TAG>
void launch_lds_96(const GemmArgs& a, hipStream_t st) {
  static bool attr_done = false;
  auto fn = gemm_lds_kernel<false, false, NS, TAG, 96>;
  constexpr int lds = NS * (16384 + 96 * 128);
  if (!attr_done) { hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds); attr_done = true; }
  hipLaunchKernelGGL(fn, dim3((a.N + 95) / 96, (a.M + GEMM_TILE - 1) / GEMM_TILE, a.splitk), dim3(256), lds, st, a);
}
template <int NS>
void launch_lds(const GemmArgs& a, int a_col, int b_col, dim3 grid, hipStream_t st, int tag) {
  switch (tag) {
    case RN_TAG_DEC_FWD: launch_lds_one<false, false, NS, RN_TAG_DEC_FWD>(a, grid, st); return;
    case RN_TAG_DEC_BWD: launch_lds_one<false, true, NS, RN_TAG_DEC_BWD>(a, grid, st); return;
    case RN_TAG_REC_FWD: launch_lds_one<false, false, NS, RN_TAG_REC_FWD>(a, grid, st); return;
    case RN_TAG_REC_BWD: launch_lds_one<false, true, NS, RN_TAG_REC_BWD>(a, grid, st); return;
    case RN_TAG_REC_ATT: launch_lds_one<false, false, NS, RN_TAG_REC_ATT>(a, grid, st); return;
    case RN_TAG_REC_ATT_BWD: launch_lds_one<false, true, NS, RN_TAG_REC_ATT_BWD>(a, grid, st); return;
    default: break;
  }
  if (!a_col && !b_col) launch_lds_one<false, false, NS, 0>(a, grid, st);
  else if (!a_col && b_col) launch_lds_one<false, true, NS, 0>(a, grid, st);
  else if (a_col && !b_col) launch_lds_one<true, false, NS, 0>(a, grid, st);
  else launch_lds_one<true, true, NS, 0>(a, grid, st);
}
template <int NG, int NKT, int TAG>
void launch_chain_one(const GemmArgs& a, hipStream_t st) {
  static bool attr_done = false;
  auto fn = gemm_chain_kernel<NG, NKT, TAG>;
  // LDS: NKT activation tiles; the epilogue reuses it for 4 x 32 x (16 NG + 4) floats of staging
  const size_t lds = (size_t)(NKT < 2 ? 2 : NKT) * 16384;
  if (!attr_done) { hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attr_done = true; }
  hipLaunchKernelGGL(fn, dim3((a.N + 64 * NG - 1) / (64 * NG), 1, a.splitk), dim3(256), lds, st, a);
}
template <int NG, int TAG>
void launch_chain_nkt(const GemmArgs& a, int nkt, hipStream_t st) {
  if (nkt <= 2) launch_chain_one<NG, 2, TAG>(a, st);
  else launch_chain_one<NG, 4, TAG>(a, st);
}
template <int NG>
bool launch_chain(const GemmArgs& a, int nkt, hipStream_t st, int tag) {
  switch (tag) {
    case RN_TAG_DEC_FWD: launch_chain_nkt<NG, RN_TAG_DEC_FWD>(a, nkt, st); return true;
    case RN_TAG_DEC_BWD: launch_chain_nkt<NG, RN_TAG_DEC_BWD>(a, nkt, st); return true;   // with the K-contiguous weight image
    case RN_TAG_REC_FWD: launch_chain_nkt<NG, RN_TAG_REC_FWD>(a, nkt, st); return true;
    case RN_TAG_REC_ATT: launch_chain_nkt<NG, RN_TAG_REC_ATT>(a, nkt, st); return true;
    default: return false;
  }
}
inline int vec_ok(const void* p, int ld, int elem) {
  return (((uintptr_t)p) % 16 == 0) && (((size_t)ld * elem) % 16 == 0);
}
}  // namespace

int rn_gemm_bk(int prec) { return prec == RN_PREC_BF16 ? GemmCfg<bf16_t>::BK : GemmCfg<float>::BK; }

// Split-K factor.  chain = 1 (recurrent-step GEMMs, M <= 128): latency matters, spread over ~256 workgroups
// whatever the slice length.  chain = 0 (batched GEMMs): fill one round of 2 workgroups per CU (tiles * s <= 512)
// while every slice keeps >= 5 k-tiles — the rule that matched the measured optimum on every shape of the step
// (scratch/gemm_shapes.py; e.g. 100 tiles x K 6144: s = 4, 460 TF vs 190 TF unsplit).
int rn_pick_splitk(int prec, int M, int N, int K, int max_split, int chain) {
  const int tiles = ((M + GEMM_TILE - 1) / GEMM_TILE) * ((N + GEMM_TILE - 1) / GEMM_TILE);
  const int bk = rn_gemm_bk(prec);
  int nkt = (K + bk - 1) / bk;
  int s = 1;
  if (chain) {
    while (s * 2 <= max_split && tiles * s * 2 <= 320 && nkt / (s * 2) >= 1) s *= 2;
  } else {
    while (s * 2 <= max_split && tiles * s * 2 <= 512 && nkt / (s * 2) >= 5) s *= 2;
  }
  return s;
}

void rn_launch_gemm(int prec, const void* A, int a_bf16, int a_col, int lda, const void* B, int b_bf16, int b_col,
                    int ldb, float* C, int ldc, const float* bias, int M, int N, int K, float alpha,
                    int accumulate, int splitk, float* ws, int reduce_after, hipStream_t st, int tag, int c_bf16, void* c2, int ldc2,
                    const RnMse* mse) {
  if (M <= 0 || N <= 0) return;
  GemmArgs a;
  a.c_bf16 = c_bf16; a.C2 = c2; a.ldc2 = ldc2; a.cnt = nullptr; a.ad_p = nullptr;
  a.mse_ref = nullptr; a.mse_part = nullptr; a.mse_bstride = a.mse_sstride = 0; a.mse_B = 1; a.mse_gcoef = a.mse_lp = 0.f;
  if (mse) {      // (the caller has checked: bf16 operands, N % 4 == 0, 16-byte aligned rows of C / ref, c2 set)
    a.mse_ref = mse->ref; a.mse_part = mse->part; a.mse_bstride = mse->bstride; a.mse_sstride = mse->sstride; a.mse_B = mse->B;
    a.mse_gcoef = mse->gcoef; a.mse_lp = mse->lp;
  }
  if (c_bf16 || c2) splitk = 1;
  a.A = A; a.B = B; a.C = C; a.bias = bias;
  a.M = M; a.N = N; a.K = K; a.lda = lda; a.ldb = ldb; a.ldc = ldc;
  a.alpha = alpha; a.accumulate = accumulate;
  const int bk = rn_gemm_bk(prec);
  if (splitk < 1) splitk = 1;
  int nkt = (K + bk - 1) / bk;
  if (nkt < 1) nkt = 1;
  if (splitk > nkt) splitk = nkt;
  int per = (nkt + splitk - 1) / splitk;
  splitk = (nkt + per - 1) / per;   // no empty slices
  a.splitk = splitk; a.kchunk = per * bk; a.ws = ws;
  a.a_vec = vec_ok(A, lda, a_bf16 ? 2 : 4);
  a.b_vec = vec_ok(B, ldb, b_bf16 ? 2 : 4);
  dim3 grid((N + GEMM_TILE - 1) / GEMM_TILE, (M + GEMM_TILE - 1) / GEMM_TILE, splitk);
  // the site-tagged symbols of the ring / register-staged kernels are instantiated for one operand layout per site
  // (forward sites: B row operand, backward sites: B col operand); any other layout runs the untagged form
  const int tag_bcol = (tag == RN_TAG_DEC_BWD || tag == RN_TAG_REC_BWD || tag == RN_TAG_REC_ATT_BWD) ? 1 : 0;
  const int tag_chain = tag;
  if (tag && (a_col || (b_col != 0) != (tag_bcol != 0))) tag = 0;
  if (prec == RN_PREC_BF16) {
    if (!a_bf16 && !b_bf16) launch_layout<bf16_t, float, float>(a, a_col, b_col, grid, st, tag);
    else if (!a_bf16 && b_bf16) launch_layout<bf16_t, float, bf16_t>(a, a_col, b_col, grid, st, tag);
    else if (a.a_vec && a.b_vec) {
      // forward-form chain launches (activations x weights^T, M <= 128, K slice <= 6 k-tiles): single round trip kernel
      static int chain_on = 1;
      static int chain_ng = 0;
      if (chain_on && tag_chain && !a_col && !b_col && M <= 128 && per <= GC_MAX_KT && !c_bf16 && !c2) {
        // 128-column workgroups unless that leaves half the chip idle (e.g. the decoder's N = 4H + A = 2176)
        const int ng = chain_ng ? chain_ng : ((((N + 127) / 128) * splitk >= 128) ? 2 : 1);
        const bool done = ng == 1 ? launch_chain<1>(a, per, st, tag_chain) : launch_chain<2>(a, per, st, tag_chain);
        if (done) goto after_launch;
      }
      // both operands bf16 in memory: the DMA-staged ring kernel.  Chain launches (tag > 0) run ~1 block
      // per CU and want the deepest ring; batched GEMMs trade ring depth for 2 resident blocks per CU.
      static const int ns_chain = 4;
      static int ns_batch = 2;
      const int ns = tag ? ns_chain : ns_batch;
      {
        // chain launch, row/row: 96-column workgroups when that turns a partial wave of workgroups into one per CU
        static int bn96 = 1;
        const int t128 = ((N + 127) / 128) * ((M + 127) / 128) * splitk, t96 = ((N + 95) / 96) * ((M + 127) / 128) * splitk;
        if (bn96 && tag == RN_TAG_REC_FWD && !a_col && !b_col && ns >= 4 && N % 96 == 0 && t128 < 224 && t96 <= 256) {
          launch_lds_96<4, RN_TAG_REC_FWD>(a, st);
          goto after_launch;
        }
      }
      if (ns >= 4) launch_lds<4>(a, a_col, b_col, grid, st, tag);
      else if (ns == 3) launch_lds<3>(a, a_col, b_col, grid, st, tag);
      else launch_lds<2>(a, a_col, b_col, grid, st, tag);
    } else launch_layout<bf16_t, bf16_t, bf16_t>(a, a_col, b_col, grid, st, 0);   // unaligned bf16 operands
  } else {
    launch_layout<float, float, float>(a, a_col, b_col, grid, st, tag);
  }
after_launch:
  if (splitk > 1 && reduce_after) {
    size_t total = (size_t)M * N;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, st, ws, splitk, M, N, C, ldc, bias, alpha,
                       accumulate);
  }
}

// effective split count rn_launch_gemm will use (the fused consumers need it to sum the slabs)
int rn_effective_splitk(int prec, int K, int splitk) {
  const int bk = rn_gemm_bk(prec);
  int nkt = (K + bk - 1) / bk;
  if (nkt < 1) nkt = 1;
  if (splitk < 1) splitk = 1;
  if (splitk > nkt) splitk = nkt;
  int per = (nkt + splitk - 1) / splitk;
  return (nkt + per - 1) / per;
}

// ---------------------------------------------------------------------------------------------- grouped launch
#include <algorithm>
#include <queue>
#include <vector>
namespace {
template <bool ACOL, bool BCOL, bool EPI = false>
void launch_group_one(const GemmGroupArgs& g, int nblocks, hipStream_t st) {
  static bool attr_done = false;
  auto fn = gemm_group_kernel<ACOL, BCOL, 2, EPI>;
  if (!attr_done) { hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * GL_STAGE_BYTES); attr_done = true; }
  hipLaunchKernelGGL(fn, dim3(nblocks), dim3(256), 2 * GL_STAGE_BYTES, st, g);
}
// finish time of greedy list scheduling (items in the given order, each to the slot that frees up first) on `slots` slots:
// what the hardware dispatcher does with the grid
double list_makespan(const std::vector<std::pair<double, int>>& runs, int slots) {      // (length, count), queue order
  std::priority_queue<double, std::vector<double>, std::greater<double>> q;
  for (int i = 0; i < slots; ++i) q.push(0.0);
  double end = 0.0;
  for (auto& r : runs)
    for (int c = 0; c < r.second; ++c) { double t = q.top(); q.pop(); t += r.first; q.push(t); if (t > end) end = t; }
  return end;
}
}  // namespace

int rn_launch_gemm_group(int a_col, int b_col, const RnGemmDesc* d, int n, float* ws, size_t ws_floats, unsigned* cnt, int cnt_words,
                         hipStream_t st, int slots_hint, const AdamShared* adam, unsigned long long* stamp) {
  if (n < 1 || n > GG_MAX) return 1;      // (RN_GEMM_GROUP=0 is a switch of the handle: host_common.inc)
  struct Prob { int idx, tiles, nkt, s; };
  std::vector<Prob> pr;
  for (int i = 0; i < n; ++i) {
    if (d[i].M <= 0 || d[i].N <= 0 || d[i].K <= 0) continue;
    if (!vec_ok(d[i].A, d[i].lda, 2) || !vec_ok(d[i].B, d[i].ldb, 2)) return 1;
    Prob q; q.idx = i; q.s = 1; q.nkt = (d[i].K + 63) / 64;
    q.tiles = ((d[i].M + GEMM_TILE - 1) / GEMM_TILE) * ((d[i].N + GEMM_TILE - 1) / GEMM_TILE);
    pr.push_back(q);
  }
  if (pr.empty()) return 0;
  // split factors: coordinate descent on the estimated finish time (k-tile units; every item pays ~3 units of pipeline fill
  // and epilogue, a split product's slabs ~1 unit per slice more for the slice that sums them), 2 workgroups per CU
  static int slots_all = 0;
  if (!slots_all) { int dev = 0, ncu = 256; hipGetDevice(&dev); hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev); slots_all = 2 * (ncu > 0 ? ncu : 256); }
  const int slots = slots_hint > 0 ? slots_hint : slots_all;
  auto per_slice = [](const Prob& q) { const int per = (q.nkt + q.s - 1) / q.s; return per; };
  auto estimate = [&]() {
    std::vector<std::pair<double, int>> runs;
    std::vector<const Prob*> o;
    for (auto& q : pr) o.push_back(&q);
    std::sort(o.begin(), o.end(), [&](const Prob* a, const Prob* b) { return per_slice(*a) > per_slice(*b); });
    for (auto* q : o) {
      const int per = per_slice(*q), s_eff = (q->nkt + per - 1) / per;
      runs.push_back({per + 3.0 + (s_eff > 1 ? 0.5 * s_eff : 0.0), q->tiles * s_eff});
    }
    return list_makespan(runs, slots);
  };
  const bool can_split = ws && cnt;
  for (int iter = 0; iter < 32 && can_split; ++iter) {
    const double base = estimate();
    int best = -1; double best_t = base * 0.97;      // a split has to pay for its slabs
    for (size_t i = 0; i < pr.size(); ++i) {
      Prob& q = pr[i];
      if (d[q.idx].c2 || d[q.idx].ad_p || (d[q.idx].N & 3) || q.s >= 16 || q.nkt / (q.s + 1) < 4) continue;      // (Adam epilogue: whole K in one workgroup)
      const int keep = q.s;
      q.s = keep + 1;
      const double t = estimate();
      q.s = keep;
      if (t < best_t) { best_t = t; best = (int)i; }
    }
    if (best < 0) break;
    pr[best].s += 1;
  }
  std::sort(pr.begin(), pr.end(), [&](const Prob& a, const Prob& b) { return per_slice(a) > per_slice(b); });
  GemmGroupArgs g;
  bool any_epi = false;
  if (adam) g.ad = *adam; else memset(&g.ad, 0, sizeof(g.ad));
  g.stamp = stamp;
  g.np = (int)pr.size();
  int blocks = 0; size_t ws_off = 0; int cnt_off = 0;
  for (int k = 0; k < g.np; ++k) {
    const RnGemmDesc& e = d[pr[k].idx];
    GemmArgs& a = g.p[k];
    a.A = e.A; a.B = e.B; a.C = e.C; a.bias = e.bias; a.M = e.M; a.N = e.N; a.K = e.K; a.lda = e.lda; a.ldb = e.ldb; a.ldc = e.ldc;
    a.alpha = e.alpha; a.accumulate = e.accumulate; a.c_bf16 = e.c_bf16; a.C2 = e.c2; a.ldc2 = e.ldc2; a.a_vec = 1; a.b_vec = 1;
    a.ad_p = adam ? e.ad_p : nullptr; a.ad_m = e.ad_m; a.ad_v = e.ad_v; a.ad_vmax = e.ad_vmax; a.ad_img = e.ad_img; a.ad_imgt = e.ad_imgt;
    a.ad_pnorm = e.ad_pnorm; a.ad_ld_img = e.ad_ld_img; a.ad_ld_imgt = e.ad_ld_imgt;
    if (a.ad_p) any_epi = true;
    const int per = per_slice(pr[k]);
    int s = (pr[k].nkt + per - 1) / per;
    const size_t slab = (size_t)e.M * e.N;
    if (s > 1 && (ws_off + (size_t)s * slab > ws_floats || cnt_off + pr[k].tiles > cnt_words || (size_t)s * slab * 4 >= ((size_t)1 << 31))) s = 1;
    a.splitk = s; a.kchunk = (s > 1 ? per : pr[k].nkt) * 64;
    a.ws = nullptr; a.cnt = nullptr; a.mse_ref = nullptr; a.mse_part = nullptr;
    if (s > 1) { a.ws = ws + ws_off; a.cnt = cnt + cnt_off; ws_off += (size_t)s * slab; cnt_off += pr[k].tiles; }
    g.first[k] = blocks;
    blocks += pr[k].tiles * s;
  }
  for (int k = g.np; k <= GG_MAX; ++k) g.first[k] = blocks;
  if (any_epi) {
    if (!(a_col && b_col)) return 1;                       // (the fused update exists for the dY^T . X form only)
    launch_group_one<true, true, true>(g, blocks, st);
    return 0;
  }
  if (!a_col && !b_col) launch_group_one<false, false>(g, blocks, st);
  else if (!a_col && b_col) launch_group_one<false, true>(g, blocks, st);
  else if (a_col && !b_col) launch_group_one<true, false>(g, blocks, st);
  else launch_group_one<true, true>(g, blocks, st);
  return 0;
}
