// The whole recurrent chain of the global reconstructor's forward pass in ONE launch (global_reconstructor.py:43,
// train.py:93-94):   for t: gates = Xg[t] + h_{t-1} . W_hh^T ;  (h_t, c_t) = cell pointwise      (LSTM or GRU)
//
// Every launch of the per-step path streams W_hh (4R x R bf16, 19 MB at R = 1536) from memory again — the L2s of the
// eight XCDs are not coherent and do not keep it between launches — writes split-K slabs and reads them back in a
// second kernel.  Here W_hh is read ONCE: a workgroup owns 8 hidden units (their four gate rows, 32 weight rows) over the
// full contraction and keeps those rows in MFMA B-operand registers for all T steps (R / 8 workgroups x 4 waves x
// <= 128 registers: the chip's register files hold the matrix).  Per step a workgroup only reads the bf16 copy of
// h_{t-1} (B x R), adds the precomputed input part, applies the cell to its 8 units x B rows (c_{t-1} stays in
// registers) and publishes its 8 columns of h_t; steps are separated by a grid-wide barrier (agent-scope release /
// acquire, i.e. L2 write-back + invalidate, around one atomic counter).  All R / 8 <= CU-count workgroups are
// co-resident (one per CU; the host checks), which is what makes the spin barrier safe.
//   * the four waves split K; activations stream through a register ring of PF pairs of k-steps; the four partial
//     [112 x 32] tiles are summed through LDS; 256 threads apply the cell;
//   * the weight rows come from the ordinary gate-major packed image (row gate*R + u): the gather happens once.
#pragma once
#include "common.hpp"

struct RecChainArgs {
  int T, B, R, gru;
  int Bs;                         // rows per time step of the [T][.][.] tensors (= batch size; B = rows of THIS launch: a row group of a batch above 112, pointers pre-offset)
  const bf16_t* W; int ldw;       // [4R][ldw]  packed W_hh, gate-major (GRU: 4-block layout, block 2 zero)
  bf16_t* Hlp; int ld_hlp;        // [T][B][ld_hlp] row-major operand copies of h_t (for the backward's batched GEMMs)
  bf16_t* Pan;                    // [T][R/8][RC_PAN_ROWS][8] k-group-major copies of h_t: what the next step reads
  const float* Xg;                // [T][B][4R] input part of the gates + biases
  float* H; float* C;             // [T][B][R]
  float* acts;                    // [T][B][4R] post-activation gates, for the backward
  unsigned* bar;                  // grid barrier flags, one word per workgroup (never cleared: see rc_epoch_read)
  unsigned* epoch;
  int master;                     // 1: gridDim.x has one extra column; block (gridDim.x - 1, 0) is the barrier master
  float* poison;                  // see rc_give_up
  float* hmean; bf16_t* hmean_lp; int ld_hmean;   // mean_t h_t [B][R] (+ operand copy, zero padded): what the output layer reads
  // Output-layer epilogue (epi = 1; global reconstructor, train.py:96-103): after the last step the launch also computes
  //   out = mean_t h_t . W_o^T + b_o ;  d = out - target ;  partial sums of d^2 ;  dout = gcoef d (fp32) and lp_scale gcoef d (operand copy)
  // one more barrier phase, its weights in the registers W_hh has left — instead of a split-K GEMM, its reduction and the MSE kernel
  // between the two chains.
  int epi;
  const bf16_t* Wo; int ldwo;     // [R][ldwo] packed W_o (row = output column)
  const float* obias;             // [R]
  const float* target;            // [B][R] mean_f enc (rows of this launch)
  float* dout;                    // [B][R] <- gcoef (out - target)
  bf16_t* dout_lp;                // [B][R] <- lp_scale gcoef (out - target)   (leading dimension R: no padding columns)
  bf16_t* PanM;                   // [R/8][RC_PAN_ROWS][8] exchange copy of mean_t h_t
  float* mse_part;                // [workers] partial sums of d^2
  float gcoef, lp_scale;
  // epi = 2: ... and, behind one more barrier phase, d loss / d mean_t h_t = dout . W_o  (what the backward chain starts from)
  const bf16_t* WoT; int ldwot;   // [R][ldwot] W_o^T (row = hidden unit, output column contiguous)
  bf16_t* PanD;                   // [R/8][RC_PAN_ROWS][8] exchange copy of the scaled dout
  float* dhr;                     // [B][R]
};

#define RC_MB 7               // 16-row blocks: B <= 112
// Acquire side of the barriers.  An agent-scope acquire fence is `buffer_inv sc1`: it drops every non-local line of the
// XCD's L2 — under all other workgroups of the chain and under the batched GEMMs of the side stream — 120-190 times
// per time step (measured: 0.12 ms of a 2.29 ms train step).  It is not needed here: (a) every exchange block has its own
// address per time step and nobody touches it before the barrier that publishes it, so no L1 / L2 can hold a line of it
// from an earlier step of this launch; (b) lines from an earlier LAUNCH are dropped by the acquire of the kernel dispatch
// itself (the same mechanism every producer -> consumer pair of kernels on different XCDs relies on); (c) producers write
// through (sc1 stores) and are acknowledged before they arrive at the barrier, so memory holds the data when the
// consumer's first — necessarily missing — load goes out; (d) flags, release words and stamped words are read with sc1
// loads.  RC_ACQUIRE_INV=1 compiles the fences back in (same results on every test).
#ifndef RC_ACQUIRE_INV
#define RC_ACQUIRE_INV 0
#endif
// Layout of the copy of h_t that the chain itself reads back.  An MFMA A-fragment load (16 rows x 32 k, 16 bytes per
// lane) is issued by the texture unit 16 lanes at a time, and in a row-major matrix those 16 lanes are 16 different
// rows = 16 different cache lines for 256 bytes (measured: ~58 clocks per wave load, 8 us per step for the 307 KB
// block).  Stored as [k / 8][row][8] the same 16 lanes read 256 contiguous bytes (two full lines), and a workgroup's
// 8 units x B rows are one contiguous run for the writer.
#define RC_PAN_ROWS (RC_MB * 16)
// elements of one step's panel for a contraction length K (k-groups padded to whole 32-k steps)
__host__ __device__ inline size_t rc_pan_elems(int K) { return (size_t)(((K + 31) >> 5) << 2) * RC_PAN_ROWS * 8; }

// Grid barrier, split in two so that the stores nobody waits for are issued between the halves.
//   arrive: the caller has already waited for its write-through stores of h_t (s_waitcnt vmcnt(0) + __syncthreads);
//           one agent-scope store of the step number into this workgroup's own flag — no read-modify-write, so the
//           arrivals of the R / 8 workgroups do not serialise on one address;
//   wait:   wave 0 polls all flags (one agent-scope load per 64 workgroups) until every one has reached the step, then
//           invalidates this CU's L1 and this XCD's L2 (acquire) so that the next loads of h_t come from memory.
// Launch epoch: one word per chain kernel in device memory, read by every workgroup at the start and incremented by
// workgroup 0 at the end (a workgroup that has passed a barrier knows every other one has started).  Barrier flags are
// epoch << 7 | phase and stamped words carry epoch << 6 | step, so neither needs clearing between launches — in a replayed
// hipGraph each clearing memset was a 6 us node on the critical path.
// Phase stamps (recnet_read_stamps): the launch-epoch words of the six chain kernels are consecutive words at the start of a
// 256-byte line; the line behind it holds one pair of 100 MHz wall-clock stamps per chain — written by workgroup 0 when it starts
// running and when it leaves — so that a REPLAYED hipGraph, which no tracer has to be attached to, reports where its step went
// (prologue / chains / gaps / tail) at the cost of two 8-byte stores per launch.
__device__ __forceinline__ unsigned long long* rc_stamp_slot(const unsigned* epoch) {
  const unsigned long long a = (unsigned long long)epoch;
  return reinterpret_cast<unsigned long long*>((a & ~255ull) + 256ull) + 2 * ((a >> 2) & 7ull);
}
__device__ __forceinline__ unsigned rc_epoch_read(const unsigned* epoch) {
  // (written through: wait_chain_kernel on another XCD polls it)
  if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) __hip_atomic_store(rc_stamp_slot(epoch), (unsigned long long)wall_clock64(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return __hip_atomic_load(epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void rc_epoch_bump(unsigned* epoch, unsigned e) {
  if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) {
    __hip_atomic_store(epoch, e + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    rc_stamp_slot(epoch)[1] = wall_clock64();
  }
}
// Ordering tool for the streams BESIDE a chain: a chain kernel's workgroups need whole CUs (up to 512 registers per lane), so
// ordinary workgroups that are already running when it launches delay its residency by as long as they run.  Work that may run
// beside chain k is therefore enqueued behind this one-wave kernel, which returns once workgroup 0 of chain k has started in the
// CURRENT step (its begin stamp is younger than the step's start stamp) — or after `limit` ticks of the 100 MHz clock: the wait
// only shapes the schedule, nothing depends on it for correctness.
__global__ void wait_chain_kernel(const unsigned long long* step_start, const unsigned long long* chain_begin, unsigned limit) {
  const unsigned long long t0 = wall_clock64();
  for (;;) {
    const unsigned long long s = __hip_atomic_load(step_start, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned long long b = __hip_atomic_load(chain_begin, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (b >= s && b != 0ull) break;
    if (wall_clock64() - t0 > (unsigned long long)limit) break;
    __builtin_amdgcn_s_sleep(8);
  }
}
// Every wait in the chain kernels is bounded: a launch whose workgroups are not all resident (two such launches sharing
// the GPU) would otherwise spin forever and take the device with it.  After ~2^22 polls (seconds) a waiter raises the
// sticky word bar[257]; every wait of this and of later launches then falls through, and the kernel poisons the step's
// total loss with NaN (rc_poison) — wrong loudly instead of hung.
#define RC_SPIN_LIMIT (1u << 22)
// 16 bytes per lane, written through to the agent's coherence point: the instruction an agent-scope relaxed atomic store compiles
// to (global_store_dwordx2 ... sc1), in its 16-byte form — one request per lane instead of two
// (the s_nop 1 = the TWO wait states a VMEM store of more than 8 bytes needs before a VALU instruction may overwrite its data
// registers: the compiler's hazard recognizer inserts `s_nop 1` behind its own dwordx4 stores and does not look inside inline asm —
// without it the next loop iteration's adds corrupted the stored values; `s_nop 0` was one state short)
__device__ __forceinline__ void rc_store16f(float* dst, f32x4 v) {
  asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" :: "v"(dst), "v"(v) : "memory");
}
__device__ __forceinline__ void rc_store16(bf16_t* dst, const bf16_t* src) {
  rc_store16f(reinterpret_cast<float*>(dst), *reinterpret_cast<const f32x4*>(src));
}
__device__ __forceinline__ bool rc_give_up(unsigned* bar, unsigned& spin) {
  if ((++spin & 0x3ffu) != 0) return false;
  if (spin <= RC_SPIN_LIMIT && __hip_atomic_load(bar + 257, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) return false;
  __hip_atomic_store(bar + 257, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return true;
}
__device__ __forceinline__ void rc_poison(unsigned* bar, float* poison) {
  if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0 && __hip_atomic_load(bar + 257, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u)
    *poison = __builtin_nanf("");
}
__device__ __forceinline__ void rc_arrive(unsigned* flags, unsigned step) {
  if (threadIdx.x == 0) __hip_atomic_store(flags + blockIdx.y * gridDim.x + blockIdx.x, step, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void rc_arrive_at(unsigned* flags, int idx, unsigned step) {
  if (threadIdx.x == 0) __hip_atomic_store(flags + idx, step, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void rc_wait(unsigned* flags, unsigned step) {
#ifndef RC_PROBE_NO_BARRIER
  if (threadIdx.x < 64) {
    // n <= 256 flags: four loads per lane, all in flight together (a loop that folds each flag into a running minimum
    // waits for every load before issuing the next one: three memory round trips per poll instead of one)
    const int n = gridDim.x * gridDim.y;
    const unsigned* f0 = flags + (threadIdx.x < n ? threadIdx.x : n - 1);
    const unsigned* f1 = flags + (threadIdx.x + 64 < n ? threadIdx.x + 64 : n - 1);
    const unsigned* f2 = flags + (threadIdx.x + 128 < n ? threadIdx.x + 128 : n - 1);
    const unsigned* f3 = flags + (threadIdx.x + 192 < n ? threadIdx.x + 192 : n - 1);
    unsigned spin = 0;
    for (;;) {
      const unsigned a0 = __hip_atomic_load(f0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned a1 = __hip_atomic_load(f1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned a2 = __hip_atomic_load(f2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned a3 = __hip_atomic_load(f3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      // flags count up across launches (base = launch epoch << 7, see rc_epoch_base): signed distance, wrap-safe
      const bool ok = (int)(a0 - step) >= 0 && (int)(a1 - step) >= 0 && (int)(a2 - step) >= 0 && (int)(a3 - step) >= 0;
      if (__all(ok)) break;
      if (rc_give_up(flags, spin)) break;
      __builtin_amdgcn_s_sleep(1);
    }
#ifndef RC_PROBE_NO_FENCE
    if (RC_ACQUIRE_INV) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
#endif
  }
#endif
  __syncthreads();
}

// Barrier through a master workgroup.  With every workgroup polling every flag, the 130-190 pollers (four wave loads each
// per round, all to the same six lines) queue up at the memory side: the flag round cost 2.7 us, while an uncontended
// store -> load hand-over between two CUs is 0.3-0.4 us (tools/micro/xcd_pingpong.hip).  Here one extra workgroup does
// nothing but poll the arrival flags and, when all have reached a phase, write that phase into eight release words (one
// 128-byte line per XCD-sized group of workgroups); the workers poll only their release word.
__device__ __forceinline__ void rc_master_loop(unsigned* flags, unsigned* release, int n, unsigned fb, int phases) {
  if (threadIdx.x >= 64) return;
  const unsigned* f0 = flags + (threadIdx.x < n ? threadIdx.x : n - 1);
  const unsigned* f1 = flags + (threadIdx.x + 64 < n ? threadIdx.x + 64 : n - 1);
  const unsigned* f2 = flags + (threadIdx.x + 128 < n ? threadIdx.x + 128 : n - 1);
  const unsigned* f3 = flags + (threadIdx.x + 192 < n ? threadIdx.x + 192 : n - 1);
  unsigned spin = 0;
  for (int ph = 1; ph <= phases; ++ph) {
    const unsigned step = fb + (unsigned)ph;
    for (;;) {
      const unsigned a0 = __hip_atomic_load(f0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned a1 = __hip_atomic_load(f1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned a2 = __hip_atomic_load(f2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned a3 = __hip_atomic_load(f3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const bool ok = (int)(a0 - step) >= 0 && (int)(a1 - step) >= 0 && (int)(a2 - step) >= 0 && (int)(a3 - step) >= 0;
      if (__all(ok)) break;
      if (rc_give_up(flags, spin)) break;
    }
    if (threadIdx.x < 8) __hip_atomic_store(release + threadIdx.x * 32, step, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}
// XCD-aware roles of a (unit groups x row parts) grid (round 5).  Workgroup L = blockIdx.y * gridDim.x + blockIdx.x runs on XCD L % 8
// and every XCD fetches what its workgroups read into its own L2.  With the unit group as the fast index every XCD hosts workgroups
// of every row part and pulls the WHOLE h_{t-1} / dG panel across the fabric each step (eight copies of it); here a row part lives on
// 8 / MS XCDs: part = (L % 8) / (8 / MS), unit group = (L / 8) (8 / MS) + L % (8 / MS).  The last workgroup of the grid is the barrier
// master, workgroups left over leave at once.  Falls back to the plain (blockIdx.x, blockIdx.y) roles where the numbers do not fit.
struct RcRole { int vx, vy; bool master, idle; };
__device__ __forceinline__ RcRole rc_role(int nwx, int has_master) {
  const int ms = (int)gridDim.y, total = (int)(gridDim.x * gridDim.y), L = (int)(blockIdx.y * gridDim.x + blockIdx.x);
  RcRole r;
  const int per = ms <= 8 && (8 % ms) == 0 ? 8 / ms : 0;
  if (per > 0 && ((nwx + per - 1) / per) * 8 <= total - (has_master ? 1 : 0)) {
    const int x = L & 7, j = (L >> 3) * per + (x % per);
    r.vy = x / per; r.vx = j;
    r.master = has_master && L == total - 1;
    r.idle = !r.master && (j >= nwx || L >= ((nwx + per - 1) / per) * 8);
    if (r.master || r.idle) { r.vx = 0; r.vy = 0; }
  } else {
    r.vx = (int)blockIdx.x; r.vy = (int)blockIdx.y;
    r.master = has_master && (int)blockIdx.x == nwx && blockIdx.y == 0;
    r.idle = has_master && (int)blockIdx.x == nwx && blockIdx.y != 0;
  }
  return r;
}
__device__ __forceinline__ void rc_wait_release(const unsigned* release, unsigned step) {
#ifndef RC_PROBE_NO_BARRIER
  if (threadIdx.x < 64) {
    const unsigned* r = release + ((blockIdx.y * gridDim.x + blockIdx.x) & 7) * 32;      // the line of this workgroup's XCD
    unsigned spin = 0;
    while ((int)(__hip_atomic_load(r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - step) < 0) { if (rc_give_up(const_cast<unsigned*>(release) - 256, spin)) break; }
    if (RC_ACQUIRE_INV) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
#endif
  __syncthreads();
}

// STEPS = k32-steps per wave (R <= 4 * 32 * STEPS, even); PF = activation prefetch distance in pairs of k-steps.
// RB x CG = 16-row blocks x 16-column groups of a workgroup's tile: it owns UW = 4 CG hidden units (16 CG weight rows)
// and reads RB * 16 of the 112 panel rows; gridDim = (R / UW, MS) with MS * RB * 16 >= 112.  <7, 2>: all rows, 8 units.
// <4, 4>: half of the rows, 16 units — the same number of workgroups and of MFMAs, but each CU pulls 64 instead of 112
// rows of h_{t-1} through its L1 every step, and that read (every CU x the whole block = 59 MB per step out of the
// L2s) is what bounds the step: 7.6 us at 112 rows, 2.5 us at 64 (tools/micro/persist_probe.hip).
template <int STEPS, int PF, int RB, int CG>
__global__ __launch_bounds__(256) void rec_chain_kernel(const RecChainArgs p) {
  constexpr int UW = 4 * CG, ROWS = RB * 16, KG = UW / 8;
  // K partials of the four waves, COLUMN-major (round 6): red[wave][col][RLD] — the four rows a lane holds of an accumulator fragment are
  // contiguous, so a fragment is one ds_write_b128 (16 stores per wave instead of 64), and the cell threads' reads (lane = unit x 4 rows)
  // fall on 64 different banks (4 ul + row mod 64)
  constexpr int NCOL = CG * 16, RLD = ROWS + 4, RED_W = NCOL * RLD;
  extern __shared__ __attribute__((aligned(16))) float rc_smem[];
  float* red = rc_smem;                                            // [4 waves][NCOL][RLD]
  bf16_t* hl = reinterpret_cast<bf16_t*>(rc_smem + 4 * RED_W);   // [ROWS][UW] this step's columns of h_t (16-byte aligned: ROWS % 16 == 0)
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // scalar: every k below is wave-uniform
  const int nwx = (int)gridDim.x - (p.master ? 1 : 0);   // workers per row part
  const RcRole role = rc_role(nwx, p.master);
  const int vx = role.vx, vy = role.vy;
  const int u0 = vx * UW, R = p.R, B = p.B, Bs = p.Bs;
  // rows: this workgroup owns panel rows [own_lo, own_lo + own) and computes [r0, r0 + ROWS) (a superset)
  const int own = RC_PAN_ROWS / gridDim.y, own_lo = vy * own;
  const int r0 = own_lo < RC_PAN_ROWS - ROWS ? own_lo : RC_PAN_ROWS - ROWS;
  const int kw0 = wave * (STEPS * 32);                   // this wave's K range
  const int kq = (lane >> 4) * 8;
  constexpr int NP = STEPS / 2;
  const unsigned ep = rc_epoch_read(p.epoch), fb = ep << 7;
  const int widx = vy * nwx + vx;                        // my flag
  if (role.master) { rc_master_loop(p.bar, p.bar + 256, nwx * (int)gridDim.y, fb, p.T - 1 + p.epi); return; }
  if (role.idle) return;
  const int rot = vx % NP;                               // workgroups start at different k: spreads the L2 channels
  auto k_of = [&](int pr, int hh) { int prr = pr + rot; prr = prr >= NP ? prr - NP : prr; return kw0 + (prr * 2 + hh) * 32; };

  // ---- resident weights: tile column g*16 + c  <->  gate (g*16+c) / UW, unit u0 + (g*16+c) % UW
  bf16x8 wb[STEPS][CG];
#pragma unroll
  for (int g = 0; g < CG; ++g) {
    const int col = g * 16 + (lane & 15), gate = col / UW, ul = col % UW;
    const bf16_t* wrow = p.W + (size_t)(gate * R + u0 + ul) * p.ldw + kq;
#pragma unroll
    for (int pr = 0; pr < NP; ++pr)
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) {
        const int k = k_of(pr, hh);
        wb[pr * 2 + hh][g] = (k + kq < R) ? *reinterpret_cast<const bf16x8*>(wrow + k) : bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
      }
  }
  // ---- this thread's cells: unit u0 + tid % UW, CPT CONSECUTIVE tile rows (tid / UW) CPT + c (round 6: the K partials of a unit's
  // four rows are one 16-byte LDS read per gate and wave; cell = row * UW + unit as before, so cell / UW and cell % UW keep their meaning)
#define RC_CELL(c) (((tid / UW) * CPT + (c)) * UW + tid % UW)
  constexpr int CPT = (ROWS * UW + 255) / 256;
  float xg[CPT][4], cpv[CPT];
  bool mine[CPT];
  float hsum[CPT];                                       // sum_t h_t of this thread's cells (the output layer wants the mean)
#pragma unroll
  for (int c = 0; c < CPT; ++c) {
    const int rg = r0 + RC_CELL(c) / UW;
    mine[c] = rg >= own_lo && rg < own_lo + own && rg < B;
    cpv[c] = 0.f; hsum[c] = 0.f;
  }
  auto load_x = [&](int t) {
    const float* X = p.Xg + (size_t)t * Bs * 4 * R;
#pragma unroll
    for (int c = 0; c < CPT; ++c) {
      const int cell = RC_CELL(c);
      const int row = mine[c] ? r0 + cell / UW : 0;
#pragma unroll
      for (int q = 0; q < 4; ++q) xg[c][q] = X[(size_t)row * 4 * R + q * R + u0 + cell % UW];
    }
  };
  load_x(0);

  const int lane_off = ((lane >> 4) * RC_PAN_ROWS + r0 + (lane & 15)) * 8;   // k-group (lane / 16), row r0 + lane % 16
  const size_t pan_t = rc_pan_elems(R);                                      // elements per time step
  if ((R & 31) && blockIdx.x == 0 && blockIdx.y == 0) {
    // zero the k-groups that pad R to a multiple of 32, in every step's panel: a partly live k-step reads them
    // (against zero weights — but 0 x garbage could be NaN).  Published by the first barrier like h_0.
    const int pad0 = R >> 3, padn = (((R + 31) >> 5) << 2) - pad0;
    for (int t = 0; t < p.T; ++t)
      for (int j = tid; j < padn * RC_PAN_ROWS * 2; j += 256)
        __hip_atomic_store(reinterpret_cast<uint64_t*>(p.Pan + (size_t)t * pan_t + (size_t)pad0 * RC_PAN_ROWS * 8) + j, (uint64_t)0,
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }

  for (int t = 0; t < p.T; ++t) {
    if (t > 0) {
      const bf16_t* A = p.Pan + (size_t)(t - 1) * pan_t + lane_off;
      bf16x8 fa[PF][2][RB];
      auto issue_pair = [&](int slot, int pr) {
#pragma unroll
        for (int hh = 0; hh < 2; ++hh)
#pragma unroll
          for (int i = 0; i < RB; ++i) {
            const int k = k_of(pr, hh);
#ifdef RC_PROBE_SKIP_A
            fa[slot][hh][i] = bf16x8{1, 1, 1, 1, 1, 1, 1, 1}; (void)k;
#else
            // rows >= B of the panel are never written: whatever they hold stays in accumulator rows nobody reads
            // (the k-groups between R and the next multiple of 32 are zeros, see below: the choice is wave-uniform)
            fa[slot][hh][i] = *reinterpret_cast<const bf16x8*>(A + ((k < R ? (k >> 3) : 0) * RC_PAN_ROWS + i * 16) * 8);
#endif
          }
      };
#pragma unroll
      for (int pr = 0; pr < PF; ++pr)
        if (pr < NP) issue_pair(pr, pr);
      __builtin_amdgcn_sched_barrier(0);
      f32x4 acc[RB][CG];
#pragma unroll
      for (int i = 0; i < RB; ++i)
#pragma unroll
        for (int g = 0; g < CG; ++g) acc[i][g] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int pr = 0; pr < NP; ++pr) {
        const int slot = pr % PF;
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
          const int s = pr * 2 + hh;
#pragma unroll
          for (int i = 0; i < RB; ++i)
#pragma unroll
            for (int g = 0; g < CG; ++g)
              acc[i][g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[slot][hh][i], wb[s][g], acc[i][g], 0, 0, 0);
        }
        if (pr + PF < NP) {
          __builtin_amdgcn_sched_barrier(0);
          issue_pair(slot, pr + PF);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      float* part = red + wave * RED_W;
      const int rr = (lane >> 4) * 4, cc = lane & 15;
#pragma unroll
      for (int i = 0; i < RB; ++i)
#pragma unroll
        for (int g = 0; g < CG; ++g) *reinterpret_cast<f32x4*>(part + (g * 16 + cc) * RLD + i * 16 + rr) = acc[i][g];
      __syncthreads();
    }
    // ---- K partials of this thread's cells: per gate and wave the four rows are 16 contiguous bytes of the column-major buffer
    static_assert(CPT == 4, "four consecutive rows per thread: one f32x4 per (gate, wave)");
    f32x4 gsum[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) gsum[q] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (t > 0) {
      const float* rp = red + (tid % UW) * RLD + (tid / UW) * CPT;
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int w = 0; w < 4; ++w) gsum[q] += *reinterpret_cast<const f32x4*>(rp + w * RED_W + q * UW * RLD);
    }
    // ---- cell pointwise for UW units x owned rows
    float hv[CPT], av[CPT][4];
#pragma unroll
    for (int c = 0; c < CPT; ++c) {
      const int cell = RC_CELL(c);
      float g4[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) g4[q] = xg[c][q] + gsum[q][c];
      if (p.gru) {
        const GruOut r = gru_point(g4[0], g4[1], g4[2], g4[3], cpv[c]);
        hv[c] = r.h; av[c][0] = r.r; av[c][1] = r.z; av[c][2] = r.n; av[c][3] = r.hn; cpv[c] = r.h;
      } else {
        const LstmOut r = lstm_point(g4[0], g4[1], g4[2], g4[3], cpv[c]);
        hv[c] = r.h; av[c][0] = r.i; av[c][1] = r.f; av[c][2] = r.g; av[c][3] = r.o; cpv[c] = r.c;
      }
      if (cell < ROWS * UW) hl[cell] = (bf16_t)hv[c];
      hsum[c] += hv[c];
    }
    __syncthreads();
    // the only data another workgroup waits for: 16 bytes per (row, 8 units), written through to memory,
    // acknowledged, then flagged
    const int it_j = tid / own, it_rg = own_lo + tid % own;              // item = (k-group of this workgroup, owned row)
    const bool it_on = tid < KG * own && it_rg < B;
    const bf16_t* it_src = hl + (it_rg - r0) * UW + it_j * 8;
    if (it_on) {
      rc_store16(p.Pan + (size_t)t * pan_t + ((size_t)(vx * KG + it_j) * RC_PAN_ROWS + it_rg) * 8, it_src);
    }
    const bool more = t + 1 < p.T;
    if (more) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      rc_arrive_at(p.bar, widx, fb + (unsigned)(t + 1));
    }
    // ---- everything below is off the critical path of the chain
    float* Ht = p.H + (size_t)t * Bs * R;
    float* Ct = p.C + (size_t)t * Bs * R;
    float* At = p.acts + (size_t)t * Bs * 4 * R;
    bf16_t* Lt = p.Hlp + (size_t)t * Bs * p.ld_hlp;
    if (it_on) *reinterpret_cast<bf16x8*>(Lt + (size_t)it_rg * p.ld_hlp + u0 + it_j * 8) = *reinterpret_cast<const bf16x8*>(it_src);
#ifndef RC_PROBE_SKIP_STORE
#pragma unroll
    for (int c = 0; c < CPT; ++c) {
      const int cell = RC_CELL(c);
      if (mine[c]) {
        const int row = r0 + cell / UW, u = u0 + cell % UW;
        const size_t o = (size_t)row * R + u;
        Ht[o] = hv[c];
        if (!p.gru) Ct[o] = cpv[c];
        float* a = At + (size_t)row * 4 * R + u;
        a[0] = av[c][0]; a[R] = av[c][1]; a[2 * R] = av[c][2]; a[3 * R] = av[c][3];
      }
    }
#endif
    if (blockIdx.x == 0 && blockIdx.y == 0 && p.ld_hlp > R)   // zero padding of the operand copy, columns [R, ld_hlp)
      for (int j = tid; j < B * (p.ld_hlp - R); j += 256) {
        const int row = j / (p.ld_hlp - R), c = R + j % (p.ld_hlp - R);
        Lt[(size_t)row * p.ld_hlp + c] = (bf16_t)0.f;
      }
    if (more) {
      load_x(t + 1);                                     // independent of the other workgroups: in flight across the barrier
      if (p.master) rc_wait_release(p.bar + 256, fb + (unsigned)(t + 1)); else rc_wait(p.bar, fb + (unsigned)(t + 1));
    }
  }
  // mean_t h_t (train.py:96-98 averages the outputs; the output layer is linear, so it runs once on the mean)
  if (p.hmean) {
    const float sc = 1.0f / (float)p.T;
#pragma unroll
    for (int c = 0; c < CPT; ++c) {
      const int cell = RC_CELL(c);
      if (mine[c]) {
        const int row = r0 + cell / UW, u = u0 + cell % UW;
        const float m = hsum[c] * sc;
        p.hmean[(size_t)row * R + u] = m;
        p.hmean_lp[(size_t)row * p.ld_hmean + u] = (bf16_t)m;
      }
    }
    if (blockIdx.x == 0 && blockIdx.y == 0)
      for (int j = tid; j < B * (p.ld_hmean - R); j += 256)
        p.hmean_lp[(size_t)(j / (p.ld_hmean - R)) * p.ld_hmean + R + j % (p.ld_hmean - R)] = (bf16_t)0.f;
  }
  if (p.epi) {
    // ---- output-layer epilogue (UW == 16, master barrier: host-checked).  Publish this workgroup's columns of mean_t h_t like a step ...
    __syncthreads();
    const float sc = 1.0f / (float)p.T;
#pragma unroll
    for (int c = 0; c < CPT; ++c) {
      const int cell = RC_CELL(c);
      if (cell < ROWS * UW) hl[cell] = (bf16_t)(hsum[c] * sc);
    }
    __syncthreads();
    {
      const int it_j = tid / own, it_rg = own_lo + tid % own;
      if (tid < KG * own && it_rg < B)
        rc_store16(p.PanM + ((size_t)(vx * KG + it_j) * RC_PAN_ROWS + it_rg) * 8, hl + (it_rg - r0) * UW + it_j * 8);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    rc_arrive_at(p.bar, widx, fb + (unsigned)p.T);
    // ... W_o rows of its 16 output columns into the registers W_hh has left, the target and the bias while the barrier completes
    bf16x8 wo[STEPS];
    {
      const bf16_t* wrow = p.Wo + (size_t)(u0 + (lane & 15)) * p.ldwo + kq;
#pragma unroll
      for (int ks = 0; ks < STEPS; ++ks) {
        const int k = kw0 + ks * 32;
        wo[ks] = (k + kq < R) ? *reinterpret_cast<const bf16x8*>(wrow + k) : bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
      }
    }
    float tg[CPT], ob[CPT];
#pragma unroll
    for (int c = 0; c < CPT; ++c) {
      const int cell = RC_CELL(c);
      const int row = mine[c] ? r0 + cell / UW : 0, u = u0 + cell % UW;
      tg[c] = p.target[(size_t)row * R + u];
      ob[c] = p.obias[u];
    }
    rc_wait_release(p.bar + 256, fb + (unsigned)p.T);
    {
      const bf16_t* A = p.PanM + lane_off;
      f32x4 acc[RB];
#pragma unroll
      for (int i = 0; i < RB; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
      constexpr int FB = STEPS <= 12 ? STEPS : 8, NFB = (STEPS + FB - 1) / FB;      // panel fragments: FB k-steps in flight (the registers of W_hh are free)
      bf16x8 fa[FB][RB];
#pragma unroll
      for (int b4 = 0; b4 < NFB; ++b4) {
#pragma unroll
        for (int j = 0; j < FB; ++j) {
          const int ks = b4 * FB + j;
          if (ks < STEPS) {
            const int k = kw0 + ks * 32;
#pragma unroll
            for (int i = 0; i < RB; ++i) fa[j][i] = *reinterpret_cast<const bf16x8*>(A + ((k < R ? (k >> 3) : 0) * RC_PAN_ROWS + i * 16) * 8);
          }
        }
#pragma unroll
        for (int j = 0; j < FB; ++j) {
          const int ks = b4 * FB + j;
          if (ks < STEPS) {
#pragma unroll
            for (int i = 0; i < RB; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[j][i], wo[ks], acc[i], 0, 0, 0);
          }
        }
      }
      float* part = red + wave * RED_W;
      const int rr = (lane >> 4) * 4, cc = lane & 15;
#pragma unroll
      for (int i = 0; i < RB; ++i) *reinterpret_cast<f32x4*>(part + cc * RLD + i * 16 + rr) = acc[i];
    }
    __syncthreads();
    float sq = 0.f;
    bf16_t gl[CPT];                       // this thread's cells of the scaled dout (operand copy)
#pragma unroll
    for (int c = 0; c < CPT; ++c) {
      const int cell = RC_CELL(c);
      const int rowl = cell < ROWS * UW ? cell / UW : 0, ul = cell % UW;
      float v = ob[c];
#pragma unroll
      for (int w = 0; w < 4; ++w) v += red[w * RED_W + ul * RLD + rowl];
      gl[c] = (bf16_t)0.f;
      if (mine[c]) {
        const float d = v - tg[c], g = p.gcoef * d;
        sq += d * d;
        const size_t o = (size_t)(r0 + rowl) * R + u0 + ul;
        p.dout[o] = g;
        gl[c] = (bf16_t)(p.lp_scale * g);
        p.dout_lp[o] = gl[c];
      }
    }
    __syncthreads();
    sq = wave_sum(sq);
    if (lane == 0) red[wave] = sq;
    __syncthreads();
    if (tid == 0) p.mse_part[widx] = (red[0] + red[1]) + (red[2] + red[3]);
    if (p.epi > 1) {
      // ---- d loss / d mean_t h_t = dout . W_o: publish this workgroup's 16 columns of dout like a step, then the product over
      // ALL output columns against the W_o^T rows of its 16 hidden units
#pragma unroll
      for (int c = 0; c < CPT; ++c) {
        const int cell = RC_CELL(c);
        if (cell < ROWS * UW) hl[cell] = gl[c];
      }
      __syncthreads();
      {
        const int it_j = tid / own, it_rg = own_lo + tid % own;
        if (tid < KG * own && it_rg < B)
          rc_store16(p.PanD + ((size_t)(vx * KG + it_j) * RC_PAN_ROWS + it_rg) * 8, hl + (it_rg - r0) * UW + it_j * 8);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      rc_arrive_at(p.bar, widx, fb + (unsigned)p.T + 1u);
      bf16x8 wt[STEPS];
      {
        const bf16_t* wrow = p.WoT + (size_t)(u0 + (lane & 15)) * p.ldwot + kq;
#pragma unroll
        for (int ks = 0; ks < STEPS; ++ks) {
          const int k = kw0 + ks * 32;
          wt[ks] = (k + kq < R) ? *reinterpret_cast<const bf16x8*>(wrow + k) : bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
        }
      }
      rc_wait_release(p.bar + 256, fb + (unsigned)p.T + 1u);
      {
        const bf16_t* A = p.PanD + lane_off;
        f32x4 acc[RB];
#pragma unroll
        for (int i = 0; i < RB; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        constexpr int FB = STEPS <= 12 ? STEPS : 8, NFB = (STEPS + FB - 1) / FB;
        bf16x8 fa[FB][RB];
#pragma unroll
        for (int b4 = 0; b4 < NFB; ++b4) {
#pragma unroll
          for (int j = 0; j < FB; ++j) {
            const int ks = b4 * FB + j;
            if (ks < STEPS) {
              const int k = kw0 + ks * 32;
#pragma unroll
              for (int i = 0; i < RB; ++i) fa[j][i] = *reinterpret_cast<const bf16x8*>(A + ((k < R ? (k >> 3) : 0) * RC_PAN_ROWS + i * 16) * 8);
            }
          }
#pragma unroll
          for (int j = 0; j < FB; ++j) {
            const int ks = b4 * FB + j;
            if (ks < STEPS) {
#pragma unroll
              for (int i = 0; i < RB; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[j][i], wt[ks], acc[i], 0, 0, 0);
            }
          }
        }
        float* part = red + wave * RED_W;
        const int rr = (lane >> 4) * 4, cc = lane & 15;
#pragma unroll
        for (int i = 0; i < RB; ++i) *reinterpret_cast<f32x4*>(part + cc * RLD + i * 16 + rr) = acc[i];
      }
      __syncthreads();
#pragma unroll
      for (int c = 0; c < CPT; ++c) {
        const int cell = RC_CELL(c);
        const int rowl = cell < ROWS * UW ? cell / UW : 0, ul = cell % UW;
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) v += red[w * RED_W + ul * RLD + rowl];
        if (mine[c]) p.dhr[(size_t)(r0 + rowl) * R + u0 + ul] = v;
      }
    }
  }
#undef RC_CELL
  rc_epoch_bump(p.epoch, ep);
  rc_poison(p.bar, p.poison);
}
template <int RB, int CG> constexpr size_t rc_smem_bytes() { return (size_t)4 * (CG * 16) * (RB * 16 + 4) * 4 + (size_t)RB * 16 * 4 * CG * 2; }

// =============================================================================================
// The backward chain of the same LSTM / GRU, one launch:
//     for t = T-1 .. 0:  dh_t = dh_direct / T + dG_{t+1} . W_hh ;  (dG_t, dc_{t-1}) = cell pointwise backward
// Same scheme as the forward kernel with the roles of the matrix transposed: a workgroup owns UW = 16 CG OUTPUT units
// (columns of W_hh, read from the K-contiguous image Wt = W_hh^T [R][4R]) over the full contraction K = 4R, keeps them
// in registers, and per step reads the bf16 gate gradients of the step before from the k-group-major exchange buffer
// (4x the forward's: [4R / 8][112][8]), finishes dh for its units, applies the cell backward (dc carry in registers)
// and publishes its 4 x UW gate-gradient columns.  The row-major copy dG [T][B][ld_dg] that the deferred weight-gradient
// GEMMs read is written off the critical path.
struct RecChainBwdArgs {
  int T, B, R, gru;
  int Bs;                          // see RecChainArgs
  const bf16_t* Wt; int ldwt;      // [R][ldwt]  W_hh^T, K (= gate row) contiguous
  bf16_t* Pan;                     // [T][4R/8][RC_PAN_ROWS][8] exchange copies of dG, indexed by chain step
  bf16_t* dG; int ld_dg;           // [T][B][ld_dg] row-major gate gradients, zero padded
  const float* dh_direct; float dh_scale;   // [B][R] the part of d loss / d h_t that is the same for every t
  const float* acts; const float* C; const float* H;   // [T][B][4R], [T][B][R], [T][B][R]
  unsigned* bar; unsigned* epoch; int master; float* poison;
};

// KL > 0: the fragments of the last KL k-steps of every wave live in LDS instead of registers (wl, 16 bytes per thread and
// fragment, conflict-free) - what lets a workgroup own 32 units over 32 rows (CG 2, RB 2, four row parts): the gate-gradient
// panel a workgroup reads every step, ROWS x 4R x 2 bytes through its CU's 64 B/clk L1 fill path, is halved.
template <int STEPS, int PF, int RB, int CG, int KL = 0>
__global__ __launch_bounds__(256) void rec_chain_bwd_kernel(const RecChainBwdArgs p) {
  constexpr int UW = 16 * CG, ROWS = RB * 16, KG = UW / 8, NP = STEPS / 2, KREG = STEPS - KL;
  constexpr int RLD = ROWS + 4, RED_W = UW * RLD;      // K partials column-major, as in rec_chain_kernel (round 6)
  extern __shared__ __attribute__((aligned(16))) float rc_smem[];
  float* red = rc_smem;                                                   // [4 waves][UW][RLD]
  bf16_t* hl = reinterpret_cast<bf16_t*>(rc_smem + 4 * RED_W);            // [ROWS][4][UW]
  bf16x8* wl = reinterpret_cast<bf16x8*>(hl + (size_t)ROWS * 4 * UW);     // [KL][CG][256]
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // scalar: every k below is wave-uniform
  const int nwx = (int)gridDim.x - (p.master ? 1 : 0);
  const RcRole role = rc_role(nwx, p.master);
  const int vx = role.vx, vy = role.vy;
  const int u0 = vx * UW, R = p.R, B = p.B, Bs = p.Bs, K = 4 * R;
  const int own = RC_PAN_ROWS / gridDim.y, own_lo = vy * own;
  const int r0 = own_lo < RC_PAN_ROWS - ROWS ? own_lo : RC_PAN_ROWS - ROWS;
  const int kw0 = wave * (STEPS * 32);
  const int kq = (lane >> 4) * 8;
  const unsigned ep = rc_epoch_read(p.epoch), fb = ep << 7;
  const int widx = vy * nwx + vx;
  if (role.master) { rc_master_loop(p.bar, p.bar + 256, nwx * (int)gridDim.y, fb, p.T - 1); return; }
  if (role.idle) return;
  const int rot = vx % NP;
  auto k_of = [&](int pr, int hh) { int prr = pr + rot; prr = prr >= NP ? prr - NP : prr; return kw0 + (prr * 2 + hh) * 32; };

  bf16x8 wb[KREG > 0 ? KREG : 1][CG];
#pragma unroll
  for (int g = 0; g < CG; ++g) {
    const bf16_t* wrow = p.Wt + (size_t)(u0 + g * 16 + (lane & 15)) * p.ldwt + kq;
#pragma unroll
    for (int pr = 0; pr < NP; ++pr)
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) {
        const int k = k_of(pr, hh), ks = pr * 2 + hh;
        const bf16x8 w = (k + kq < K) ? *reinterpret_cast<const bf16x8*>(wrow + k) : bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
        if (ks < KREG) wb[ks][g] = w; else wl[((ks - KREG) * CG + g) * 256 + tid] = w;
      }
  }
  constexpr int CPT = (ROWS * UW + 255) / 256;
  static_assert(CPT == 4, "four consecutive rows per thread");
  // this thread's cells: unit u0 + tid % UW, rows (tid / UW) CPT + c (cell = row * UW + unit: cell / UW and cell % UW as before)
#define RCB_CELL(c) (((tid / UW) * CPT + (c)) * UW + tid % UW)
  bool mine[CPT];
  float direct[CPT], carry[CPT], av[CPT][4], cc[CPT], cp[CPT];
#pragma unroll
  for (int c = 0; c < CPT; ++c) {
    const int cell = RCB_CELL(c), rg = r0 + cell / UW;
    mine[c] = cell < ROWS * UW && rg >= own_lo && rg < own_lo + own && rg < B;
    direct[c] = mine[c] ? p.dh_scale * p.dh_direct[(size_t)rg * R + u0 + cell % UW] : 0.f;
    carry[c] = 0.f;
  }
  auto prefetch = [&](int t) {                            // saved activations and states of step t
#pragma unroll
    for (int c = 0; c < CPT; ++c) {
      const int cell = RCB_CELL(c);
      const size_t row = mine[c] ? r0 + cell / UW : 0;
      const int u = u0 + cell % UW;
      const float* a = p.acts + ((size_t)t * Bs + row) * 4 * R + u;
#pragma unroll
      for (int q = 0; q < 4; ++q) av[c][q] = a[(size_t)q * R];
      cc[c] = p.gru ? 0.f : p.C[((size_t)t * Bs + row) * R + u];
      cp[c] = t > 0 ? (p.gru ? p.H : p.C)[((size_t)(t - 1) * Bs + row) * R + u] : 0.f;
    }
  };
  prefetch(p.T - 1);

  const int lane_off = ((lane >> 4) * RC_PAN_ROWS + r0 + (lane & 15)) * 8;
  const size_t pan_t = rc_pan_elems(K);

  // what this thread publishes every step: items (gate, k-group, owned row) of 16 bytes
  constexpr int IPT = (4 * KG * (KL ? ROWS : RC_PAN_ROWS) + 255) / 256;   // upper bound: own <= 112 (KL: own <= ROWS)
  const bf16_t* it_src[IPT]; int it_rg[IPT], it_col[IPT]; bool it_on[IPT];
#pragma unroll
  for (int j = 0; j < IPT; ++j) {
    const int idx = tid + j * 256;
    const int q = idx / (KG * own), rem = idx - q * (KG * own), kgi = rem / own, rg = own_lo + rem % own;
    it_on[j] = idx < 4 * KG * own && rg < B;
    it_rg[j] = rg; it_col[j] = q * R + u0 + kgi * 8;
    it_src[j] = hl + ((size_t)(it_on[j] ? rg - r0 : 0) * 4 + (it_on[j] ? q : 0)) * UW + kgi * 8;
  }

  for (int s = 0; s < p.T; ++s) {
    const int t = p.T - 1 - s;
    if (s > 0) {
      const bf16_t* A = p.Pan + (size_t)(s - 1) * pan_t + lane_off;
      bf16x8 fa[PF][2][RB];
      auto issue_pair = [&](int slot, int pr) {
#pragma unroll
        for (int hh = 0; hh < 2; ++hh)
#pragma unroll
          for (int i = 0; i < RB; ++i) {
            const int k = k_of(pr, hh);
            fa[slot][hh][i] = *reinterpret_cast<const bf16x8*>(A + ((k < K ? (k >> 3) : 0) * RC_PAN_ROWS + i * 16) * 8);   // K % 64 == 0
          }
      };
#pragma unroll
      for (int pr = 0; pr < PF; ++pr)
        if (pr < NP) issue_pair(pr, pr);
      __builtin_amdgcn_sched_barrier(0);
      f32x4 acc[RB][CG];
#pragma unroll
      for (int i = 0; i < RB; ++i)
#pragma unroll
        for (int g = 0; g < CG; ++g) acc[i][g] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int pr = 0; pr < NP; ++pr) {
        const int slot = pr % PF;
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
          const int ks = pr * 2 + hh;
#pragma unroll
          for (int i = 0; i < RB; ++i)
#pragma unroll
            for (int g = 0; g < CG; ++g)
              acc[i][g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[slot][hh][i], ks < KREG ? wb[ks < KREG ? ks : 0][g] : wl[((ks - KREG) * CG + g) * 256 + tid],
                                                                  acc[i][g], 0, 0, 0);
        }
        if (pr + PF < NP) {
          __builtin_amdgcn_sched_barrier(0);
          issue_pair(slot, pr + PF);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      float* part = red + wave * RED_W;
      const int rr = (lane >> 4) * 4, cl = lane & 15;
#pragma unroll
      for (int i = 0; i < RB; ++i)
#pragma unroll
        for (int g = 0; g < CG; ++g) *reinterpret_cast<f32x4*>(part + (g * 16 + cl) * RLD + i * 16 + rr) = acc[i][g];
      __syncthreads();
    }
    // ---- cell pointwise backward for UW units x owned rows; the K partials of the thread's four rows: one 16-byte read per wave
    f32x4 dsum = {0.f, 0.f, 0.f, 0.f};
    if (s > 0) {
      const float* rp = red + (tid % UW) * RLD + (tid / UW) * CPT;
#pragma unroll
      for (int w = 0; w < 4; ++w) dsum += *reinterpret_cast<const f32x4*>(rp + w * RED_W);
    }
#pragma unroll
    for (int c = 0; c < CPT; ++c) {
      const int cell = RCB_CELL(c);
      const int row = cell < ROWS * UW ? cell / UW : 0, ul = cell % UW;
      const float dh = direct[c] + dsum[c];
      const LstmGrad g = p.gru ? gru_point_bwd(dh + carry[c], av[c][0], av[c][1], av[c][2], av[c][3], cp[c])
                               : lstm_point_bwd(dh, carry[c], av[c][0], av[c][1], av[c][2], av[c][3], cc[c], cp[c]);
      carry[c] = g.dc_prev;
      if (cell < ROWS * UW) {
        bf16_t* d = hl + (size_t)row * 4 * UW + ul;
        d[0] = (bf16_t)g.di; d[UW] = (bf16_t)g.df; d[2 * UW] = (bf16_t)g.dg; d[3 * UW] = (bf16_t)g.d_o;
      }
    }
    __syncthreads();
    // publish: 16 bytes per (gate, k-group, owned row), written through, acknowledged, then flagged
#pragma unroll
    for (int j = 0; j < IPT; ++j)
      if (it_on[j]) {
        rc_store16(p.Pan + (size_t)s * pan_t + ((size_t)(it_col[j] >> 3) * RC_PAN_ROWS + it_rg[j]) * 8, it_src[j]);
      }
    const bool more = s + 1 < p.T;
    if (more) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      rc_arrive_at(p.bar, widx, fb + (unsigned)(s + 1));
    }
    // ---- off the critical path: the row-major copy for the deferred weight-gradient GEMMs
    bf16_t* Gt = p.dG + (size_t)t * Bs * p.ld_dg;
#pragma unroll
    for (int j = 0; j < IPT; ++j)
      if (it_on[j]) *reinterpret_cast<bf16x8*>(Gt + (size_t)it_rg[j] * p.ld_dg + it_col[j]) = *reinterpret_cast<const bf16x8*>(it_src[j]);
    if (blockIdx.x == 0 && blockIdx.y == 0 && p.ld_dg > K)
      for (int j = tid; j < B * (p.ld_dg - K); j += 256) {
        const int row = j / (p.ld_dg - K), c = K + j % (p.ld_dg - K);
        Gt[(size_t)row * p.ld_dg + c] = (bf16_t)0.f;
      }
    if (more) {
      prefetch(t - 1);
      if (p.master) rc_wait_release(p.bar + 256, fb + (unsigned)(s + 1)); else rc_wait(p.bar, fb + (unsigned)(s + 1));
    }
  }
#undef RCB_CELL
  rc_epoch_bump(p.epoch, ep);
  rc_poison(p.bar, p.poison);
}
template <int RB, int CG, int KL = 0> constexpr size_t rc_bwd_smem_bytes() {
  return (size_t)4 * (16 * CG) * (RB * 16 + 4) * 4 + (size_t)RB * 16 * 4 * 16 * CG * 2 + (size_t)KL * CG * 256 * 16;
}
