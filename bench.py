#!/usr/bin/env python3
"""Benchmark of the RecNet train step (train.py:248-273) on MI355X.

  python bench.py --gpus N --steps K --warmup W        (N > 1: launched by torch.distributed.run)

Workload (BASELINE.json configs[1]): decoder + GLOBAL reconstructor, B=100 captions per GPU, 28x1536
features, V=4188, E=468, H=512, A=128, R=1536, LSTM/LSTM, 30-token cap with caption 0 at full length
(T = 31 decoder steps), dropout 0.5 active, clip 50, AMSGrad/Adam — bf16 MFMA operands, fp32
accumulate/state.  Synthetic features, random-init weights.  Weak scaling: every rank keeps B=100,
gradients are SUM-all-reduced over RCCL.  One JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def build_models(R, cfg_over, V):
    import torch
    C = R.make_config(**cfg_over)
    torch.manual_seed(0)
    dec = R.build_decoder(V, C)
    rec = R.build_reconstructor(C) if C.use_recon else None
    return C, dec, rec


def cpu_baseline(kind, B, F, D, V, steps, warmup, cell="LSTM"):
    """The oracle (CPU port of the reference algorithm, oracle/recnet_oracle.py) timed on this host."""
    import torch
    from oracle import recnet_oracle as O
    O.RNN_IMPL = "aten"       # the fused one-step RNN op nn.LSTM dispatches to: the reference's own CPU cost structure
    # small per-step ops: more than ~32 threads only adds synchronisation cost to the CPU port
    torch.set_num_threads(max(1, min(32, os.cpu_count() or 1)))
    torch.manual_seed(0)
    decP = O.init_decoder_params(V, D=D, cell=cell)
    recP = O.init_rec_params(kind, R=D, cell=cell) if kind else None
    st = O.TrainState(decP, recP, kind, cell=cell, rec_cell=cell)
    enc, targets, masks = O.synthetic_batch(B, F, D, V)
    drop = O.Dropper("rng")
    ts = []
    for i in range(warmup + steps):
        t0 = time.perf_counter()
        st.step(enc, targets, masks, drop)
        if i >= warmup:
            ts.append(time.perf_counter() - t0)
    ts.sort()
    med = ts[len(ts) // 2]
    return B / med, torch.get_num_threads(), med


def algorithmic_flops(kind, B, F, D, T, V=4188, E=468, H=512, A=128, RA=128):
    """loop invariants counted once, multiply-add = 2, backward = 2 x forward (train = 3 x)."""
    R = D
    per = 2 * F * D * A + T * (2 * H * A + 2 * F * A + 2 * F * D + 8 * H * (E + D + H) + 2 * H * V)
    if kind == "local":
        per += 2 * T * H * RA + F * (2 * R * RA + 2 * T * RA + 2 * T * H + 8 * R * (H + R) + 2 * R * R)
    elif kind == "global":
        per += T * (8 * R * (2 * H + R) + 2 * R * R)
    return 3.0 * per * B


def algorithmic_hbm_bytes(kind, B, F, D, V=4188, E=468, H=512, A=128, RA=128):
    """optimiser + regulariser traffic (36 B / parameter with AMSGrad, 28 B without) + the inputs."""
    R = D
    p_dec = A + V * E + A * H + A * D + A + 4 * H * (E + D) + 4 * H * H + 8 * H + V * H + V
    p_rec = 0
    if kind == "local":
        p_rec = RA + RA * R + RA * H + RA + 4 * R * H + 4 * R * R + 8 * R + R * R + R
    elif kind == "global":
        p_rec = 4 * R * 2 * H + 4 * R * R + 8 * R + R * R + R
    return 36.0 * p_dec + 28.0 * p_rec + 4.0 * B * F * D + 8.0 * 31 * B


def stored_traffic(kernel_full, kind, dims, T, cell):
    """HBM-side bytes per launch from the rocprofv3 --pmc passes stored under profiles/ (tools/pmc_traffic.py; counters
    cannot be read from inside the timed process): the newest round's file whose metadata names exactly this kernel,
    workload shape and cell.  Returns (bytes, source) or (None, None)."""
    import glob
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_pmc_traffic_*.json"))):
        try:
            meta = json.load(open(f))
        except Exception:
            continue
        if (meta.get("kernel", "") in (kernel_full, kernel_full.split("<")[0]) and meta.get("kind") == (kind or "none") and meta.get("B") == dims["B"] and
                meta.get("D") == dims["D"] and meta.get("F") == dims["F"] and meta.get("T", 31) == T and meta.get("cell", "LSTM") == cell):
            best = (int(meta["traffic_bytes_per_launch"]), "stored: profiles/" + os.path.basename(f))
    return best or (None, None)


def roofline(eng, run_step, kind, precision, iters=5, T=31, cell="LSTM"):
    """Dominant kernel = the kernel of the recurrent part with the largest TIME PER TRAIN STEP (launches x average duration):
    one of the persistent chain kernels (one launch = all T / F dependent steps of a chain, weights resident on chip) or,
    where a chain runs on per-step kernels (R = 3584, B > 112, the fp32 path), its recurrent-step GEMM (28-31 launches per
    step).  `achieved` = algorithmic bytes of one launch (recnet_recurrent_step_bytes) / its average duration, measured
    with hipEvents on the launch stream; bound = HBM, the bound SURVEY.md section 8d names for the recurrent kernels.
    run_step(site) -> (launches per step, avg ms per launch)."""
    import torch
    torch.cuda.synchronize()
    local = kind == "local"
    # (site, `which` of recnet_recurrent_step_bytes, name)
    cands = [(9, 5, "dec_chain_kernel (decoder forward chain, T steps in one launch)"),
             (10, 6, "dec_chain_bwd_kernel (decoder BPTT chain, T steps in one launch)"),
             (8, 4, ("lcbig_bwd_kernel (local reconstructor backward chain for R > 2048: three phases per step, F steps in one launch)"
                     if eng.dims["D"] > 2048 else "loc_chain_bwd_kernel (local reconstructor backward chain, F steps in one launch)") if local else
                    "rec_chain_bwd_kernel (reconstructor backward chain, T steps in one launch)"),
             (7, 3, "loc_chain_kernel (local reconstructor forward chain, F steps in one launch)" if local else
                    "rec_chain_kernel (reconstructor forward chain, T steps in one launch)"),
             (1, 0, "gemm_chain_kernel (recurrent-step GEMM, decoder fwd, one launch per step)"),
             (3, 1, "gemm_lds_kernel<false, false, 4, 3, 96> (recurrent-step GEMM, reconstructor fwd, one launch per step)"),
             (4, 2, "gemm_lds_kernel<false, true, 4, 4, 128> (recurrent-step GEMM, reconstructor bwd, one launch per step)")]
    chains, per_step, meas, covered = {}, {}, [], set()
    for s_id, wh, nm in cands:
        if wh in (1, 2, 3, 4) and kind is None:
            continue
        if s_id in covered:                      # the chain kernel of this recurrence ran: its per-step GEMM site has no launches
            continue
        n_, ms_ = run_step(s_id)
        if n_ > 0:
            short = nm.split(" ")[0].split("<")[0]
            if s_id >= 7:
                chains[short] = round(ms_ * 1e3, 1)
                covered.add({9: 1, 10: 2, 7: 3, 8: 4}[s_id])
            per_step["site %d: %s" % (s_id, nm.split(" (")[0])] = {"launches_per_step": n_, "avg_us": round(ms_ * 1e3, 1),
                                                                    "us_per_step": round(n_ * ms_ * 1e3, 1)}
            meas.append((n_, ms_, s_id, wh, nm))
    if not meas:
        return None
    ranked = sorted(meas, key=lambda m: -m[0] * m[1])
    n, ms_raw, site, which, kname = ranked[0]                 # the true maximum; the runner-up is listed beside it
    runner_up = ranked[1][4].split(" (")[0] if len(ranked) > 1 else None
    # What the two event records add to a bracket (E), from brackets around 1 and around 17 empty kernels in the same
    # mode (eager launches): b(c) = E + c * f.  The kernel's dispatch-to-completion time — what rocprofv3 reports as
    # its duration — is its bracket minus E.
    _, b1 = run_step(-1)
    _, b17 = run_step(-17)
    f_empty = max((b17 - b1) / 16.0, 0.0)
    ms_null = max(b1 - f_empty, 0.0)
    ms = max(ms_raw - ms_null, 1e-6)
    bytes_launch = eng.recurrent_step_bytes(which)
    exchange = eng.chain_exchange_bytes(which) if which >= 3 else 0.0
    achieved = bytes_launch / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
    peak = 8000.0
    traffic, traffic_src = stored_traffic(kname.split(" (")[0], kind, eng.dims, T, cell) if precision == "bf16" else (None, None)
    return {"bound": "hbm", "achieved": round(achieved, 1), "peak": peak, "unit": "GB/s",
            "frac": round(achieved / peak, 4), "traffic": traffic, "traffic_source": traffic_src,
            "kernel": kname, "runner_up": runner_up, "launches_per_step": n, "us_per_step": round(n * ms * 1e3, 1), "exchange_bytes_per_launch": int(exchange),
            "avg_launch_us": round(ms * 1e3, 3), "bracket_us": round(ms_raw * 1e3, 3),
            "event_pair_overhead_us": round(ms_null * 1e3, 3), "empty_kernel_us": round(f_empty * 1e3, 3), "algorithmic_bytes_per_launch": int(bytes_launch),
            "chain_kernel_brackets_us": chains, "recurrent_kernels": per_step}


def phase_table(eng, run, replays=12):
    """Untraced decomposition of the replayed step: the chain kernels stamp the 100 MHz wall clock when their first workgroup
    starts and when it leaves, the step's first / last kernels stamp its start / end (Engine.read_stamps) — medians over
    `replays` replays, microseconds.  prologue = start -> first chain; gaps = between consecutive chains; tail = last chain ->
    end of the step's last kernel.  (between_steps and the back-to-back span are added by the caller from Engine.step_ring.)"""
    import statistics
    rows = []
    burst = int(os.environ.get("RN_PHASE_BURST", "1"))      # (diagnostic: stamps of the LAST of `burst` back-to-back replays)
    for _ in range(replays):
        for _ in range(burst):
            run()
        rows.append(eng.read_stamps())
    rows = [r for r in rows if r["end"] and r["chains"]]
    if not rows:
        return None
    order = sorted(rows[0]["chains"], key=lambda k: rows[0]["chains"][k][0])
    med = lambda xs: round(statistics.median(xs), 1)
    out = {"prologue_us": med([r["chains"][order[0]][0] for r in rows])}
    for i, nm in enumerate(order):
        out["chain_%s_us" % nm] = med([r["chains"][nm][1] - r["chains"][nm][0] for r in rows])
        if i + 1 < len(order):
            out["gap_%s_to_%s_us" % (nm, order[i + 1])] = med([r["chains"][order[i + 1]][0] - r["chains"][nm][1] for r in rows])
    # grouped GEMM launches (first workgroup started, last workgroup left), offsets from the step's start
    for nm in rows[0].get("groups", {}):
        if all(nm in r["groups"] for r in rows):
            out["group_%s_us" % nm] = [med([r["groups"][nm][0] for r in rows]), med([r["groups"][nm][1] for r in rows])]
    out["tail_us"] = med([r["end"] - r["chains"][order[-1]][1] for r in rows])
    out["span_us"] = med([r["end"] for r in rows])
    out["chain_offsets_us"] = {nm: [med([r["chains"][nm][0] for r in rows]), med([r["chains"][nm][1] for r in rows])] for nm in order}
    out["outside_chains_us"] = round(out["span_us"] - sum(v for k, v in out.items() if k.startswith("chain_") and isinstance(v, float)), 1)
    return out


def fp32_exact(R, cfg_over, V, enc, targets, targets_g, B, F, steps=10, warmup=3):
    """The same workload on the exact-fp32 path (`precision="f32"`: fp32 operands, v_mfma_f32_16x16x4_f32, the arithmetic of
    the reference — the path the 1e-4 gradient parity bar is held on), timed the same way; reported beside the bf16 headline."""
    import torch
    C, dec, rec = build_models(R, dict(cfg_over, precision="f32"), V)
    step = R.DataParallelTrainStep(dec, rec, B, 0, 1, n_frames=F)
    T, w = step.prepare(targets_g.numpy())
    run = R.GraphedStep(step, enc, targets, T, w)
    for _ in range(warmup):
        run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        run()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    del run, step
    return {"ms_per_step": round(ms, 4), "value": round(B * 1e3 / ms, 1), "unit": "captions/s", "dtype": "f32", "steps": steps, "warmup": warmup,
            "note": "same workload, precision=f32 (fp32 MFMA operands); the parity path of TOL['f32'] in tests/gpu_util.py"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=100, help="captions per GPU")
    ap.add_argument("--rec", default="global", choices=["global", "local", "none"])
    ap.add_argument("--precision", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--frames", type=int, default=28, help="encoder_output_len F (C4: 40)")
    ap.add_argument("--feat", type=int, default=1536, help="encoder_output_size D = reconstructor size (C4: 2048, C5: 3584)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-fp32-exact", action="store_true", help="skip the fp32_exact sub-record (the same workload on the exact-fp32 path)")
    ap.add_argument("--cpu-steps", type=int, default=3)
    ap.add_argument("--graph", type=int, default=1, help="replay the step from a captured hipGraph")
    ap.add_argument("--defer", type=int, default=2, choices=[0, 1, 2], help="one rank, graph mode — 2 (default): SPLIT reconstructor update: d W_hh and its "
                    "Adam step of step n run under the decoder forward chain of step n + 1 (119 CUs idle there), everything else inside "
                    "step n; 1: the whole update deferred (global reconstructor; measured slower); 0: everything inside the step.  The "
                    "last step's pending half is flushed INSIDE the timed region: every timed step's work is timed")
    ap.add_argument("--cell", default="LSTM", choices=["LSTM", "GRU"], help="recurrent cell of decoder and reconstructor "
                    "(the north-star workload is LSTM; GRU is config.py:31's literal default)")
    ap.add_argument("--lengths", default="uniform", choices=["uniform", "msvd"], help="caption lengths: the benchmark's "
                    "U{4..30} with one full-length caption (T = 31), or MSVD-like 3 + Poisson(5) (the loop exits early)")
    ap.add_argument("--force-allreduce", action="store_true", help="keep the gradient all-reduce (and the three-graph step "
                    "built around it) with a single rank too: exercises the RCCL path on one GPU")
    ap.add_argument("--global-batch", type=int, default=0, help="STRONG scaling: this many captions in total, sharded over the "
                    "ranks (BASELINE configs[3]: 256 over 8, configs[4]: 512 over 8); default 0 = weak scaling, --batch per rank")
    ap.add_argument("--grad-dtype", default="f32", choices=["f32", "bf16"], help="gradient transport of the all-reduce")
    ap.add_argument("--feed", type=int, default=0, help="1: every step takes a fresh HOST batch through feed.DeviceFeeder "
                    "(pinned staging + H2D on a side stream); reports the PCIe-inclusive rate, not the headline value.  1: the batch "
                    "is staged by the calling thread (measured faster: 1.70 against 1.81 ms), 2: by the feeder's worker thread.  "
                    "Every slot of the feeder has a captured graph of its own; at least 40 untimed warm-up steps are run so that each "
                    "has been replayed a few times (the first replays of an executable graph are slow: 1.87 against 1.70 ms over "
                    "50 timed steps with 10 warm-up steps)")
    args = ap.parse_args()

    import torch
    import recnet_amd as R
    from recnet_amd.synthetic import synthetic_features, synthetic_targets

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit("--gpus %d needs torch.distributed.run with %d ranks (WORLD_SIZE=%d)" % (args.gpus, args.gpus, world))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    under_launcher = "RANK" in os.environ          # torch.distributed.run, any world size
    if world > 1 or under_launcher:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        dist.init_process_group(os.environ.get("RN_DIST_BACKEND", "nccl"), rank=rank, world_size=world, device_id=dev)

    B, F, D, V = args.batch, args.frames, args.feat, 4188
    if args.global_batch:
        B = R.shard_bounds(args.global_batch, world, 0)[1]          # the largest shard sizes the engines
    kind = None if args.rec == "none" else args.rec
    cfg_over = dict(batch_size=B, use_recon=kind is not None, reconstructor_type=kind or "global",
                    encoder_output_len=F, encoder_output_size=D, reconstructor_hidden_size=D,
                    precision=args.precision, device=str(dev), decoder_model=args.cell, reconstructor_model=args.cell)
    C, dec, rec = build_models(R, cfg_over, V)
    Bg = args.global_batch if args.global_batch else B * world
    targets_g = synthetic_targets(Bg, V, seed=1234, lengths=args.lengths)
    lo, hi = R.shard_bounds(Bg, world, rank)
    enc = synthetic_features(hi - lo, F, D, seed=1234 + rank).to(dev)
    targets = targets_g[:, lo:hi].contiguous().to(dev)
    step = R.DataParallelTrainStep(dec, rec, Bg, rank, world, n_frames=F, always_reduce=args.force_allreduce and under_launcher,
                                   grad_dtype=args.grad_dtype)
    T, w = step.prepare(targets_g.numpy())

    def sync_all():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    runner = lambda: step(enc, targets, T, w)
    graphed = None
    if args.graph:
        # one rank: the reconstructor's weight-gradient products + Adam step of replay n run under the decoder forward
        # chain of replay n + 1 (api.GraphedStep: defer_reconstructor_update); the last one is flushed INSIDE the timed region
        runner = graphed = R.GraphedStep(step, enc, targets, T, w, defer_reconstructor_update={0: False, 1: True, 2: "recurrent"}[args.defer])
    if args.feed:
        # PCIe-inclusive variant: host batches -> pinned staging -> H2D (side stream) -> the step's input buffers
        import itertools
        from recnet_amd.feed import DeviceFeeder
        host = [(synthetic_features(Bg, F, D, seed=77 + i).numpy(), targets_g.numpy()) for i in range(3)]
        # The feeder's H2D copies land in the device buffers of a ring of slots, two batches ahead of the step, and every slot
        # has its OWN captured graph reading those buffers in place (round 3 copied each batch into one graph's fixed inputs:
        # three device-to-device copies per step on the launch stream, +19 %)
        feeder = DeviceFeeder(itertools.cycle(host), dev, 30, shard=(lo, hi), threaded=args.feed == 2, depth=4, ahead=2)
        slot_graphs = {}

        def runner():
            e, t, T_, wd = next(feeder)
            if not args.graph:
                return step(e, t, T_, wd)
            key = (e.data_ptr(), int(T_))      # (a captured graph is specific to the slot's buffers AND the decode length)
            g_ = slot_graphs.get(key)
            if g_ is None:      # (first use of a slot, inside the warm-up: one capture per slot)
                g_ = slot_graphs[key] = R.GraphedStep(step, e, t, T_, wd, warmup=0,
                                                      defer_reconstructor_update={0: False, 1: True, 2: "recurrent"}[args.defer])
            return g_()
    if args.feed:
        args.warmup = max(args.warmup, 40)          # (reported in the line: see --feed)
    for _ in range(args.warmup):
        runner()
    sync_all()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    for _ in range(args.steps):
        runner()
    if graphed is not None:
        graphed.flush()                                 # every timed step's optimiser work is inside the timed region
    ev1.record()
    host_ms = (time.perf_counter() - t0) / args.steps * 1e3      # what the host needed to ENQUEUE a step (no wait for the device)
    sync_all()
    el = time.perf_counter() - t0
    ms_ev = ev0.elapsed_time(ev1) / args.steps          # the same region by hipEvents on the launch stream
    if world > 1:
        t = torch.tensor([el], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t)
    ms = el / args.steps * 1e3
    sc = step.step_impl.engine.scalar_dict()

    out = None
    if rank == 0:
        # roofline of the dominant kernel, measured live with HIP events around its launches
        eng = step.step_impl.engine

        def prof_pass(site):
            # hipEvent brackets around every launch of `site` in EAGER launches of the step (engine.profile_site); site < 0: around
            # -site empty kernels, the calibration of what the two event records add to a bracket
            def one():
                if site < 0:
                    for _ in range(16):
                        eng.profile_null_launch(-site)
                    return
                eng.train_step_dev(enc, targets, T, w, step.step_impl.seed_base, 3)
            # (never the data-parallel step itself: this runs on rank 0 only, a collective here would wait for ever)
            n_, ms_ = eng.profile_site(site if site > 0 else 5, one, 5)
            return (n_ // 5 if site > 0 else n_), ms_
        phases = None
        if not step.reduce:
            phases = phase_table(eng, runner)
            if phases:
                # The phase table is taken from ISOLATED replays (a synchronisation after each, to read the stamps).  Back to back — the
                # timed loop — a replay's span is 20-35 us LONGER than isolated (the successor's packets are already in the hardware
                # queues) and the idle time between the last kernel of one replay and the first kernel of the next is only ~5-8 us:
                # both read DIRECTLY from the rings of start / end stamps the step's first and last kernels keep (Engine.step_ring,
                # round 5; rounds 3-4 reported ms_per_step - isolated span as "between steps", 35-54 us, and could not explain it).
                for _ in range(8):
                    runner()
                ring = eng.step_ring()
                if len(ring) >= 3:
                    import statistics as _st
                    phases["span_back_to_back_us"] = round(_st.median([b - a for a, b in ring[:-1]]), 1)
                    phases["between_steps_us"] = round(_st.median([ring[i + 1][0] - ring[i][1] for i in range(len(ring) - 1)]), 1)
                phases["ms_per_step_minus_isolated_span_us"] = round(ms * 1e3 - phases["span_us"], 1)
        for _ in range(int(os.environ.get("RN_BENCH_ROOFLINE_REPEATS", "1")) - 1):       # (stress of the measurement path: tools/crash_hunt.sh)
            roofline(eng, prof_pass, kind, args.precision, T=T, cell=args.cell)
        prof = roofline(eng, prof_pass, kind, args.precision, T=T, cell=args.cell)
        if prof and phases:
            # cross-check of the hipEvent brackets (taken around EAGER launches of the step) against the device's own stamps inside
            # the REPLAYED graph (first workgroup started -> last statement of workgroup 0): what the timed region really ran
            prof["chain_kernel_stamps_us"] = {k[len("chain_"):-len("_us")]: v for k, v in phases.items() if k.startswith("chain_") and isinstance(v, float)}
        # whole-step roofline fractions: algorithmic FLOPs against the dense bf16 MFMA peak and
        # algorithmic HBM bytes (optimiser + inputs) against 8 TB/s; per GPU (every rank does the same work)
        Bl = hi - lo
        flops = algorithmic_flops(kind, Bl, F, D, T)
        hbm = algorithmic_hbm_bytes(kind, Bl, F, D)
        mfma_peak = 2.5e15 if args.precision == "bf16" else 157.3e12
        whole = {"algorithmic_gflop_per_step": round(flops / 1e9, 1), "mfma_frac": round(flops / (ms * 1e-3) / mfma_peak, 4),
                 "mfma_peak_tflops": mfma_peak / 1e12, "algorithmic_hbm_mb_per_step": round(hbm / 1e6, 1),
                 "hbm_frac": round(hbm / (ms * 1e-3) / 8e12, 4), "binding_floor_us": round(max(flops / mfma_peak, hbm / 8e12) * 1e6, 1)}
        out = {
            "metric": "captions/sec (train step) MSVD bs=100 28x1536 feats", "value": round(Bg * 1e3 / ms, 1),
            "unit": "captions/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms, 4), "ms_per_step_hipevent": round(ms_ev, 4), "host_enqueue_ms_per_step": round(host_ms, 4), "higher_is_better": True, "scaling": "strong" if args.global_batch else "weak", "vs_baseline": None,
            "dtype": args.precision, "data": "synthetic",
            "config": {"workload": "decoder + %s reconstructor train step (fwd+bwd+clip+Adam), B=%d per GPU, F=%d, "
                                   "D=R=%d, V=4188, E=468, H=512, A=128, T=%d, dropout 0.5, %s cells" % (args.rec, hi - lo, F, D, T, args.cell),
                       "global_batch": Bg, "parallelism": "dp%d" % world, "hipgraph": bool(args.graph), "deferred_reconstructor_update": bool(graphed is not None and graphed.deferred and (graphed.defer_mode is True or eng.lib.recnet_dim(eng.handle, 10) == 1)), "host_feed": bool(args.feed), "grad_allreduce": bool(step.reduce),
                       "loss": round(sc["total_loss"], 5)},
            "roofline": prof, "whole_step": whole, "phases": phases,
        }
        if not args.no_fp32_exact and world == 1 and args.precision == "bf16" and not args.feed:
            out["fp32_exact"] = fp32_exact(R, cfg_over, V, enc, targets, targets_g, B, F)
        if not args.no_cpu_baseline and world == 1:
            v, cores, med = cpu_baseline(kind, B, F, D, V, args.cpu_steps, 1, args.cell)
            out["cpu_baseline"] = {"value": round(v, 2), "unit": "captions/s", "cores": cores, "kind": "port",
                                   "sample": "1 warm-up + %d timed train steps of the same workload (B=%d, T=31) by "
                                             "oracle/recnet_oracle.py on torch-CPU; median %.2f s/step" % (args.cpu_steps, B, med)}
            # BASELINE.json configs[0] (mandatory): decoder only, B=8, the reference's CPU-runnable case
            v1, c1, m1 = cpu_baseline(None, 8, 28, 1536, V, 5, 2, args.cell)
            out["cpu_baseline_c1"] = {"value": round(v1, 2), "unit": "captions/s", "cores": c1, "kind": "port",
                                      "sample": "configs[0]: decoder only, B=8, 28x1536, T=31; 2 warm-up + 5 timed train steps by "
                                                "oracle/recnet_oracle.py on torch-CPU; median %.3f s/step (tools/c1_reference_vs_oracle.py: "
                                                "the oracle times within noise of the imported reference)" % m1}
        print(json.dumps(out), flush=True)
    if world > 1 or under_launcher:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
