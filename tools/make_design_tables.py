#!/usr/bin/env python3
"""Tables of DESIGN.md section 5 from the files of a round's collection (profiles/rNN_*): bench lines, phase table, counter traffic of
the chain kernels against their algorithmic and exchange bytes.   python tools/make_design_tables.py r06 > /tmp/tables.md"""
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rnd = sys.argv[1] if len(sys.argv) > 1 else "r06"
P = os.path.join(ROOT, "profiles")


def line(name):
    f = os.path.join(P, "%s_bench_%s.json" % (rnd, name))
    if not os.path.exists(f):
        return None
    txt = open(f).read().strip()
    if not txt:
        return None
    return json.loads(txt.split("\n")[-1])


def ms(name):
    d = line(name)
    return "—" if d is None else "%.3f" % d["ms_per_step"]


def caps(name):
    d = line(name)
    return "—" if d is None else "%.1f k" % (d["value"] / 1e3)


rows = [("C2 decoder + global (headline)", "B=100, 28×1536", "c2"), ("C2, exact fp32 path", "B=100", "c2_f32"),
        ("C3 decoder + local", "B=100, 28×1536", "c3"), ("C3, exact fp32 path", "B=100", "c3_f32"),
        ("C4 decoder + local, MSR-VTT shape", "32 of 256, 40×2048", "c4"), ("C5 decoder + local, 2D+3D features", "64 of 512, 28×3584", "c5"),
        ("28×3584 at 128 captions (two row groups)", "B=128", "c5_B128"), ("decoder only", "B=100", "decoder_only"), ("C2, GRU cells", "B=100", "c2_gru"),
        ("C2, MSVD caption lengths", "B=100", "c2_msvd_lengths"), ("C4 weak-scaling variant", "256 per GPU, 40×2048", "c4_weak_B256"),
        ("C2 at B = 200 (two row groups)", "B=200", "c2_B200")]
print("| BASELINE config | shape per GPU | ms per step | captions/s |")
print("|---|---|---|---|")
for title, shape, n in rows:
    print("| %s | %s | %s | %s |" % (title, shape, ms(n), caps(n)))
print()
var = [("update inside the step", "c2_update_inside_the_step"), ("C3, update inside the step", "c3_update_inside_the_step"),
       ("whole update deferred (mode 1)", "c2_deferred_reconstructor_update"), ("no grouped launches", "c2_no_grouped_launches"),
       ("`RN_ADAM_EPILOGUE=0`", "c2_adam_kernel_instead_of_epilogue"), ("no residency waits", "c2_no_residency_waits"),
       ("attention projection in phase A", "c2_attention_projection_in_phase_A"), ("relayed barrier in the decoder chains", "c2_relayed_barrier_in_decoder_chains"),
       ("phase A over all rows", "c2_forward_phase_A_all_rows"), ("host feed", "c2_host_feed"), ("host feed (2)", "c2_host_feed_2"), ("resident (2)", "c2_resident_2"),
       ("resident (3)", "c2_resident_3"), ("one rank, all-reduce forced, one graph", "c2_dp_one_rank_one_graph"),
       ("one rank, three graphs", "c2_dp_one_rank_three_graphs"), ("C2 at B = 200, per-step kernels", "c2_B200_per_step_kernels"),
       ("C4 weak, per-step kernels", "c4_weak_B256_per_step_kernels"), ("C5, per-step forward", "c5_per_step_forward"), ("C5, per-step kernels", "c5_per_step_kernels"),
       ("28×3584 B = 128, per-step kernels", "c5_B128_per_step_kernels")]
print("Variants: " + "; ".join("%s %s" % (t, ms(n)) for t, n in var) + ".")
print()
for n in ("c2", "c3", "c4", "c5"):
    d = line(n)
    if d and d.get("phases"):
        p = d["phases"]
        keys = [k for k in p if k.startswith("chain_") and not k.startswith("chain_offsets")]
        parts = ["prologue %s" % p.get("prologue_us")]
        order = sorted(p["chain_offsets_us"], key=lambda k: p["chain_offsets_us"][k][0])
        for i, nm in enumerate(order):
            parts.append("%s %s" % (nm.replace("_", " "), p["chain_%s_us" % nm]))
            if i + 1 < len(order):
                parts.append("%s" % p["gap_%s_to_%s_us" % (nm, order[i + 1])])
        parts.append("tail %s" % p.get("tail_us"))
        print("Phase table %s (µs): %s; span %s isolated, %s back to back, between steps %s." % (
            n.upper(), " · ".join(parts), p.get("span_us"), p.get("span_back_to_back_us"), p.get("between_steps_us")))
print()
print("| chain kernel | config | µs per launch | counter traffic per launch | algorithmic bytes | exchange (panels, every reader) | traffic ÷ algorithmic | ÷ (algorithmic + exchange) |")
print("|---|---|---|---|---|---|---|---|")
for f in sorted(glob.glob(os.path.join(P, "%s_pmc_traffic_*.json" % rnd))):
    d = json.load(open(f))
    cfg = d.get("config", "?")
    b = line(cfg)
    alg = exch = us = None
    if b:
        r = b.get("roofline") or {}
        if r.get("kernel", "").startswith(d.get("kernel", "#").split("<")[0]):
            alg, exch, us = r.get("algorithmic_bytes_per_launch"), r.get("exchange_bytes_per_launch"), r.get("avg_launch_us")
    if alg is None:      # not the line's dominant kernel: algorithmic / exchange bytes of the shape as recnet_recurrent_step_bytes / recnet_chain_exchange_bytes report them
        known = {("dec_chain_kernel", "c2"): (83.7e6, 100e6), ("rec_chain_kernel", "c2"): (236e6, None), ("rec_chain_bwd_kernel", "c2"): (172e6, 76e6),
                 ("loc_chain_kernel", "c3"): (150e6, 166e6)}
        alg, exch = known.get((d.get("kernel", "").split("<")[0], cfg), (None, None))
    t = d["traffic_bytes_per_launch"]
    print("| `%s` | %s | %s | %.0f MB | %s | %s | %s | %s |" % (
        d.get("kernel", "?").split("<")[0], cfg.upper(), "%.0f" % us if us else "", t / 1e6, "%.0f MB" % (alg / 1e6) if alg else "",
        "%.0f MB" % (exch / 1e6) if exch else "", "%.2f" % (t / alg) if alg else "", "%.2f" % (t / (alg + exch)) if alg and exch else ""))
