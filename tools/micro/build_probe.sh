#!/bin/bash
# builds the micro-benchmarks quoted in DESIGN.md section 5 (run them on an MI355X; each prints its own numbers)
cd "$(dirname "$0")"
F="-O3 -std=c++17 --offload-arch=gfx950 -Wno-unused-result -Wno-unused-value"
# per-step ring GEMM with pieces compiled out
for v in full SKIP_STORE; do
  d=""; [ "$v" != full ] && d="-DGC_PROBE_$v"
  hipcc $F $d chain_probe.hip -o probe_$v &
done
# persistent reconstructor forward chain: both tilings, pieces compiled out (MASTER=0/1 picks the barrier)
hipcc $F persist_probe.hip -o pprobe_full &
hipcc $F -DRC_PROBE_MS2 persist_probe.hip -o pprobe_MS2 &
hipcc $F -DRC_PROBE_MS2 -DRC_PROBE_SKIP_A persist_probe.hip -o pprobe_MS2_SKIPA &
hipcc $F -DRC_PROBE_NO_BARRIER -DRC_PROBE_SKIP_A persist_probe.hip -o pprobe_NOBAR_SKIPA &
wait
# persistent decoder forward chain: per-phase timeline of one workgroup (LL=0/1, MASTER=0/1; SHORT=1 withholds a workgroup)
hipcc $F dec_probe.hip -o dprobe_0 &
hipcc $F -DDC_PROBE_WG=77 dec_probe.hip -o dprobe_77 &
# store -> load hand-over latency between two CUs, same / different XCD
hipcc $F xcd_pingpong.hip -o xcd_pingpong &
hipcc $F stream_floor.hip -o stream_floor &
wait
ls -la probe_* pprobe_* dprobe_* xcd_pingpong stream_floor
