#!/bin/bash
# builds tools/micro/probe_<variant> for each compiled-out piece
cd "$(dirname "$0")"
for v in full SKIP_STORE; do
  d=""; [ "$v" != full ] && d="-DGC_PROBE_$v"
  hipcc -O3 --offload-arch=gfx950 -Wno-unused-result $d chain_probe.hip -o probe_$v &
done
wait
ls -la probe_*
