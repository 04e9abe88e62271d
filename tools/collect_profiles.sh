#!/bin/bash
# Runs on the GPU box (gpurun): bench lines, rocprofv3 kernel statistics and PMC traffic passes of the four GPU
# configurations of BASELINE.json, written under gpurun_out/r02/ (copied into profiles/ afterwards).
#   gpurun --timeout 1500 -- 'bash tools/collect_profiles.sh'
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r02; mkdir -p $O
run() {  # name rec batch frames feat extra
  local n=$1 args="--rec $2 --batch $3 --frames $4 --feat $5"
  python3 bench.py $args $6 > $O/bench_$n.json 2> $O/bench_$n.err
  rocprofv3 --kernel-trace --stats -d $O/prof_$n -o $n -- python3 bench.py $args --no-cpu-baseline --steps 30 > $O/bench_under_rocprof_$n.json 2>/dev/null
  python3 tools/rocpd_stats.py $O/prof_$n/${n}_results.db > $O/kernel_stats_$n.csv
  python3 tools/step_timeline.py $O/prof_$n/${n}_results.db 5 > $O/timeline_$n.txt
}
pmc() {  # name rec batch frames feat kind kernel-pattern...   (one pair of counter passes per configuration, parsed per kernel)
  local n=$1 args="--rec $2 --batch $3 --frames $4 --feat $5" kind=$6 B=$3 F=$4 D=$5
  shift 6
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_${n}_$c -- python3 bench.py $args --no-cpu-baseline --steps 10 --warmup 3 > /dev/null 2>&1
  done
  for pat in "$@"; do
    python3 tools/pmc_traffic.py $O/pmc_${n}_FETCH_SIZE $O/pmc_${n}_WRITE_SIZE "$pat" | python3 -c "
import json,sys; d=json.load(sys.stdin); d.update(B=$B, F=$F, D=$D, kind='$kind', T=31, cell='LSTM'); print(json.dumps(d))" > $O/pmc_traffic_${kind}_$pat.json
  done
}
# counter passes first: the bench lines below read the traffic of their dominant kernel from profiles/r02_pmc_traffic_*.json
pmc c2 global 100 28 1536 global dec_chain_kernel dec_chain_bwd_kernel rec_chain_kernel rec_chain_bwd_kernel
pmc c3 local 100 28 1536 local loc_chain_kernel loc_chain_bwd_kernel
for f in $O/pmc_traffic_*.json; do cp $f profiles/r02_$(basename $f); done
run c2 global 100 28 1536 ""
run c3 local 100 28 1536 "--no-cpu-baseline"
run c4 local 32 40 2048 "--no-cpu-baseline"
run c5 local 64 28 3584 "--no-cpu-baseline"
python3 bench.py --rec global --precision f32 --no-cpu-baseline > $O/bench_c2_f32.json 2>/dev/null
python3 bench.py --rec local --precision f32 --no-cpu-baseline > $O/bench_c3_f32.json 2>/dev/null
python3 bench.py --rec none --no-cpu-baseline > $O/bench_decoder_only.json 2>/dev/null
python3 bench.py --rec global --lengths msvd --no-cpu-baseline > $O/bench_c2_msvd_lengths.json 2>/dev/null
python3 bench.py --rec global --cell GRU --no-cpu-baseline > $O/bench_c2_gru.json 2>/dev/null
rm -rf $O/pmc_*_FETCH_SIZE $O/pmc_*_WRITE_SIZE $O/prof_*
ls -la $O
