#!/bin/bash
# Runs on the GPU box (gpurun): bench lines, rocprofv3 kernel statistics, step timelines and PMC traffic passes of the GPU
# configurations of BASELINE.json, written under gpurun_out/$RND/ (copied into profiles/ afterwards).
#   gpurun --timeout 2400 -- 'bash tools/collect_profiles.sh'
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
RND=${RND:-r06}
O=gpurun_out/$RND; mkdir -p $O
run() {  # name "bench args" extra
  local n=$1 args="$2"
  python3 bench.py $args $3 > $O/bench_$n.json 2> $O/bench_$n.err
  rocprofv3 --kernel-trace --stats -d $O/prof_$n -o $n -- python3 bench.py $args --no-cpu-baseline --no-fp32-exact --steps 30 > $O/bench_under_rocprof_$n.json 2>/dev/null
  python3 tools/rocpd_stats.py $O/prof_$n/${n}_results.db > $O/kernel_stats_$n.csv
  python3 tools/step_timeline.py $O/prof_$n/${n}_results.db 5 > $O/timeline_$n.txt
}
pmc() {  # name "bench args" kind B F D  kernel-pattern...   (one pair of counter passes per configuration, parsed per kernel)
  local n=$1 args="$2" kind=$3 B=$4 F=$5 D=$6
  shift 6
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_${n}_$c -- python3 bench.py $args --no-cpu-baseline --no-fp32-exact --steps 10 --warmup 3 > /dev/null 2>&1
  done
  for pat in "$@"; do
    local fn=$(echo "$pat" | tr -c 'A-Za-z0-9_' '_' | sed 's/__*/_/g; s/_$//')
    python3 tools/pmc_traffic.py $O/pmc_${n}_FETCH_SIZE $O/pmc_${n}_WRITE_SIZE "$pat" | python3 -c "
import json,sys; d=json.load(sys.stdin); d.update(B=$B, F=$F, D=$D, kind='$kind', T=31, cell='LSTM', config='$n'); print(json.dumps(d))" > $O/pmc_traffic_${n}_$fn.json
  done
}
# PART=1: counter passes + traced runs of C2..C5 and the fp32 path; PART=2: everything else; unset: both (about 55 minutes)
if [ "${PART:-0}" != "2" ]; then
C2=""; C3="--rec local"; C4="--rec local --batch 32 --frames 40 --feat 2048"; C5="--rec local --batch 64 --frames 28 --feat 3584"
# counter passes first: the bench lines below read the traffic of their dominant kernel from profiles/${RND}_pmc_traffic_*.json
pmc c2 "$C2" global 100 28 1536 dec_chain_kernel dec_chain_bwd_kernel rec_chain_kernel rec_chain_bwd_kernel
pmc c3 "$C3" local 100 28 1536 loc_chain_kernel loc_chain_bwd_kernel
pmc c4 "$C4" local 32 40 2048 loc_chain_kernel loc_chain_bwd_kernel dec_chain_kernel dec_chain_bwd_kernel
pmc c5 "$C5" local 64 28 3584 loc_chain_kernel lcbig_bwd_kernel dec_chain_bwd_kernel
for f in $O/pmc_traffic_*.json; do cp $f profiles/${RND}_$(basename $f); done
run c2 "$C2" ""
run c3 "$C3" "--no-cpu-baseline"
run c4 "$C4" "--no-cpu-baseline --no-fp32-exact"
run c5 "$C5" "--no-cpu-baseline --no-fp32-exact"
# the exact-fp32 path under the tracer: how much of its step is launch / dependency latency
run c2_f32 "--precision f32" "--no-cpu-baseline --no-fp32-exact"
if [ "${PART:-0}" = "1" ]; then rm -rf $O/pmc_*_FETCH_SIZE $O/pmc_*_WRITE_SIZE $O/prof_*; ls $O | wc -l; exit 0; fi
fi      # PART != 2
x="--no-cpu-baseline --no-fp32-exact"
C2=""; C3="--rec local"; C4="--rec local --batch 32 --frames 40 --feat 2048"; C5="--rec local --batch 64 --frames 28 --feat 3584"
python3 bench.py --rec local --precision f32 $x > $O/bench_c3_f32.json 2>/dev/null
python3 bench.py --rec none $x > $O/bench_decoder_only.json 2>/dev/null
python3 bench.py --lengths msvd $x > $O/bench_c2_msvd_lengths.json 2>/dev/null
python3 bench.py --cell GRU $x > $O/bench_c2_gru.json 2>/dev/null
# batches above the 112-row panels (row groups): C2 at B = 200, C4's weak-scaling variant (256 captions per GPU, 40 x 2048), and
# the same on the per-step kernels (RN_ROW_GROUPS=0)
python3 bench.py --batch 200 $x > $O/bench_c2_B200.json 2>/dev/null
RN_ROW_GROUPS=0 python3 bench.py --batch 200 $x > $O/bench_c2_B200_per_step_kernels.json 2>/dev/null
python3 bench.py --rec local --batch 256 --frames 40 --feat 2048 $x > $O/bench_c4_weak_B256.json 2>/dev/null
RN_ROW_GROUPS=0 python3 bench.py --rec local --batch 256 --frames 40 --feat 2048 $x > $O/bench_c4_weak_B256_per_step_kernels.json 2>/dev/null
RN_ALT=loc_no_hybrid python3 bench.py $C5 $x > $O/bench_c5_per_step_forward.json 2>/dev/null
# round 6: 28 x 3584 at 128 captions per GPU — the local chains in two row groups, and the per-step kernels
python3 bench.py --rec local --batch 128 --frames 28 --feat 3584 $x > $O/bench_c5_B128.json 2>/dev/null
RN_ALT=loc_no_hybrid RN_PER_STEP=loc_big python3 bench.py --rec local --batch 128 --frames 28 --feat 3584 $x > $O/bench_c5_B128_per_step_kernels.json 2>/dev/null
# PCIe-inclusive rate (never `value`): every step takes a fresh host batch through feed.DeviceFeeder
python3 bench.py --feed 1 $x > $O/bench_c2_host_feed.json 2>/dev/null
python3 bench.py --defer 1 $x > $O/bench_c2_deferred_reconstructor_update.json 2>/dev/null
# the step with everything inside it (round 3's form) and the switches of round 4, one at a time
python3 bench.py --defer 0 $x > $O/bench_c2_update_inside_the_step.json 2>/dev/null
python3 bench.py --rec local --defer 0 $x > $O/bench_c3_update_inside_the_step.json 2>/dev/null
RN_GEMM_GROUP=0 python3 bench.py $x > $O/bench_c2_no_grouped_launches.json 2>/dev/null
RN_ALT=dec_wh_in_phase_a python3 bench.py $x > $O/bench_c2_attention_projection_in_phase_A.json 2>/dev/null
RN_ADAM_EPILOGUE=0 python3 bench.py $x > $O/bench_c2_adam_kernel_instead_of_epilogue.json 2>/dev/null
RN_WAIT_CHAIN=0 python3 bench.py $x > $O/bench_c2_no_residency_waits.json 2>/dev/null
RN_ALT=dec_relayed_barrier python3 bench.py $x > $O/bench_c2_relayed_barrier_in_decoder_chains.json 2>/dev/null
RN_ALT=dec_all_rows python3 bench.py $x > $O/bench_c2_forward_phase_A_all_rows.json 2>/dev/null
# the data-parallel step at ONE rank (no byte crosses xGMI): one captured graph with the collectives inside / three graphs
python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --force-allreduce $x > $O/bench_c2_dp_one_rank_one_graph.json 2>/dev/null
RN_DP_ONE_GRAPH=0 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29518 bench.py --gpus 1 --force-allreduce $x > $O/bench_c2_dp_one_rank_three_graphs.json 2>/dev/null
for i in 2 3; do python3 bench.py --feed 1 $x > $O/bench_c2_host_feed_$i.json 2>/dev/null; python3 bench.py $x > $O/bench_c2_resident_$i.json 2>/dev/null; done
python3 tools/rccl_bucket_bench.py > $O/rccl_buckets_1rank_c2.json 2>/dev/null
# in-kernel stamps of the chain kernels (probe build of the library)
RN_LIB_VARIANT=probe python3 tools/loc_chain_probe.py 100 28 1536 > $O/chain_probe_c3.txt 2>/dev/null
RN_LIB_VARIANT=probe python3 tools/loc_chain_probe.py 64 28 3584 > $O/chain_probe_c5.txt 2>/dev/null
RN_ALT=loc_no_hybrid RN_PER_STEP=loc_big python3 bench.py $C5 $x > $O/bench_c5_per_step_kernels.json 2>/dev/null
for cfg in c2 c5; do
  a="$C2"; [ $cfg = c5 ] && a="$C5"
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_mfma_$cfg -- python3 bench.py $a $x --steps 10 --warmup 3 > /dev/null 2>&1
  python3 tools/pmc_mfma.py $O/pmc_mfma_$cfg > $O/pmc_mfma_$cfg.json 2>/dev/null
  rm -rf $O/pmc_mfma_$cfg
done
rm -rf $O/pmc_*_FETCH_SIZE $O/pmc_*_WRITE_SIZE $O/prof_*
ls -la $O
# round 5 additions (hipBLASLt only as the yardstick, through torch.matmul): GEMM K-loop / epilogue stamps, cold / warm shapes against the vendor library, the idle time between replays
./tools/micro/gemm_probe > $O/gemm_probe.txt 2>&1
python3 tools/gemm_cold_probe.py > $O/gemm_cold_vs_hipblaslt.txt 2>&1
python3 tools/between_steps_probe.py global > $O/between_steps_c2.txt 2>&1
python3 tools/soak.py 3000 > $O/soak.json 2> $O/soak.err
ls $O | wc -l
