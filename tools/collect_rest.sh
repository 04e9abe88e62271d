#!/bin/bash
# second part of the round's collection (bench lines only, no tracer): gpurun --timeout 1500 -- 'bash tools/collect_rest.sh'
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
RND=r04; O=gpurun_out/$RND; mkdir -p $O
x="--no-cpu-baseline --no-fp32-exact"
C5="--rec local --batch 64 --frames 28 --feat 3584"
python3 bench.py $C5 $x > $O/bench_c5.json 2>/dev/null
python3 bench.py --rec none $x > $O/bench_decoder_only.json 2>/dev/null
python3 bench.py --lengths msvd $x > $O/bench_c2_msvd_lengths.json 2>/dev/null
python3 bench.py --cell GRU $x > $O/bench_c2_gru.json 2>/dev/null
python3 bench.py --batch 200 $x > $O/bench_c2_B200.json 2>/dev/null
python3 bench.py --rec local --batch 256 --frames 40 --feat 2048 $x > $O/bench_c4_weak_B256.json 2>/dev/null
python3 bench.py --defer 0 $x > $O/bench_c2_update_inside_the_step.json 2>/dev/null
python3 bench.py --rec local --defer 0 $x > $O/bench_c3_update_inside_the_step.json 2>/dev/null
RN_GEMM_GROUP=0 python3 bench.py $x > $O/bench_c2_no_grouped_launches.json 2>/dev/null
RN_ALT=dec_wh_in_phase_a python3 bench.py $x > $O/bench_c2_attention_projection_in_phase_A.json 2>/dev/null
RN_ADAM_EPILOGUE=0 python3 bench.py $x > $O/bench_c2_adam_kernel_instead_of_epilogue.json 2>/dev/null
RN_WAIT_CHAIN=0 python3 bench.py $x > $O/bench_c2_no_residency_waits.json 2>/dev/null
python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --force-allreduce $x 2>/dev/null | tail -1 > $O/bench_c2_dp_one_rank_one_graph.json
RN_DP_ONE_GRAPH=0 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29518 bench.py --gpus 1 --force-allreduce $x 2>/dev/null | tail -1 > $O/bench_c2_dp_one_rank_three_graphs.json
for i in 1 2 3; do python3 bench.py --feed 1 $x 2>/dev/null | tail -1 > $O/bench_c2_host_feed_$i.json; python3 bench.py $x > $O/bench_c2_resident_$i.json 2>/dev/null; done
python3 tools/soak.py 20000 > $O/soak.json 2> $O/soak.err
ls $O | wc -l
