#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05; mkdir -p $O
x="--no-cpu-baseline --no-fp32-exact"
for i in 1 2; do
python3 bench.py --rec local --batch 64 --frames 28 --feat 3584 $x | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('C5', d['ms_per_step'], d.get('phases',{}).get('tail_us'))"
python3 bench.py --batch 200 $x | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('C2 B200', d['ms_per_step'])"
done
python3 bench.py --rec local --batch 256 --frames 40 --feat 2048 $x | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('C4 weak B256', d['ms_per_step'])"
python3 bench.py --precision f32 $x | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('C2 f32', d['ms_per_step'])"
python3 bench.py $x | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('C2', d['ms_per_step'])"
python3 bench.py --rec local $x | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('C3', d['ms_per_step'])"
timeout 1500 python3 -m pytest tests/test_gpu_deferred.py tests/test_gpu_configs.py tests/test_gpu_parity.py tests/test_gpu_loop.py tests/test_gpu_acquire_inv.py -q -x > $O/t10_pytest.log 2>&1; echo "pytest rc=$?" >> $O/t10_pytest.log
tail -4 $O/t10_pytest.log
