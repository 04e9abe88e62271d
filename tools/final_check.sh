#!/bin/bash
# The round's closing run on the GPU box: the whole GPU suite, then the profile collection.   gpurun --timeout 3300 -- 'bash tools/final_check.sh'
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/${RND:-r06}
timeout 2700 python3 -m pytest tests -x -q -m gpu > gpurun_out/${RND:-r06}/full_gpu_suite.log 2>&1; echo "pytest rc=$?" >> gpurun_out/${RND:-r06}/full_gpu_suite.log
tail -3 gpurun_out/${RND:-r06}/full_gpu_suite.log
# (the hipGraphLaunch segfault of round 5 showed in this file, in full-suite order: loop it behind the suite)
for i in 1 2 3 4 5; do timeout 600 python3 -m pytest tests/test_gpu_deferred.py tests/test_gpu_loop.py -x -q -m gpu 2>&1 | tail -1 >> gpurun_out/${RND:-r06}/deferred_loop.log; done
cat gpurun_out/${RND:-r06}/deferred_loop.log
bash tools/collect_profiles.sh > gpurun_out/${RND:-r06}/collect.log 2>&1
tail -2 gpurun_out/${RND:-r06}/collect.log
