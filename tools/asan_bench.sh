#!/bin/bash
# Host-side AddressSanitizer pass over the whole train step ON the GPU box: the asan build of the C-ABI library (host code
# of api.hip / gemm.hip instrumented, device code not) under the bench driver, a few shapes.  Reports go to gpurun_out/asan/.
mkdir -p gpurun_out/asan
RT=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so | head -1)
export ASAN_OPTIONS="detect_leaks=0:halt_on_error=0:log_path=gpurun_out/asan/log:alloc_dealloc_mismatch=0:new_delete_type_mismatch=0"
export RN_LIB_VARIANT=asan
for args in "" "--batch 200" "--rec local" "--rec none" "--feed 1"; do
  tag=$(echo "$args" | tr -d ' -')
  LD_PRELOAD=$RT timeout 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline $args > gpurun_out/asan/out_$tag.txt 2>&1
  echo "rc=$? [$args]: $(tail -c 300 gpurun_out/asan/out_$tag.txt | tr '\n' ' ' | cut -c1-200)"
done
ls gpurun_out/asan/
for f in gpurun_out/asan/log*; do [ -f "$f" ] && { echo "== $f"; grep -m3 -A12 "ERROR: AddressSanitizer" "$f" | head -60; }; done
