#!/bin/bash
# A/B across CODE versions on one GPU box: builds the library of a git revision (default HEAD) as csrc/librecnet_hip_base.so;
#   tools/build_base.sh [rev] ; gpurun -- 'ROUNDS=3 bash tools/ab.sh "" "RN_LIB_VARIANT=base" "RN_LIB_VARIANT="'
set -e
rev=${1:-HEAD}
root=$(cd "$(dirname "$0")/.." && pwd)
rm -rf /tmp/rn_base && git -C "$root" worktree prune && git -C "$root" worktree add -f /tmp/rn_base "$rev" > /dev/null 2>&1
make -C "/tmp/rn_base/reconstruction-network-for-video-captioning_amd/csrc" -j4 librecnet_hip.so > /dev/null 2>&1
cp "/tmp/rn_base/reconstruction-network-for-video-captioning_amd/csrc/librecnet_hip.so" "$root/reconstruction-network-for-video-captioning_amd/csrc/librecnet_hip_base.so"
git -C "$root" worktree remove --force /tmp/rn_base
ls -la "$root/reconstruction-network-for-video-captioning_amd/csrc/librecnet_hip_base.so"
