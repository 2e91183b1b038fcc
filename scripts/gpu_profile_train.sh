#!/usr/bin/env bash
# rocprofv3 kernel stats of one end-to-end train configuration (run on the GPU box via gpurun).
# usage: bash scripts/gpu_profile_train.sh <tag> [train args...]
set -u
tag=${1:-train}; shift || true
mkdir -p gpurun_out/prof_$tag
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$tag -o trace -- \
  python3 -m neighbour_feature_pooling_amd.train "$@" > gpurun_out/prof_$tag/train.log 2>&1
rc=$?
echo "rocprof rc=$rc"; tail -n 2 gpurun_out/prof_$tag/train.log
f=$(find gpurun_out/prof_$tag -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] && head -25 "$f" | cut -c1-220
exit $rc
