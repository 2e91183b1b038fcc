#!/usr/bin/env bash
# rocprofv3 kernel trace + stats of the default bench (run on the GPU box via gpurun).
# usage: bash scripts/gpu_profile.sh <tag> [bench args...]
set -u
tag=${1:-run}; shift || true
mkdir -p gpurun_out/prof_$tag
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$tag -o trace -- \
  python3 bench.py --no-cpu-baseline --no-extras "$@" > gpurun_out/prof_$tag/bench.log 2>&1
rc=$?
echo "rocprof rc=$rc"; tail -n 3 gpurun_out/prof_$tag/bench.log
find gpurun_out/prof_$tag -name '*stats*' | head
f=$(find gpurun_out/prof_$tag -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] && head -20 "$f"
exit $rc
