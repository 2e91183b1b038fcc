#!/usr/bin/env bash
# Run on the GPU box (via gpurun): smoke -> gpu tests -> bench, each under its own timeout.
# An ordinary failure lets later steps run (so one call yields as much evidence as possible);
# a step that is KILLED (timeout/signal) stops everything after it.
set -u
mkdir -p gpurun_out
status=0
step() {  # step <name> <timeout_s> <cmd...>
  local name=$1 t=$2; shift 2
  echo "=== $name ==="
  timeout -k 10 "$t" "$@" > "gpurun_out/$name.log" 2>&1
  local rc=$?
  echo "$name rc=$rc"; tail -n 15 "gpurun_out/$name.log"
  if [ $rc -eq 124 ] || [ $rc -ge 128 ]; then echo "$name was killed; stopping"; exit $rc; fi
  [ $rc -ne 0 ] && status=$rc
  return 0
}
step smoke 300 python -c "import __graft_entry__ as g; g.smoke()"
step gpu_tests 600 python -m pytest tests -m gpu -x -q
step bench 420 python bench.py "$@"
exit $status
