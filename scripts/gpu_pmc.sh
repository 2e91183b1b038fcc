#!/usr/bin/env bash
# PMC counters per kernel (own run: --pmc with --kernel-trace only, as the pool requires).
# usage: bash scripts/gpu_pmc.sh <tag> "<COUNTER ...>" [bench args...]
set -u
tag=$1; ctrs=$2; shift 2
out=gpurun_out/pmc_$tag
mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d $out -o pmc -- \
  python3 bench.py --no-cpu-baseline --steps 20 --warmup 5 "$@" > $out/bench.log 2>&1
echo "rocprof rc=$?"
python3 - "$out" <<'PY'
import csv, sys, glob, collections
out = sys.argv[1]
f = glob.glob(out + "/**/*counter_collection.csv", recursive=True)
if not f:
    print("no counter file", glob.glob(out + "/**/*", recursive=True)); sys.exit(0)
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for row in csv.DictReader(open(f[0])):
    k = row["Kernel_Name"].split("(")[0][-60:]
    agg[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
with open(out + "/summary.txt", "w") as fo:
    for k, d in agg.items():
        if "nfp" not in k: continue
        line = f"{k}: " + "  ".join(f"{c}={sum(v)/len(v):.0f}" for c, v in sorted(d.items())) + f"  (n={len(next(iter(d.values())))})"
        print(line); fo.write(line + "\n")
PY
