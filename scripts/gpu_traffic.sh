#!/usr/bin/env bash
# HBM traffic of the hot kernels: FETCH_SIZE and WRITE_SIZE in SEPARATE rocprofv3 --pmc passes
# (TCC has 4 slots; FETCH_SIZE takes 3, WRITE_SIZE 2), --kernel-trace only.  Writes
# gpurun_out/traffic.json; copy it to profiles/traffic_latest.json to have bench.py report it.
#   usage: bash scripts/gpu_traffic.sh [bench args...]
set -u
export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  out=gpurun_out/pmc_$c; rm -rf $out; mkdir -p $out
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out -o pmc -- \
    python3 bench.py --no-cpu-baseline --no-extras --steps 20 --warmup 5 "$@" > $out/bench.log 2>&1
  echo "$c rc=$?"
done
python3 - <<'PY'
import csv, glob, json, collections
res = collections.defaultdict(dict)
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f"gpurun_out/pmc_{c}/**/*counter_collection.csv", recursive=True)
    agg = collections.defaultdict(list)
    for row in csv.DictReader(open(f[0])):
        if row["Counter_Name"] != c: continue
        name = row["Kernel_Name"]
        if "nfp::" not in name: continue
        key = name.split("nfp::")[1].split("(")[0]
        agg[key].append(float(row["Counter_Value"]))
    for k, v in agg.items():
        res[k][c + "_KB"] = sum(v) / len(v)
        res[k]["launches"] = len(v)
out = {"note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), KB per launch averaged over launches; "
               "hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024: FETCH_SIZE is doubled per MI355X_MICROARCH.md "
               "(gfx950 counts 128-B requests of wide streaming reads at 64 B)", "kernels": {}}
for k, v in res.items():
    f, w = v.get("FETCH_SIZE_KB", 0.0), v.get("WRITE_SIZE_KB", 0.0)
    out["kernels"][k] = {"FETCH_SIZE_KB": round(f, 1), "WRITE_SIZE_KB": round(w, 1), "launches": v.get("launches"),
                         "hbm_bytes_per_launch": int((2 * f + w) * 1024)}
json.dump(out, open("gpurun_out/traffic.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
# merge into gpurun_out/traffic_workloads.json under this workload's key (bench.py::workload_key)
python3 - "$@" <<'PY'
import json, os, sys
sys.path.insert(0, os.getcwd())
import bench
sys.argv = ["bench.py"] + sys.argv[1:]
key = bench.workload_key(bench.parse())
one = json.load(open("gpurun_out/traffic.json"))
path = "gpurun_out/traffic_workloads.json" if os.path.exists("gpurun_out/traffic_workloads.json") else "profiles/traffic_latest.json"
allw = json.load(open(path)) if os.path.exists(path) else {"note": one["note"]}
if allw.get("source_hash") != bench.source_hash():   # profiles of other kernel sources are stale: start over
    allw = {"note": one["note"], "source_hash": bench.source_hash()}
allw.setdefault("workloads", {})[key] = one["kernels"]
json.dump(allw, open("gpurun_out/traffic_workloads.json", "w"), indent=1)
print("merged", key, "->", "gpurun_out/traffic_workloads.json (copy to profiles/traffic_latest.json)")
PY
