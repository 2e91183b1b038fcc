#!/usr/bin/env bash
# Where the hot-path forward spends its cycles at the headline shape and at a saturating batch: waits, LDS conflicts,
# instruction mix, occupancy.  rocprofv3 --pmc with --kernel-trace only, one small counter set per pass.
# usage: bash scripts/gpu_pmc_fwd.sh [kernel-substring=fwd_]   -> gpurun_out/r03_fwd_pmc.csv
set -u
export TMPDIR=/tmp
K=${1:-fwd_}
cat > /tmp/fwd_run.py <<'PY'
import sys, torch
sys.path.insert(0, ".")
from neighbour_feature_pooling_amd import NFPPooling
B = int(sys.argv[1])
m = NFPPooling(512, R=1, measure="cosine", padding=1)
n = max(2, (300 << 20) // (B * 512 * 49 * 4))          # rotating sets: more than the Infinity Cache of x
xs = [torch.randn(B, 512, 7, 7, device="cuda") for _ in range(min(n, 48))]
with torch.no_grad():
    for i in range(24):
        m(xs[i % len(xs)])
torch.cuda.synchronize()
PY
for B in 64 4096; do
for c in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS" \
         "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_WAVES SQ_BUSY_CYCLES SQ_LEVEL_WAVES GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
  tag=$(echo $c | tr ' ' '_'); out=gpurun_out/pmc_fwd_${B}_$tag; rm -rf $out; mkdir -p $out
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out -o pmc -- python3 /tmp/fwd_run.py $B > $out/log.txt 2>&1
  echo "B=$B $c rc=$?"
done
done
python3 - "$K" <<'PY'
import csv, glob, collections, sys, re
K = sys.argv[1]
lines = []
for B in (64, 4096):
    agg = collections.defaultdict(list)
    dur = []
    for f in glob.glob(f"gpurun_out/pmc_fwd_{B}_*/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if K in row["Kernel_Name"]:
                agg[row["Counter_Name"]].append(float(row["Counter_Value"]))
                name = row["Kernel_Name"]
    for f in glob.glob(f"gpurun_out/pmc_fwd_{B}_SQ_WAVE*/**/*kernel_trace.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if K in row["Kernel_Name"]:
                dur.append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3)
    lines.append(f"# forward on [{B},512,7,7] f32 cosine k3, rotating inputs, per launch (mean of the launches; SQ counters summed over the chip); kernel {name[:90]}")
    if dur:
        lines.append(f"duration_us_under_pmc,{sum(dur)/len(dur):.2f}")
    for k, v in sorted(agg.items()):
        lines.append(f"{k},{sum(v)/len(v):.0f}")
open("gpurun_out/r03_fwd_pmc.csv", "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
