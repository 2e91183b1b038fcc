#!/usr/bin/env bash
# Matrix-core evidence for fwd_gram: MFMA busy cycles and instruction counts of the config-5 forward
# (rocprofv3 --pmc with --kernel-trace only).   usage: bash scripts/gpu_pmc_mfma.sh
set -u
export TMPDIR=/tmp
cat > /tmp/mfma_run.py <<'PY'
import sys, torch
sys.path.insert(0, ".")
from neighbour_feature_pooling_amd import NFPPooling
m = NFPPooling(192, R=2, measure="norm", p=2, padding=2)
x = torch.randn(256, 192, 14, 14, device="cuda").bfloat16().contiguous(memory_format=torch.channels_last)
with torch.no_grad():
    for _ in range(20):
        m(x)
torch.cuda.synchronize()
PY
for c in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU" "SQ_INSTS_MFMA SQ_WAVE_CYCLES"; do
  tag=$(echo $c | tr ' ' '_'); out=gpurun_out/pmc_mfma_$tag; rm -rf $out; mkdir -p $out
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out -o pmc -- python3 /tmp/mfma_run.py > $out/log.txt 2>&1
  echo "$c rc=$?"
done
python3 - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(list)
for f in glob.glob("gpurun_out/pmc_mfma_*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "fwd_gram" in row["Kernel_Name"]:
            agg[row["Counter_Name"]].append(float(row["Counter_Value"]))
lines = ["# fwd_gram<R2,l2,bf16,nhwc> on [256,192,14,14] bf16 channels-last, per launch (mean of 20)"]
for k, v in sorted(agg.items()):
    lines.append(f"{k},{sum(v)/len(v):.0f}")
open("gpurun_out/pmc_mfma_summary.csv", "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
