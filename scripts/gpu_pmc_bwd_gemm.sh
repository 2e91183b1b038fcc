#!/usr/bin/env bash
# Where the config-5 backward (bwd_fast<R2,l2,bf16,nhwc,mfma>) spends its cycles: MFMA busy, LDS bank conflicts, waits.
# rocprofv3 --pmc with --kernel-trace only, one small counter set per pass.   usage: bash scripts/gpu_pmc_bwd_gemm.sh
set -u
export TMPDIR=/tmp
cat > /tmp/bwd_run.py <<'PY'
import sys, torch
sys.path.insert(0, ".")
from neighbour_feature_pooling_amd import NFPPooling
m = NFPPooling(192, R=2, measure="norm", p=2, padding=2)
x = torch.randn(256, 192, 14, 14, device="cuda").bfloat16().contiguous(memory_format=torch.channels_last).requires_grad_(True)
out = m(x)
go = torch.randn_like(out)
for _ in range(20):
    torch.autograd.grad(out, x, go, retain_graph=True)
torch.cuda.synchronize()
PY
for c in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_WAIT_INST_LDS SQ_WAVE_CYCLES" \
         "SQ_INSTS_LDS SQ_INSTS_VALU" "SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD" "SQ_INSTS_SALU SQ_INSTS_MFMA"; do
  tag=$(echo $c | tr ' ' '_'); out=gpurun_out/pmc_bwdg_$tag; rm -rf $out; mkdir -p $out
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out -o pmc -- python3 /tmp/bwd_run.py > $out/log.txt 2>&1
  echo "$c rc=$?"
done
python3 - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(list)
for f in glob.glob("gpurun_out/pmc_bwdg_*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "bwd_fast" in row["Kernel_Name"]:
            agg[row["Counter_Name"]].append(float(row["Counter_Value"]))
lines = ["# bwd_fast<R2,l2,bf16,nhwc,mfma*> (whichever form the dispatcher picks) on [256,192,14,14] bf16 channels-last, per launch (mean of 20 launches; SQ counters summed over the chip)"]
for k, v in sorted(agg.items()):
    lines.append(f"{k},{sum(v)/len(v):.0f}")
open("gpurun_out/config5_backward_pmc.csv", "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
