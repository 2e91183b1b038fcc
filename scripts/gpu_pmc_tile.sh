#!/usr/bin/env bash
# Instruction mix, waits and occupancy of the row-band kernels (nfp_tile.h) on one large map, per launch.
# rocprofv3 --pmc with --kernel-trace only, one small counter set per pass.
# usage: bash scripts/gpu_pmc_tile.sh C S [B=256] [layout=nchw|nhwc|nhwc-bf16]   -> gpurun_out/tile_pmc_<C>x<S>_<layout>.csv
set -u
export TMPDIR=/tmp
C=${1:-16}; S=${2:-112}; B=${3:-256}; LAY=${4:-nchw}
for c in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" \
         "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT" "SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU" "FETCH_SIZE" "WRITE_SIZE"; do
  tag=$(echo $c | tr ' ' '_'); out=gpurun_out/pmc_tile_${C}x${S}_${LAY}_$tag; rm -rf $out; mkdir -p $out
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out -o pmc -- python3 scripts/run_bigmaps_for_rocprof.py $B $C $S $LAY > $out/log.txt 2>&1
  echo "$c rc=$?"
done
python3 - "$C" "$S" "$B" "$LAY" <<'PY'
import csv, glob, collections, sys
C, S, B, LAY = sys.argv[1:5]
lines = []
for K in ("fwd_tile", "bwd_tile"):
    agg = collections.defaultdict(list); dur = []; name = ""
    for f in glob.glob(f"gpurun_out/pmc_tile_{C}x{S}_{LAY}_*/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if K in row["Kernel_Name"]:
                agg[row["Counter_Name"]].append(float(row["Counter_Value"])); name = row["Kernel_Name"]
    for f in glob.glob(f"gpurun_out/pmc_tile_{C}x{S}_{LAY}_SQ_WAVE_CYCLES*/**/*kernel_trace.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if K in row["Kernel_Name"]:
                dur.append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3)
    lines.append(f"# [{B},{C},{S},{S}] {LAY} cosine k3, per launch (mean over launches; SQ counters summed over the chip); {name[:80]}")
    if dur:
        lines.append(f"duration_us_under_pmc,{sum(dur)/len(dur):.2f}")
    for k, v in sorted(agg.items()):
        lines.append(f"{k},{sum(v)/len(v):.0f}")
open(f"gpurun_out/tile_pmc_{C}x{S}_{LAY}.csv", "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
