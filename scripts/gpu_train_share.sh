#!/usr/bin/env bash
# Kernel shares of the 1-GPU train steps (rocprofv3 --kernel-trace --stats; run on the GPU box via gpurun):
# where the NFP kernels stand inside a whole backbone + NFP step.  Writes gpurun_out/r02_k_train_step_kernel_share.csv
set -u
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
run() {  # tag, train args...
  tag=$1; shift
  rm -rf gpurun_out/prof_$tag; mkdir -p gpurun_out/prof_$tag
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$tag -o trace -- \
    python3 -m neighbour_feature_pooling_amd.train "$@" --steps 10 --warmup 3 > gpurun_out/prof_$tag/train.log 2>&1
  echo "$tag rc=$?"
}
run share_r18 --model resnet18 --batch 256 --image 224 --channels-last --autotune --dtype bf16
run share_vit --model vit_tiny_patch16_224 --batch 256 --image 224 --dtype bf16 --nfp-radius 2 --nfp-measure norm
python3 - <<'PY'
import collections, csv, glob
out = open("gpurun_out/r02_k_train_step_kernel_share.csv", "w")
for tag, what in (("share_r18", "resnet18 224x224 bs256 bf16 channels-last + MIOpen find, NFP cosine R=1 (fused pooling tail)"),
                  ("share_vit", "vit_tiny_patch16_224 bs256 bf16, NFP L2 k=5 on the 14x14 token map (fused pooling tail)")):
    f = glob.glob(f"gpurun_out/prof_{tag}/**/*kernel_trace.csv", recursive=True)[0]
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    fw = [i for i, r in enumerate(rows) if "nfp::fwd" in r["Kernel_Name"]]
    seg = rows[fw[-4]:fw[-1]]       # three whole steps (MIOpen's search kernels of the first steps stay out)
    agg = collections.defaultdict(lambda: [0, 0])
    for r in seg:
        a = agg[r["Kernel_Name"]]
        a[0] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        a[1] += 1
    tot = sum(v[0] for v in agg.values())
    wall = int(seg[-1]["End_Timestamp"]) - int(seg[0]["Start_Timestamp"])
    nfp = {k: v for k, v in agg.items() if "nfp::" in k}
    out.write(f"# {what}: last 3 steps of the kernel trace (rocprofv3): {wall / 3e6:.3f} ms/step wall, GPU busy "
              f"{tot / 3e6:.3f} ms/step; NFP kernels {sum(v[0] for v in nfp.values()) / tot * 100:.3f} % of the busy time\n")
    out.write("ms_per_step,calls_per_step,share_pct,kernel\n")
    top = sorted(agg.items(), key=lambda kv: -kv[1][0])[:10]
    for k, v in top + [kv for kv in nfp.items() if kv not in top]:
        name = k if len(k) < 100 else k[:97] + "..."
        out.write(f"{v[0] / 3e6:.4f},{v[1] / 3:.1f},{v[0] / tot * 100:.3f},\"{name}\"\n")
out.close()
print(open("gpurun_out/r02_k_train_step_kernel_share.csv").read())
PY
