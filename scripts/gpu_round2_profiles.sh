#!/usr/bin/env bash
# Round-2 evidence in one gpurun call: rocprofv3 kernel stats + FETCH/WRITE_SIZE passes for the headline, config 5 and
# the saturating batch; shape sweep; 1-GPU train steps; the default bench line.  Copy what it leaves under
# gpurun_out/ into profiles/ (see DESIGN.md section 5 for the file names).
# The other round-2 files come from their own scripts: r02_j  scripts/gpu_matrix_core_ab.sh (after scripts/ab_flags.py
# --build and a library of the "before" commit in neighbour_feature_pooling_amd/ab/), r02_k  scripts/gpu_train_share.sh,
# r02_l  scripts/gpu_fused_callers.py, r02_i  scripts/ab_flags.py with -DNFP_BWD_STORE_AUX variants.
set -u
rm -f gpurun_out/traffic_workloads.json
C5="--batch 256 --channels 192 --size 14 --radius 2 --measure norm --dtype bf16 --layout nhwc"
bash scripts/gpu_profile.sh r02_headline --steps 20 --warmup 5 > gpurun_out/prof_r02_headline.log 2>&1; echo "headline rc=$?"
bash scripts/gpu_profile.sh r02_config5 --steps 20 --warmup 5 $C5 > gpurun_out/prof_r02_config5.log 2>&1; echo "config5 rc=$?"
bash scripts/gpu_profile.sh r02_b4096 --steps 4 --warmup 2 --batch 4096 > gpurun_out/prof_r02_b4096.log 2>&1; echo "b4096 rc=$?"
bash scripts/gpu_traffic.sh > gpurun_out/traffic_headline.log 2>&1; echo "traffic headline rc=$?"
bash scripts/gpu_traffic.sh $C5 > gpurun_out/traffic_config5.log 2>&1; echo "traffic config5 rc=$?"
bash scripts/gpu_traffic.sh --batch 4096 > gpurun_out/traffic_b4096.log 2>&1; echo "traffic b4096 rc=$?"
timeout -k 10 300 python scripts/sweep.py > gpurun_out/r02_sweep.jsonl 2> gpurun_out/r02_sweep.err; echo "sweep rc=$?"
: > gpurun_out/r02_train.jsonl
for a in "--model resnet18 --batch 256 --image 64 --in-chans 13" "--model resnet18 --batch 256 --image 224" \
         "--model resnet18 --batch 256 --image 224 --channels-last --autotune --dtype bf16" \
         "--model vit_tiny_patch16_224 --batch 256 --image 224 --dtype bf16 --nfp-radius 2 --nfp-measure norm"; do
  timeout -k 10 300 python -m neighbour_feature_pooling_amd.train $a --steps 10 --warmup 3 >> gpurun_out/r02_train.jsonl 2>> gpurun_out/r02_train.err
done
echo "train rc=$?"
timeout -k 10 400 python bench.py --steps 20 --warmup 5 > gpurun_out/r02_bench_20.json 2> gpurun_out/r02_bench.err; echo "bench20 rc=$?"
timeout -k 10 400 python bench.py > gpurun_out/r02_bench_200.json 2>> gpurun_out/r02_bench.err; echo "bench200 rc=$?"
tail -n 5 gpurun_out/r02_sweep.jsonl gpurun_out/r02_train.jsonl
