#!/usr/bin/env bash
# Round-3 evidence in one gpurun call (scripts/collect_round3_profiles.py copies what it leaves under gpurun_out/ into profiles/r03_*; DESIGN.md section 5
# names the files): rocprofv3 kernel stats of the headline bench, the saturating batch, config 5 and the large maps;
# FETCH / WRITE_SIZE passes; the forward's PMC counters; the bench lines, incl. `--gpus 2` started by bench.py itself.
set -u
export TMPDIR=/tmp
rm -f gpurun_out/traffic_workloads.json
C5="--batch 256 --channels 192 --size 14 --radius 2 --measure norm --dtype bf16 --layout nhwc"
bash scripts/gpu_profile.sh r03_headline --steps 20 --warmup 5 > gpurun_out/prof_r03_headline.log 2>&1; echo "headline rc=$?"
bash scripts/gpu_profile.sh r03_config5 --steps 20 --warmup 5 $C5 > gpurun_out/prof_r03_config5.log 2>&1; echo "config5 rc=$?"
bash scripts/gpu_profile.sh r03_b4096 --steps 4 --warmup 2 --batch 4096 > gpurun_out/prof_r03_b4096.log 2>&1; echo "b4096 rc=$?"
mkdir -p gpurun_out/prof_r03_bigmaps
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r03_bigmaps -o trace -- \
  python3 scripts/run_bigmaps_for_rocprof.py > gpurun_out/prof_r03_bigmaps/run.log 2>&1; echo "bigmaps rocprof rc=$?"
bash scripts/gpu_traffic.sh > gpurun_out/traffic_headline.log 2>&1; echo "traffic headline rc=$?"
bash scripts/gpu_traffic.sh --batch 4096 > gpurun_out/traffic_b4096.log 2>&1; echo "traffic b4096 rc=$?"
bash scripts/gpu_pmc_fwd.sh > gpurun_out/r03_fwd_pmc_after.log 2>&1; echo "fwd pmc rc=$?"; cp gpurun_out/r03_fwd_pmc.csv gpurun_out/r03_fwd_pmc_after.csv
timeout -k 10 400 python bench.py --steps 20 --warmup 5 > gpurun_out/r03_bench_20.json 2> gpurun_out/r03_bench.err; echo "bench20 rc=$?"
timeout -k 10 400 python bench.py > gpurun_out/r03_bench_200.json 2>> gpurun_out/r03_bench.err; echo "bench200 rc=$?"
NFP_BENCH_BACKEND=gloo timeout -k 10 300 python bench.py --gpus 2 --no-cpu-baseline --steps 20 --warmup 5 > gpurun_out/r03_bench_gpus2_gloo.json 2> gpurun_out/r03_bench_gpus2.err; echo "bench --gpus 2 (gloo, self-launched) rc=$?"
cut -c1-400 gpurun_out/r03_bench_200.json; cut -c1-300 gpurun_out/r03_bench_gpus2_gloo.json
# round-2 kernels (commit bee0a3b, built into neighbour_feature_pooling_amd/ab/ by hand) beside this round's, same box,
# alternating processes: the headline, the per-GPU batch of config 4, the saturating batch
if [ -f neighbour_feature_pooling_amd/ab/manifest.json ]; then
  for shp in 64,512,7,1,cosine 256,512,7,1,cosine 4096,512,7,1,cosine; do
    echo "== $shp"; AB_COLD=1 AB_SHAPE=$shp timeout -k 10 300 python scripts/ab_flags.py --run 2>&1 | grep "fwd" | cut -c1-170
  done > gpurun_out/r03_ab_round2_vs_round3.txt
  cat gpurun_out/r03_ab_round2_vs_round3.txt
fi
timeout -k 10 400 python scripts/sweep_bigmaps.py gpurun_out/r03_bigmaps_final.jsonl > gpurun_out/r03_bigmaps_final.log 2>&1; echo "bigmaps sweep rc=$?"
timeout -k 10 200 python scripts/sweep.py > gpurun_out/r03_shape_sweep.jsonl 2>&1; echo "shape sweep rc=$?"
timeout -k 10 200 python scripts/gpu_fused_callers.py > gpurun_out/r03_fused_callers.jsonl 2>&1; echo "fused callers rc=$?"
# ---- the rewritten row-band kernels (one thread per padded position) ---------------------------------------------------
bash scripts/gpu_pmc_tile.sh 16 112 > gpurun_out/pmc_tile_after.log 2>&1; echo "tile pmc rc=$?"
# grad_x stores of the row-band backward: the product, the same without its stores, and round 2's any-geometry kernels
if [ -f neighbour_feature_pooling_amd/ab/manifest.json ]; then
  for shp in 256,64,56,1,cosine 256,64,56,1,cosine,nhwc 256,40,28,1,cosine 256,40,28,1,cosine,nhwc 256,16,112,1,cosine; do
    echo "== $shp"; AB_SHAPE=$shp timeout -k 10 300 python scripts/ab_flags.py --run 2>&1 | grep "fwd" | cut -c1-170
  done > gpurun_out/r03_tile_backward_stores_ab.txt
  cat gpurun_out/r03_tile_backward_stores_ab.txt
fi
timeout -k 10 400 python scripts/tile_vs_table_kernels.py gpurun_out/r03_tile_vs_table.jsonl > gpurun_out/r03_tile_vs_table.log 2>&1; echo "tile vs table rc=$?"
bash scripts/gpu_multistage_step.sh > gpurun_out/r03_multistage.log 2>&1; echo "multistage rc=$?"
bash scripts/gpu_train_steps.sh > gpurun_out/r03_train.log 2>&1; echo "train steps rc=$?"
