#!/usr/bin/env bash
# Round-4 evidence in one gpurun call (scripts/collect_round4_profiles.py copies what it leaves under gpurun_out/ into profiles/r04_*; DESIGN.md
# section 5 names the files): rocprofv3 kernel stats of the headline bench, the saturating batch, config 5 and the large maps; FETCH / WRITE_SIZE
# passes; the forward's PMC counters; the bench lines; the sweeps.
set -u
export TMPDIR=/tmp
rm -f gpurun_out/traffic_workloads.json
C5="--batch 256 --channels 192 --size 14 --radius 2 --measure norm --dtype bf16 --layout nhwc"
bash scripts/gpu_profile.sh r04_headline --steps 20 --warmup 5 > gpurun_out/prof_r04_headline.log 2>&1; echo "headline rc=$?"
bash scripts/gpu_profile.sh r04_config5 --steps 20 --warmup 5 $C5 > gpurun_out/prof_r04_config5.log 2>&1; echo "config5 rc=$?"
bash scripts/gpu_profile.sh r04_b4096 --steps 4 --warmup 2 --batch 4096 > gpurun_out/prof_r04_b4096.log 2>&1; echo "b4096 rc=$?"
mkdir -p gpurun_out/prof_r04_bigmaps
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r04_bigmaps -o trace -- \
  python3 scripts/run_bigmaps_for_rocprof.py > gpurun_out/prof_r04_bigmaps/run.log 2>&1; echo "bigmaps rocprof rc=$?"
bash scripts/gpu_traffic.sh > gpurun_out/traffic_headline.log 2>&1; echo "traffic headline rc=$?"
bash scripts/gpu_traffic.sh --batch 4096 > gpurun_out/traffic_b4096.log 2>&1; echo "traffic b4096 rc=$?"
bash scripts/gpu_pmc_fwd.sh > gpurun_out/r04_fwd_pmc.log 2>&1; echo "fwd pmc rc=$?"; cp gpurun_out/r03_fwd_pmc.csv gpurun_out/r04_fwd_pmc.csv
timeout -k 10 400 python bench.py --steps 20 --warmup 5 > gpurun_out/r04_bench_20.json 2> gpurun_out/r04_bench.err; echo "bench20 rc=$?"
timeout -k 10 400 python bench.py > gpurun_out/r04_bench_200.json 2>> gpurun_out/r04_bench.err; echo "bench200 rc=$?"
cut -c1-400 gpurun_out/r04_bench_200.json
timeout -k 10 400 python scripts/sweep_bigmaps.py gpurun_out/r04_bigmaps_final.jsonl > gpurun_out/r04_bigmaps_final.log 2>&1; echo "bigmaps sweep rc=$?"
timeout -k 10 200 python scripts/sweep.py > gpurun_out/r04_shape_sweep.jsonl 2>&1; echo "shape sweep rc=$?"
timeout -k 10 300 python scripts/gpu_fused_callers.py > gpurun_out/r04_fused_callers_final.jsonl 2>&1; echo "fused callers rc=$?"
timeout -k 10 400 python scripts/sweep_measures.py gpurun_out/r04_all_measures_final.jsonl > gpurun_out/r04_all_measures_final.log 2>&1; echo "measures rc=$?"
bash scripts/gpu_train_steps.sh > gpurun_out/r04_train.log 2>&1; echo "train steps rc=$?"
