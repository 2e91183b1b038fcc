#!/usr/bin/env bash
# MobileNetV3 MultiStage-NFP (texture_pooling.py:211-268) train step, 224x224 bs 256: the large-map kernels of round 3
# against the any-geometry kernels they replace (NFP_FORCE_GENERIC=1), f32 NCHW and bf16 channels-last.
set -u
: > gpurun_out/r03_multistage.jsonl
for a in "" "--channels-last --autotune --dtype bf16"; do
  for env in "" "NFP_FORCE_GENERIC=1"; do
    echo "# ${env:-row-band + table kernels} $a" >> gpurun_out/r03_multistage.jsonl
    env $env timeout -k 10 300 python -m neighbour_feature_pooling_amd.train --model mobilenetv3_multistage --batch 256 --image 224 $a --steps 10 --warmup 3 >> gpurun_out/r03_multistage.jsonl 2>> gpurun_out/r03_multistage.err
  done
done
cat gpurun_out/r03_multistage.jsonl
