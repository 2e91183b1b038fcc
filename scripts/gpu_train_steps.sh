#!/usr/bin/env bash
# End-to-end train steps of BASELINE.json configs 3-5 on ONE GPU (synthetic data, random-init trunks): images/s per config.
set -u
: > gpurun_out/r04_train.jsonl
for a in "--model resnet18 --batch 256 --image 64 --in-chans 13" "--model resnet18 --batch 256 --image 224" \
         "--model resnet18 --batch 256 --image 224 --channels-last --autotune --dtype bf16" \
         "--model vit_tiny_patch16_224 --batch 256 --image 224 --dtype bf16 --nfp-radius 2 --nfp-measure norm" \
         "--model mobilenetv3_large_100 --batch 256 --image 224 --channels-last --autotune --dtype bf16"; do
  timeout -k 10 300 python -m neighbour_feature_pooling_amd.train $a --steps 10 --warmup 3 >> gpurun_out/r04_train.jsonl 2>> gpurun_out/r04_train.err
done
cat gpurun_out/r04_train.jsonl
