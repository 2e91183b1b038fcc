#!/usr/bin/env bash
# End-to-end train step of the BASELINE configs under the layout / autotune options of train.py (1 GPU).
set -u
mkdir -p gpurun_out
out=gpurun_out/train_variants.jsonl; : > $out
run() { timeout -k 10 400 python -m neighbour_feature_pooling_amd.train "$@" 2>/dev/null | grep '^{' | tee -a $out; }
for extra in "" "--autotune" "--channels-last" "--channels-last --autotune"; do
  run --model resnet18 --batch 256 --image 64 --in-chans 13 --steps 20 --warmup 5 $extra
  run --model resnet18 --batch 256 --image 224 --steps 10 --warmup 4 $extra
done
run --model resnet18 --batch 256 --image 224 --steps 10 --warmup 4 --dtype bf16 --channels-last --autotune
run --model vit_tiny_patch16_224 --batch 256 --dtype bf16 --nfp-radius 2 --nfp-measure norm --steps 10 --warmup 4 --autotune
run --model mobilenetv3_large_100 --batch 256 --image 224 --steps 6 --warmup 3 --channels-last --autotune
