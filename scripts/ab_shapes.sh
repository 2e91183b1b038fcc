#!/bin/bash
# A/B of the libraries in neighbour_feature_pooling_amd/ab/ (scripts/ab_flags.py --build ...) over several shapes in one GPU call.
# usage: bash scripts/ab_shapes.sh out.txt "B,C,S,R,measure[,bf16][,nhwc]" ...
out=$1; shift
mkdir -p gpurun_out
: > "$out"
for shp in "$@"; do
  echo "=== $shp" >> "$out"
  AB_COLD=${AB_COLD:-1} AB_SHAPE="$shp" timeout -k 10 600 python scripts/ab_flags.py --run >> "$out" 2>&1 || { echo "FAILED $shp" >> "$out"; exit 1; }
done
