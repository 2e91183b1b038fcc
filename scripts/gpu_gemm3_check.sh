#!/bin/bash
# matrix-core backward, second form (nfp_fast.h::bwd_gemm_phase3): parity tests, in-kernel stamps, A/B against round 3's form
set -u
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "matrix_core or config5_nfp or fused" > gpurun_out/gemm3_tests.log 2>&1
rc=$?; tail -2 gpurun_out/gemm3_tests.log
[ $rc -ne 0 ] && exit $rc
DIAG_BF16=1 DIAG_NHWC=1 timeout -k 10 200 python scripts/diag_stamps.py 256 192 14 2 norm > gpurun_out/gemm3_stamps.txt 2>&1 &&
AB_COLD=0 bash scripts/ab_shapes.sh gpurun_out/gemm3_ab.txt ${SHAPES:-"256,192,14,2,norm,bf16,nhwc" "256,192,14,2,norm,bf16" "256,512,7,1,cosine,bf16,nhwc" "64,512,7,1,cosine,bf16"} && grep -v amdgpu.ids gpurun_out/gemm3_ab.txt | grep -v cold
