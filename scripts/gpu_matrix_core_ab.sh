set -e
o=gpurun_out/r02_j_matrix_core_backward_ab.txt
: > $o
for shp in 256,192,14,2,norm,bf16,nhwc 256,192,14,2,norm,bf16 64,512,7,1,cosine,bf16 64,512,7,1,cosine,bf16,nhwc 256,512,7,1,cosine,bf16,nhwc 64,512,7,1,cosine; do
  echo "== AB_SHAPE=$shp (B,C,S,R,measure[,dtype][,layout]); 'cold' = rotating sets > 256 MiB, else one resident set" >> $o
  AB_COLD=1 AB_SHAPE=$shp timeout -k 10 300 python scripts/ab_flags.py --run 2>/dev/null | grep -v amdgpu >> $o
done
echo "== in-kernel stamps of the final config-5 backward (scripts/diag_stamps.py, diagnostic build; stamps 7..12: Xt staged, round 0 barrier / zeroed / scattered / tiles done, round 1 barrier)" >> $o
DIAG_BF16=1 DIAG_NHWC=1 timeout -k 10 200 python scripts/diag_stamps.py 256 192 14 2 norm 2>/dev/null | grep -v "amdgpu\|phase 4->5\|phase 5->6" >> $o
tail -n 12 $o
