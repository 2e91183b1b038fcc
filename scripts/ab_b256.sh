for shp in "256,512,7,1,cosine" "1024,512,7,1,cosine"; do
  echo "== $shp"; AB_SHAPE=$shp python scripts/ab_flags.py "" "-DNFP_BWD_WGS=512" "-DNFP_BWD_WGS=1024" "-DNFP_BWD_SLAB_KB=30" 2>&1 | grep "^\[.*fwd" | head -4
done
