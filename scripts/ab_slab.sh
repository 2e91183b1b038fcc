set -e
for shp in "64,512,7,1,cosine" "1024,512,7,1,cosine"; do
  echo "== $shp"
  AB_SHAPE=$shp python scripts/ab_flags.py "" "-DNFP_FWD_SLAB_KB=64" "-DNFP_FWD_SLAB_KB=40" "-DNFP_BWD_SLAB_KB=30" "-DNFP_BWD_SLAB_KB=120" 2>&1 | grep "^\[.*fwd" | head -5
done
