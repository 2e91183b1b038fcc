for shp in "64,512,7,1,cosine" "256,192,14,2,norm" "256,192,14,1,cosine"; do
  echo "== $shp"; AB_SHAPE=$shp python scripts/ab_flags.py "" 2>&1 | grep "^\[.*fwd" | head -1
done
