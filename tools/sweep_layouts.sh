for shp in "32 540 960 128" "32 540 960 64" "8 1536 2048 256" "32 375 1242 192" "16 1536 2048 256"; do
  for e in "VPPX_VERT=0" "VPPX_VERT=3 VPPX_V3_PPW=8" "VPPX_VERT=3 VPPX_V3_PPW=16"; do
    env $e timeout 300 python tools/agg_probe.py $shp 2>&1 | tail -1
  done
done
