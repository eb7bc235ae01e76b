#!/bin/bash
# Dev tool (GPU box): time Winograd F(4x4) tiles under library variants.  usage: VARIANTS="a b" tools/run_variants.sh
set -e
shapes=${SHAPES:-"32,128,128,104,80 32,32,32,432,320 32,32,32,368,256 32,256,256,56,40 32,64,64,200,160 32,128,128,88,64"}
tiles=${TILES:-"42 43"}
echo "== baseline"; timeout -k 10 120 python tools/time_tiles.py $shapes -- $tiles
for v in $VARIANTS; do
  echo "== $v"
  CCVPE_LIB_PATH=variants/libccvpe_$v.so timeout -k 10 120 python tools/time_tiles.py $shapes -- $tiles
done
