#!/bin/bash
# Dev tool (GPU box): time the wino4 experiment variants (tools/build_variant.sh w4expN kernels_wino4.hip -DW4_EXP=N)
set -e
shapes="32,128,128,104,80 32,32,32,432,320 32,32,32,368,256 32,256,256,56,40"
echo "== baseline"; timeout -k 10 120 python tools/time_tiles.py $shapes -- 42 43
for e in ${EXPS:-1 2 3 4 5}; do
  echo "== W4_EXP=$e"
  CCVPE_LIB_PATH=variants/libccvpe_w4exp$e.so timeout -k 10 120 python tools/time_tiles.py $shapes -- 42 43
done
