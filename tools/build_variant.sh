#!/bin/bash
# Dev tool: link a variant of the library with one source rebuilt under extra flags.
#   tools/build_variant.sh <name> <source.hip> <flags...>   ->  variants/libccvpe_<name>.so  (use with CCVPE_LIB_PATH)
set -e
name=$1; src=$2; shift 2
cd "$(dirname "$0")/.."
python -m ccvpe_amd.build > /dev/null
mkdir -p variants
c=ccvpe_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-function "$@" -c $c/$src -o variants/${name}.o
objs=""
for f in $c/*.o; do
  if [ "$(basename $f)" != "${src%.hip}.o" ]; then objs="$objs $f"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -Wl,--no-undefined -o variants/libccvpe_${name}.so $objs variants/${name}.o
echo variants/libccvpe_${name}.so
