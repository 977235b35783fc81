#!/bin/bash
# A/B builds of libghip.so: recompiles ONE translation unit with extra flags and links it with the
# regular objects into gadget-leicester_amd/variants/libghip_<name>.so (select with GHIP_LIBGHIP).
#   tests/build_variant.sh <name> <file.hip> <flags...>
set -e
cd "$(dirname "$0")/../gadget-leicester_amd"
name=$1; src=$2; shift 2
mkdir -p variants
obj=variants/${src%.hip}_$name.o
hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 "$@" -c csrc/$src -o $obj
objs=""
for f in csrc/*.hip; do
  b=$(basename $f .hip)
  if [ "$b.hip" == "$src" ]; then objs="$objs $obj"; else objs="$objs csrc/$b.o"; fi
done
hipcc -shared -fPIC --offload-arch=gfx950 $objs -Wl,--version-script=csrc/ghip.map -lhipfft -ldl -o variants/libghip_$name.so
echo built variants/libghip_$name.so
