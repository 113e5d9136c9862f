#!/bin/bash
# Build a libpss variant with extra -D flags for interleaved A/B runs (tests/tools/ab.py):
#   tests/tools/variant.sh name "-DPSS_RR_ROWS=16"   ->  variants/libpss_name.so   (git-ignored, travels with gpurun)
set -e
name=$1; flags=$2
root=$(cd "$(dirname "$0")/../.." && pwd)
src=$root/pysubstringsearch_amd/csrc
obj=$root/variants/obj_$name
mkdir -p $obj
for f in radix_sort msd_sort rle_build sa_build search; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $flags -x hip -c $src/$f.hip -o $obj/$f.o &
done
for f in common capi corpus; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC $flags -c $src/$f.cpp -o $obj/$f.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $root/variants/libpss_$name.so $obj/*.o
rm -rf $obj
echo built variants/libpss_$name.so
