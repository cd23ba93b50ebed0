#!/bin/bash
# Builds an experimental variant of libbPSW_hip.so with extra compiler flags:
#   tools/build_variant.sh m2 -DBPSW_EXT_VECTOR_CONTROL=2   ->  cloud-scale-bwamem_amd/lib_exp/libbPSW_hip_m2.so
# and run anything against it with BPSW_LIB=<that path>.
set -e
name=$1; shift
cd "$(dirname "$0")/../cloud-scale-bwamem_amd"
mkdir -p build_$name lib_exp
for f in csrc/*.hip csrc/*.cpp; do
  case "$f" in csrc/bpsw_synth.cpp|csrc/bpsw_feeder.cpp) continue;; esac  # harness library, not part of the product
  [ -f "$f" ] || continue
  o=build_$name/$(basename ${f%.*}).o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../include -Icsrc -Wno-unused-function "$@" -c $f -o $o &
  while [ $(jobs -r | wc -l) -ge 6 ]; do sleep 0.2; done
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o lib_exp/libbPSW_hip_$name.so build_$name/*.o -lpthread
rm -rf build_$name
echo built lib_exp/libbPSW_hip_$name.so
