#!/bin/bash
# Build variants/lib_<name>.so with extra hipcc flags (A/B experiments: PSDR_HIP_LIB=$PWD/variants/lib_<name>.so).
# usage: tools/build_variant_lib.sh <name> [-DPSDR_WAVES_REV=1 ...]
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
name=$1; shift
obj=/tmp/psdr_variant_$name; mkdir -p $obj $ROOT/variants
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -fno-hip-fp32-correctly-rounded-divide-sqrt -fgpu-flush-denormals-to-zero -fno-slp-vectorize -freciprocal-math -I$ROOT/include"
cd $ROOT/psdr-cuda_amd/csrc
for v in 0 1 2 3 4 6; do hipcc $FLAGS "$@" -DPSDR_VARIANT_FLAGS=$v -c psdr_variant.hip -o $obj/variant$v.o & done
hipcc $FLAGS "$@" -c psdr_hip.hip -o $obj/host.o &
wait
hipcc --offload-arch=gfx950 -shared -fPIC $obj/*.o -o $ROOT/variants/lib_$name.so
echo built variants/lib_$name.so
