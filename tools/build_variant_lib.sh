#!/bin/bash
# Build variants/lib_<name>.so with extra hipcc flags (A/B experiments: PSDR_HIP_LIB=$PWD/variants/lib_<name>.so).
# usage: [ONLY="4 6"] tools/build_variant_lib.sh <name> [-DPSDR_WAVES_REV=1 ...]
#   ONLY: rebuild these flag sets only (+ never the host unit); the other objects come from the product build (psdr-cuda_amd/lib/obj)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
name=$1; shift
obj=/tmp/psdr_variant_$name; rm -rf $obj; mkdir -p $obj $ROOT/variants
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -fno-hip-fp32-correctly-rounded-divide-sqrt -fgpu-flush-denormals-to-zero -fno-slp-vectorize -freciprocal-math -fapprox-func -I$ROOT/include"
cd $ROOT/psdr-cuda_amd/csrc
ALL="0 1 2 3 4 6 8 10"
#   ONLY=host: rebuild the host unit (psdr_hip.hip: C ABI, k_trace, k_wf_trace) only
if [ -n "${ONLY:-}" ]; then cp $ROOT/psdr-cuda_amd/lib/obj/*.o $obj/; rm -f $obj/*_defect.o; SET="$ONLY"; else SET="$ALL"; fi
#   ONLY="4 host": both
for v in $SET; do if [ "$v" = "host" ]; then hipcc $FLAGS "$@" -c psdr_hip.hip -o $obj/host.o & fi; done
SET=$(echo $SET | sed "s/host//")
for v in $SET; do hipcc $FLAGS "$@" -DPSDR_VARIANT_FLAGS=$v -c psdr_variant.hip -o $obj/variant$v.o & done
if [ -z "${ONLY:-}" ]; then hipcc $FLAGS "$@" -c psdr_hip.hip -o $obj/host.o & hipcc $FLAGS "$@" -c psdr_tables.hip -o $obj/tables.o & fi
wait
hipcc --offload-arch=gfx950 -shared -fPIC $obj/*.o -o $ROOT/variants/lib_$name.so
echo built variants/lib_$name.so
