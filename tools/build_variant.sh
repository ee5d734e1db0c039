#!/bin/bash
# build an A/B variant of the library: tools/build_variant.sh <name> <extra hipcc -D flags...>  ->  tools/_bin/libnereus_hip_<name>.so
# (only the fp32 Muller unit is recompiled with the flags; select it with NEREUS_HIP_LIB=<path>)
set -e
cd "$(dirname "$0")/../nereus_amd/csrc"
NAME=$1; shift
mkdir -p build ../../tools/_bin
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -w "$@" -c -o build/var_$NAME.o nrs_inst_f32_muller.hip
/opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared -o ../../tools/_bin/libnereus_hip_$NAME.so build/nrs_abi.o build/nrs_boundary.o build/nrs_debug.o build/var_$NAME.o build/nrs_inst_f32_monaghan.o build/nrs_inst_f64_muller.o build/nrs_inst_f64_monaghan.o
echo built tools/_bin/libnereus_hip_$NAME.so
