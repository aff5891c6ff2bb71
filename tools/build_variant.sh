#!/bin/bash
# Build variants/libmpe_<name>.so from one source of the working tree (e.g. mpe_wave.hip, mpe_small_f32_core.hip) with extra flags,
# linking the other objects of the last full build.  usage: tools/build_variant.sh name file.hip [-Dflags]
set -e
name=$1; src=$2; shift 2
cd "$(dirname "$0")/../multiagent-particle-envs_amd/csrc"
mkdir -p ../../variants
obj=/tmp/variant_$name.o
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off "$@" -c $src -o $obj
objs=""
for o in mpe_small_f64_pp mpe_small_f64_collector mpe_small_f64_core mpe_small_f64_classic mpe_small_f32_pp mpe_small_f32_collector mpe_small_f32_core mpe_small_f32_classic mpe_small mpe_wave mpe_consensus mpe_gae mpe_reset mpe_abi; do
  if [ "$o.hip" = "$src" ]; then objs="$objs $obj"; else objs="$objs build/$o.o"; fi
done
hipcc --offload-arch=gfx950 -shared -fPIC -o ../../variants/libmpe_$name.so $objs
