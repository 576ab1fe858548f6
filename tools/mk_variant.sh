#!/bin/bash
# usage: tools/mk_variant.sh TAG 'sed expression' [extra hipcc flags]  ->  reart_amd/csrc/libreart_hip_TAG.so from lap_mw.hip with the
# expression applied (experiment variants of constants that are plain #defines in the product source); never the product path
set -e
cd "$(dirname "$0")/../reart_amd/csrc"
tag=$1; expr=$2; shift 2
mkdir -p build
sed -e "$expr" lap_mw.hip > build/lap_mw_$tag.hip
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -Wall -Wno-unused-function -I. "$@" -c build/lap_mw_$tag.hip -o build/lap_mw_$tag.o
objs=""
for f in *.hip; do [ "$f" != lap_mw.hip ] && objs="$objs build/${f%.hip}.o"; done      # every product object but lap_mw.o (make first)
if [[ " $* " == *REART_PRUNE_PHASE* ]]; then objs="${objs/build\/lap.o/build/lap_phase.o}"; objs="${objs/build\/prune.o/build/prune_phase.o}"; fi
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libreart_hip_$tag.so $objs build/lap_mw_$tag.o
echo built libreart_hip_$tag.so
