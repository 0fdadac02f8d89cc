#!/bin/bash
# Diagnostic build of the library with extra -D flags for ballquery.hip: build_variant.sh NAME -DFOO=1 ...  -> scripts/probes/libhfops_NAME.so
set -e
name=$1; shift
cd "$(dirname "$0")/../../heterofusionrcnn_amd/csrc"
make -s >/dev/null
mkdir -p build_variants
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -fvisibility=hidden -std=c++17 -I../../include "$@" -x hip -c ballquery.hip -o build_variants/ballquery_$name.o
objs=$(ls build/*.o | grep -v ballquery)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../scripts/probes/libhfops_$name.so $objs build_variants/ballquery_$name.o
