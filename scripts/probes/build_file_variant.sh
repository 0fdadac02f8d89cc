#!/bin/bash
# Diagnostic build of the library with extra -D flags for ONE source file: build_file_variant.sh FILE NAME -DFOO=1 ... -> scripts/probes/libhfops_NAME.so
set -e
file=$1; name=$2; shift; shift
cd "$(dirname "$0")/../../heterofusionrcnn_amd/csrc"
make -s >/dev/null
mkdir -p build_variants
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -fvisibility=hidden -std=c++17 -I../../include "$@" -x hip -c $file.hip -o build_variants/${file}_$name.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../scripts/probes/libhfops_$name.so $(ls build/*.o | grep -v "build/$file.o") build_variants/${file}_$name.o
