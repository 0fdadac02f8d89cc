#!/bin/bash
# Diagnostic build of the library with extra -D flags for bev_iou.hip: build_bev_variant.sh NAME -DFOO=1 ... -> scripts/probes/libhfops_NAME.so
set -e
name=$1; shift
cd "$(dirname "$0")/../../heterofusionrcnn_amd/csrc"
make -s >/dev/null
mkdir -p build_variants
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -fvisibility=hidden -std=c++17 -I../../include "$@" -x hip -c bev_iou.hip -o build_variants/bev_iou_$name.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../scripts/probes/libhfops_$name.so $(ls build/*.o | grep -v bev_iou) build_variants/bev_iou_$name.o
