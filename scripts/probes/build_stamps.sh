#!/bin/bash
# Diagnostic build of the library with in-kernel stamps in the cell ball-query kernel and the bucketed FPS kernel (never shipped / never loaded by the package).
set -e
cd "$(dirname "$0")/../../heterofusionrcnn_amd/csrc"
mkdir -p build_stamps
for f in hf_api.cpp sampling.hip grouping.hip ballquery.hip ballquery_sorted.hip interpolate.hip bev_iou.hip cropping.hip mlp.hip gemm.hip glue.hip xconv.hip optim.hip; do
  o=build_stamps/${f%.*}.o
  if [ "$f" = "ballquery.hip" ] || [ "$f" = "sampling.hip" ] || [ ! -f "$o" ] || [ "$f" -nt "$o" ] || [ hf_common.h -nt "$o" ] || [ ../../include/hfops.h -nt "$o" ]; then
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -fvisibility=hidden -std=c++17 -I../../include -DHF_QBP_STAMPS -DHF_FPS_STAMPS -x hip -c $f -o $o &
  fi
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../scripts/probes/libhfops_stamps.so build_stamps/*.o
