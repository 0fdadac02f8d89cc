#!/bin/bash
# Diagnostic build of the C-ABI library: -DHF_DIAG turns HF_DIAG_INT(name, dflt) into an environment read, so the probes can
# force tile shapes / phase exits (HF_QBP_*, HF_BQ_G, HF_BEV_STOP, HF_NMS_STOP, HF_GEMM_ROUNDS).  Outputs with a *_STOP set
# are INVALID.  Use it with HFOPS_LIBRARY=$PWD/heterofusionrcnn_amd/csrc/build_diag/libhfops_diag.so; the product
# library (make -C heterofusionrcnn_amd/csrc) has none of this.
set -e
cd "$(dirname "$0")/../../heterofusionrcnn_amd/csrc"
mkdir -p build_diag
for f in hf_api.cpp sampling.hip grouping.hip ballquery.hip ballquery_sorted.hip interpolate.hip bev_iou.hip cropping.hip mlp.hip gemm.hip glue.hip xconv.hip optim.hip; do
    o=build_diag/${f%.*}.o
    if [ ! -f $o ] || [ $f -nt $o ] || [ hf_common.h -nt $o ] || [ bq_common.h -nt $o ]; then
        /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -fvisibility=hidden -std=c++17 -I../../include -DHF_DIAG ${HF_DIAG_EXTRA} -x hip -c $f -o $o &
    fi
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build_diag/libhfops_diag.so build_diag/*.o
echo build_diag/libhfops_diag.so
