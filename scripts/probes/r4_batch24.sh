#!/bin/bash
set -o pipefail
OUT=$GRAFT_REPO_ROOT/gpurun_out/r4b; mkdir -p $OUT; cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -m pytest tests -m gpu -x -q -k "nms or iou or two_stage or rcnn or inference or fuzz" > $OUT/pytest_nms.log 2>&1; tail -3 $OUT/pytest_nms.log
(cd scripts/probes && timeout -k 10 300 python nms_variant.py libhfops_prev.so libhfops_new.so 2>&1 | tee $OUT/nms_variants.txt)
timeout -k 10 200 python scripts/bev_nms_timing.py 2>/dev/null | tee $OUT/bev_nms_timing_b24.json
