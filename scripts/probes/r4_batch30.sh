#!/bin/bash
set -o pipefail
OUT=$GRAFT_REPO_ROOT/gpurun_out/r4b; mkdir -p $OUT; cd $GRAFT_REPO_ROOT
timeout -k 10 500 python -m pytest tests -m gpu -x -q -k "pointcnn or rpn or graph or rcnn or two_stage or inference or ddp or bench" > $OUT/pytest_local.log 2>&1; tail -3 $OUT/pytest_local.log
B="python bench.py --no-op-table --no-cpu-baseline --no-side-runs"
$B --steps 20 2>>$OUT/b30.err | python scripts/probes/show_bench.py - --short
$B --frames-per-gpu 1 --steps 32 2>>$OUT/b30.err | python scripts/probes/show_bench.py - --short
$B --frames-per-gpu 2 --steps 32 2>>$OUT/b30.err | python scripts/probes/show_bench.py - --short
