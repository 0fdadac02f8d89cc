#!/bin/bash
# dropout fused into the BatchNorm passes + narrow head gradient: tests, then the step at eight frames and one frame
set -o pipefail
OUT=$GRAFT_REPO_ROOT/gpurun_out/r4b; mkdir -p $OUT; cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -m pytest tests -m gpu -x -q -k "dropout or narrow or bn" > $OUT/pytest_drop.log 2>&1; tail -5 $OUT/pytest_drop.log
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $OUT/pytest_all.log 2>&1; tail -3 $OUT/pytest_all.log
B="python bench.py --no-op-table --no-cpu-baseline --no-side-runs"
$B --steps 20 2>>$OUT/b20.err | python scripts/probes/show_bench.py - --short
$B --frames-per-gpu 1 --steps 32 2>>$OUT/b20.err | python scripts/probes/show_bench.py - --short
