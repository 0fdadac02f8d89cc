#!/bin/bash
set -o pipefail
OUT=$GRAFT_REPO_ROOT/gpurun_out/r4; mkdir -p $OUT; cd $GRAFT_REPO_ROOT
B="python bench.py --no-op-table --no-cpu-baseline --no-side-runs"
for f in 1 1 2 4; do $B --frames-per-gpu $f --steps 32 2>>$OUT/b15.err | python scripts/probes/show_bench.py - --short; done
$B --steps 20 2>>$OUT/b15.err | python scripts/probes/show_bench.py - --short
$B --steps 20 2>>$OUT/b15.err | python scripts/probes/show_bench.py - --short
