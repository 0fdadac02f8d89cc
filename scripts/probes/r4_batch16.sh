#!/bin/bash
set -o pipefail
OUT=$GRAFT_REPO_ROOT/gpurun_out/r4; mkdir -p $OUT; cd $GRAFT_REPO_ROOT
B="python bench.py --no-op-table --no-cpu-baseline --no-side-runs"
for t in 0 32768 131072 262144 100000000; do echo "lift-branch-max-rows $t"; $B --steps 20 --lift-branch-max-rows $t 2>>$OUT/b16.err | python scripts/probes/show_bench.py - --short; done
for t in 0 32768 131072; do echo "f1 lift-branch-max-rows $t"; $B --frames-per-gpu 1 --steps 32 --lift-branch-max-rows $t 2>>$OUT/b16.err | python scripts/probes/show_bench.py - --short; done
for t in 0 131072 262144; do echo "f2 lift-branch-max-rows $t"; $B --frames-per-gpu 2 --steps 32 --lift-branch-max-rows $t 2>>$OUT/b16.err | python scripts/probes/show_bench.py - --short; done
