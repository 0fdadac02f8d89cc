#!/bin/bash
# coordinate-only branches running ahead on two side streams: tests, replay-only probe, bench at 1/2/4/8 frames
set -o pipefail
OUT=$GRAFT_REPO_ROOT/gpurun_out/r4; mkdir -p $OUT; cd $GRAFT_REPO_ROOT
timeout -k 10 500 python -m pytest tests/test_graph_step.py tests/test_ops_gpu.py -q -m gpu -k "graph or step or pointcnn or xconv or rpn or branch" > $OUT/t14.log 2>&1; rc=$?; tail -3 $OUT/t14.log
if [ $rc -ne 0 ]; then exit 1; fi
HF_ONLY_AB=1 timeout -k 10 200 python scripts/probes/replay_only.py 1 on 2>&1 | grep -v amdgpu
B="python bench.py --no-op-table --no-cpu-baseline --no-side-runs"
for f in 1 1 2 4; do $B --frames-per-gpu $f --steps 32 2>>$OUT/b14.err | python scripts/probes/show_bench.py - --short; done
$B --steps 20 2>>$OUT/b14.err | python scripts/probes/show_bench.py - --short
$B --steps 20 --x-branch-stream off 2>>$OUT/b14.err | python scripts/probes/show_bench.py - --short
