#!/bin/bash
# streaming BatchNorm passes: previous build against this one, then the affected tests and the step
set -o pipefail
OUT=$GRAFT_REPO_ROOT/gpurun_out/r4b; mkdir -p $OUT; cd $GRAFT_REPO_ROOT
HFOPS_LIBRARY=$GRAFT_REPO_ROOT/heterofusionrcnn_amd/csrc/build_prev/libhfops_prev.so timeout -k 10 200 python scripts/probes/bn_stream_timing.py 2>/dev/null | tee $OUT/bn_stream_prev.txt
timeout -k 10 200 python scripts/probes/bn_stream_timing.py 2>/dev/null | tee $OUT/bn_stream_new.txt
timeout -k 10 400 python -m pytest tests -m gpu -x -q -k "bn or elu or dense or pointcnn or graph or rpn or lift or xconv" > $OUT/pytest_bn.log 2>&1; tail -3 $OUT/pytest_bn.log
B="python bench.py --no-op-table --no-cpu-baseline --no-side-runs"
$B --steps 20 2>>$OUT/b18.err | python scripts/probes/show_bench.py - --short
$B --frames-per-gpu 1 --steps 32 2>>$OUT/b18.err | python scripts/probes/show_bench.py - --short
