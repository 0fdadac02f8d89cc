#!/bin/bash
set -o pipefail
OUT=$GRAFT_REPO_ROOT/gpurun_out/r4b; mkdir -p $OUT; cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -m pytest tests -m gpu -x -q -k "bn or dropout or dense or mlp or lift or chain or sa_module or stack" > $OUT/pytest_fin.log 2>&1; tail -3 $OUT/pytest_fin.log
timeout -k 10 200 python scripts/probes/bn_stream_timing.py 2>/dev/null | head -12
B="python bench.py --no-op-table --no-cpu-baseline --no-side-runs"
$B --steps 20 2>>$OUT/b21.err | python scripts/probes/show_bench.py - --short
$B --frames-per-gpu 1 --steps 32 2>>$OUT/b21.err | python scripts/probes/show_bench.py - --short
