#!/bin/bash
set -o pipefail
OUT=$GRAFT_REPO_ROOT/gpurun_out/r4; mkdir -p $OUT; cd $GRAFT_REPO_ROOT
timeout -k 10 300 python scripts/probes/abi_call_shapes.py 1 > $OUT/abi_shapes_1frame.txt 2>&1; echo "exit $?"
head -60 $OUT/abi_shapes_1frame.txt
