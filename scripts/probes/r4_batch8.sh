#!/bin/bash
# single-launch BatchNorm for short tensors: tests, then the one-frame and eight-frame step with the diag build (HF_BN_SMALL_ROWS=0 = off)
set -o pipefail
OUT=$GRAFT_REPO_ROOT/gpurun_out/r4; mkdir -p $OUT; cd $GRAFT_REPO_ROOT
timeout -k 10 400 python -m pytest tests/test_ops_gpu.py -q -m gpu -k "bn or mlp or linear or pointcnn or xconv" > $OUT/t8.log 2>&1; rc=$?; tail -5 $OUT/t8.log
if [ $rc -ne 0 ]; then exit 1; fi
export HFOPS_LIBRARY=$PWD/heterofusionrcnn_amd/csrc/build_diag/libhfops_diag.so
B="python bench.py --no-op-table --no-cpu-baseline --no-side-runs"
run() { name=$1; shift; echo "== $name" >> $OUT/b8.log; timeout -k 10 300 "$@" 2>>$OUT/b8.err | python scripts/probes/show_bench.py - --short >> $OUT/b8.log 2>&1; }
: > $OUT/b8.log
HF_BN_SMALL_ROWS=0 run f1_off $B --frames-per-gpu 1 --steps 32
run f1_on $B --frames-per-gpu 1 --steps 32
HF_BN_SMALL_ROWS=0 run f1_off $B --frames-per-gpu 1 --steps 32
run f1_on $B --frames-per-gpu 1 --steps 32
HF_BN_SMALL_ROWS=2048 run f1_2048 $B --frames-per-gpu 1 --steps 32
HF_BN_SMALL_ROWS=0 run f8_off $B --steps 20
run f8_on $B --steps 20
cat $OUT/b8.log
