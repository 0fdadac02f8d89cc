#!/bin/bash
# BatchNorm geometry (>= 8 rows per thread; wave-per-channel finalize): tests, then A/B with the diag build
set -o pipefail
OUT=$GRAFT_REPO_ROOT/gpurun_out/r4; mkdir -p $OUT; cd $GRAFT_REPO_ROOT
timeout -k 10 400 python -m pytest tests/test_ops_gpu.py -q -m gpu -k "bn or mlp or linear or pointcnn or xconv or sa_ or shared" > $OUT/t12.log 2>&1; rc=$?; tail -3 $OUT/t12.log
if [ $rc -ne 0 ]; then exit 1; fi
export HFOPS_LIBRARY=$PWD/heterofusionrcnn_amd/csrc/build_diag/libhfops_diag.so
B="python bench.py --no-op-table --no-cpu-baseline --no-side-runs"
run() { name=$1; shift; echo "== $name" >> $OUT/b12.log; timeout -k 10 300 "$@" 2>>$OUT/b12.err | python scripts/probes/show_bench.py - --short >> $OUT/b12.log 2>&1; }
: > $OUT/b12.log
HF_BN_ROWS_PER_THREAD=1 run f1_rpt1 $B --frames-per-gpu 1 --steps 32
run f1_rpt8 $B --frames-per-gpu 1 --steps 32
HF_BN_ROWS_PER_THREAD=4 run f1_rpt4 $B --frames-per-gpu 1 --steps 32
HF_BN_ROWS_PER_THREAD=16 run f1_rpt16 $B --frames-per-gpu 1 --steps 32
HF_BN_ROWS_PER_THREAD=1 run f1_rpt1 $B --frames-per-gpu 1 --steps 32
run f1_rpt8 $B --frames-per-gpu 1 --steps 32
HF_BN_ROWS_PER_THREAD=1 run f8_rpt1 $B --steps 20
run f8_rpt8 $B --steps 20
HF_BN_ROWS_PER_THREAD=16 run f8_rpt16 $B --steps 20
cat $OUT/b12.log
