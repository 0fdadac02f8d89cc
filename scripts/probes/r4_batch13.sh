#!/bin/bash
# second BatchNorm pass walking the tensor from its end: product library (reverse) against a diag build with -DHF_BN_FORWARD_ORDER
set -o pipefail
OUT=$GRAFT_REPO_ROOT/gpurun_out/r4; mkdir -p $OUT; cd $GRAFT_REPO_ROOT
timeout -k 10 400 python -m pytest tests/test_ops_gpu.py -q -m gpu -k "bn or mlp or linear or pointcnn or xconv or sa_ or shared" > $OUT/t13b.log 2>&1; rc=$?; tail -3 $OUT/t13b.log
if [ $rc -ne 0 ]; then exit 1; fi
D=$PWD/heterofusionrcnn_amd/csrc/build_diag/libhfops_diag.so
B="python bench.py --no-op-table --no-cpu-baseline --no-side-runs"
run() { name=$1; shift; echo "== $name" >> $OUT/b13.log; timeout -k 10 300 "$@" 2>>$OUT/b13.err | python scripts/probes/show_bench.py - --short >> $OUT/b13.log 2>&1; }
: > $OUT/b13.log
HFOPS_LIBRARY=$D run f8_forward $B --steps 20
run f8_reverse $B --steps 20
HFOPS_LIBRARY=$D run f8_forward $B --steps 20
run f8_reverse $B --steps 20
HFOPS_LIBRARY=$D run f1_forward $B --frames-per-gpu 1 --steps 32
run f1_reverse $B --frames-per-gpu 1 --steps 32
cat $OUT/b13.log
HFOPS_LIBRARY=$D timeout -k 10 200 python scripts/probes/abi_call_shapes.py 8 bn_relu_bwd 2>&1 | grep "hf_bn_relu_bwd  " | head -8
timeout -k 10 200 python scripts/probes/abi_call_shapes.py 8 bn_relu_bwd 2>&1 | grep "hf_bn_relu_bwd  " | head -8
