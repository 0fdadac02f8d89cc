#!/bin/bash
OUT=$PWD/gpurun_out/r4
mkdir -p $OUT
B="python bench.py --no-op-table --no-cpu-baseline --no-side-runs"
run() { name=$1; shift; echo "== $name: $*" >> $OUT/batch5.log; timeout -k 10 300 "$@" 2>>$OUT/batch5.err | python scripts/probes/show_bench.py - --short >> $OUT/batch5.log 2>&1; }
: > $OUT/batch5.log
HFOPS_LIBRARY=$PWD/build_variants/libhfops_expm1.so run f8_expm1 $B --steps 20
run f8_exp $B --steps 20
HFOPS_LIBRARY=$PWD/build_variants/libhfops_expm1.so run f8_expm1 $B --steps 20
run f8_exp $B --steps 20
HFOPS_LIBRARY=$PWD/build_variants/libhfops_expm1.so run f1_expm1 $B --frames-per-gpu 1 --steps 32
run f1_exp $B --frames-per-gpu 1 --steps 32
cat $OUT/batch5.log
python scripts/bev_nms_timing.py 2>/dev/null | tail -8
