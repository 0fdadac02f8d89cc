#!/bin/bash
OUT=$PWD/gpurun_out/r4
mkdir -p $OUT
B="python bench.py --no-op-table --no-cpu-baseline --no-side-runs"
run() { name=$1; shift; echo "== $name: $*" >> $OUT/batch6.log; timeout -k 10 300 "$@" 2>>$OUT/batch6.err | python scripts/probes/show_bench.py - --short >> $OUT/batch6.log 2>&1; }
: > $OUT/batch6.log
run f1_base $B --frames-per-gpu 1 --steps 32
HIP_FORCE_DEV_KERNARG=1 run f1_devkernarg $B --frames-per-gpu 1 --steps 32
run f1_base $B --frames-per-gpu 1 --steps 32
HIP_FORCE_DEV_KERNARG=1 run f1_devkernarg $B --frames-per-gpu 1 --steps 32
run f8_base $B --steps 20
HIP_FORCE_DEV_KERNARG=1 run f8_devkernarg $B --steps 20
GPU_MAX_HW_QUEUES=8 run f1_hwq8 $B --frames-per-gpu 1 --steps 32
GPU_MAX_HW_QUEUES=2 run f1_hwq2 $B --frames-per-gpu 1 --steps 32
cat $OUT/batch6.log
