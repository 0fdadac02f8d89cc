#!/bin/bash
OUT=$PWD/gpurun_out/r4
mkdir -p $OUT
B="python bench.py --no-op-table --no-cpu-baseline --no-side-runs"
run() { name=$1; shift; echo "== $name: $*" >> $OUT/batch4.log; timeout -k 10 300 "$@" 2>>$OUT/batch4.err | python scripts/probes/show_bench.py - --short >> $OUT/batch4.log 2>&1; }
: > $OUT/batch4.log
run f1_wgrad_off $B --frames-per-gpu 1 --steps 32 --wgrad-stream off
run f1_wgrad_on  $B --frames-per-gpu 1 --steps 32 --wgrad-stream on
run f1_wgrad_off $B --frames-per-gpu 1 --steps 32 --wgrad-stream off
run f1_wgrad_on  $B --frames-per-gpu 1 --steps 32 --wgrad-stream on
run f8_wgrad_off $B --steps 20 --wgrad-stream off
run f8_wgrad_on  $B --steps 20 --wgrad-stream on
run f8_graph_on  $B --steps 20 --wgrad-stream on --graph
run f2 $B --frames-per-gpu 2 --steps 24
run f4 $B --frames-per-gpu 4 --steps 24
run stack $B --workload stack --steps 40
run rpn $B --workload rpn --steps 16
cat $OUT/batch4.log
