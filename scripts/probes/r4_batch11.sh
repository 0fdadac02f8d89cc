#!/bin/bash
# HIP runtime knobs against the replayed one-frame step (product library)
set -o pipefail
OUT=$GRAFT_REPO_ROOT/gpurun_out/r4; mkdir -p $OUT; cd $GRAFT_REPO_ROOT
B="python bench.py --no-op-table --no-cpu-baseline --no-side-runs --frames-per-gpu 1 --steps 32"
run() { name=$1; shift; echo "== $name" >> $OUT/b11.log; timeout -k 10 200 env "$@" $B 2>>$OUT/b11.err | python scripts/probes/show_bench.py - --short >> $OUT/b11.log 2>&1; }
: > $OUT/b11.log
run base X=1
run packet_capture1 DEBUG_CLR_GRAPH_PACKET_CAPTURE=1
run packet_capture0 DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
run opt_flush0 AMD_OPT_FLUSH=0
run sys_scope0 ROC_SYSTEM_SCOPE_SIGNAL=0
run graph_queues1 DEBUG_HIP_FORCE_GRAPH_QUEUES=1
run graph_queues4 DEBUG_HIP_FORCE_GRAPH_QUEUES=4
run graph_batch DEBUG_HIP_GRAPH_BATCH_SIZE=1024
run dyn_queues0 DEBUG_HIP_DYNAMIC_QUEUES=0
run kernarg_opt DEBUG_HIP_KERNARG_COPY_OPT=1
run base X=1
cat $OUT/b11.log
