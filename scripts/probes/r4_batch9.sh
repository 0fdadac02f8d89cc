#!/bin/bash
# one-frame step under rocprofv3 (kernel trace): the trace of the replayed steps is kept for scripts/probes/step_timeline.py
set -o pipefail
OUT=$GRAFT_REPO_ROOT/gpurun_out/r4; mkdir -p $OUT; cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_f1 -o f1 -- python3 bench.py --frames-per-gpu ${FRAMES:-1} --steps 12 --warmup 3 --no-cpu-baseline --no-op-table --no-side-runs > $OUT/f1_prof.log 2>&1; echo "exit $?"
find $OUT/prof_f1 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/f1_kernel_stats.csv
find $OUT/prof_f1 -name "*kernel_trace.csv" | head -1 | xargs -I{} cp {} $OUT/f1_kernel_trace.csv
rm -rf $OUT/prof_f1
python scripts/probes/step_timeline.py $OUT/f1_kernel_trace.csv $OUT/f1_step.txt | head -30
