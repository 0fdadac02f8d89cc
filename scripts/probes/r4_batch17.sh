#!/bin/bash
# one frame per GPU at HEAD: kernel trace of the replayed step, timeline of one step, per-kernel stats
set -o pipefail
OUT=$GRAFT_REPO_ROOT/gpurun_out/r4b; mkdir -p $OUT; cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof1 -o f1 -- python3 bench.py --frames-per-gpu 1 --steps 24 --warmup 3 --no-cpu-baseline --no-op-table --no-side-runs > $OUT/f1_prof.log 2>&1; echo "rocprof exit $?"
tail -1 $OUT/f1_prof.log | python scripts/probes/show_bench.py - --short
T=$(find $OUT/prof1 -name "*kernel_trace.csv" | head -1)
python scripts/probes/step_timeline.py $T $OUT/f1_timeline.txt | tee $OUT/f1_timeline_summary.txt
cp $(find $OUT/prof1 -name "*kernel_stats.csv" | head -1) $OUT/f1_kernel_stats.csv
rm -rf $OUT/prof1
