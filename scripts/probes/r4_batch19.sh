#!/bin/bash
# eight frames and one frame at HEAD: kernel trace, one step's timeline, per-kernel stats
set -o pipefail
OUT=$GRAFT_REPO_ROOT/gpurun_out/r4b; mkdir -p $OUT; cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for F in 8 1; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof$F -o f$F -- python3 bench.py --frames-per-gpu $F --steps 12 --warmup 3 --no-cpu-baseline --no-op-table --no-side-runs > $OUT/f${F}_prof.log 2>&1; echo "rocprof exit $?"
  T=$(find $OUT/prof$F -name "*kernel_trace.csv" | head -1)
  python scripts/probes/step_timeline.py $T $OUT/f${F}_timeline.txt > $OUT/f${F}_timeline_summary.txt; head -12 $OUT/f${F}_timeline_summary.txt
  cp $(find $OUT/prof$F -name "*kernel_stats.csv" | head -1) $OUT/f${F}_kernel_stats.csv
  rm -rf $OUT/prof$F
done
