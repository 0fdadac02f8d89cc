#!/bin/bash
# FETCH_SIZE / WRITE_SIZE (separate passes) of one SA level, in-place against materialised first layer; GPU box, repo root
OUT=$PWD/gpurun_out/r4
mkdir -p $OUT
export TMPDIR=/tmp
: > $OUT/sa_gather_traffic.txt
for level in 1 2; do
for mode in materialised inplace; do
  python3 scripts/probes/sa_gather_traffic.py $mode $level >> $OUT/sa_gather_traffic.txt 2>/dev/null
  for ctr in FETCH_SIZE WRITE_SIZE; do
    rm -rf $OUT/sa_pmc
    timeout -k 10 200 rocprofv3 --pmc $ctr --output-format csv -d $OUT/sa_pmc -o p -- python3 scripts/probes/sa_gather_traffic.py $mode $level > /dev/null 2>&1
    python3 - "$mode" "$level" "$ctr" >> $OUT/sa_gather_traffic.txt <<'PY'
import csv, glob, sys
mode, level, ctr = sys.argv[1:4]
tot = 0.0
for f in glob.glob("gpurun_out/r4/sa_pmc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == ctr:
            tot += float(r["Counter_Value"])
# 13 iterations of forward + backward in the script (3 warm-up + 10 timed); the geometry ops and the set-up run once
scale = 2.0 if ctr == "FETCH_SIZE" else 1.0      # gfx950: FETCH_SIZE counts 128-byte requests as 64 (MI355X_MICROARCH.md)
print("%-12s level %s  %-10s %8.1f MB per forward+backward (all kernels of the process / 13)" % (mode, level, ctr, tot * scale * 1024 / 13 / 1e6))
PY
  done
done
done
cat $OUT/sa_gather_traffic.txt
