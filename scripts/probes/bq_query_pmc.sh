#!/bin/bash
# SQ / TA counters of the cell-sorted pair at the batched shape (80 clouds); GPU box, repo root
OUT=$PWD/gpurun_out/r4
mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 -L > $OUT/counters_list.txt 2>&1
rm -rf $OUT/bqq_pmc1 $OUT/bqq_pmc2
BQ_SHAPES=1 timeout -k 10 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM --output-format csv -d $OUT/bqq_pmc1 -o p -- python3 scripts/probes/bq_sorted_timing.py > /dev/null 2>&1
BQ_SHAPES=1 timeout -k 10 200 rocprofv3 --pmc TA_TA_BUSY_sum TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE --output-format csv -d $OUT/bqq_pmc2 -o p -- python3 scripts/probes/bq_sorted_timing.py > $OUT/bqq_pmc2.log 2>&1
python3 - <<'PY'
import csv, glob, collections
for d in ("gpurun_out/r4/bqq_pmc1", "gpurun_out/r4/bqq_pmc2"):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        per = collections.defaultdict(float)
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0][:40]
            if "bq_" in k or "qbp_" in k:
                per[(k, r["Counter_Name"], r["Dispatch_Id"])] += float(r["Counter_Value"])
        for (k, c, _), v in per.items():
            agg[k][c].append(v)
    for k in agg:
        print(d.split("/")[-1], k, {c: round(sorted(v)[len(v) // 2]) for c, v in agg[k].items()})
PY
