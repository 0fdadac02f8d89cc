#!/bin/bash
# Where the cell ball-query kernel's time goes at the headline shape (B=8): cumulative phase exits (HF_QBP_STOP, diagnostic
# build: outputs invalid) under rocprofv3 -- one kernel-trace pass (durations) and one --pmc pass (SQ counters) per exit.
# stop: -1 empty kernel | -2 after the marking | 1 after the bitmap pass + candidate lists | 2 same + barrier | 4 after the search
#        | 0 whole kernel.  Run on the GPU box from the repo root; output gpurun_out/r4/qbp_phase_pmc.txt
set -o pipefail
OUT=$PWD/gpurun_out/r4
mkdir -p $OUT
export TMPDIR=/tmp
export HFOPS_LIBRARY=$PWD/heterofusionrcnn_amd/csrc/build_diag/libhfops_diag.so
CTRS="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS"
: > $OUT/qbp_phase_pmc.txt
for stop in -1 -2 1 4 0; do
  rm -rf $OUT/qp_tr $OUT/qp_pmc
  HF_QBP_STOP=$stop timeout -k 10 150 rocprofv3 --kernel-trace --output-format csv -d $OUT/qp_tr -o t -- python3 scripts/roofline_kernel.py > /dev/null 2>&1 || exit 1
  HF_QBP_STOP=$stop timeout -k 10 150 rocprofv3 --pmc $CTRS --output-format csv -d $OUT/qp_pmc -o p -- python3 scripts/roofline_kernel.py > /dev/null 2>&1 || exit 1
  python3 scripts/probes/qbp_phase_pmc_parse.py $stop $OUT/qp_tr $OUT/qp_pmc >> $OUT/qbp_phase_pmc.txt
done
cat $OUT/qbp_phase_pmc.txt
