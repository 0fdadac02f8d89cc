#!/bin/bash
# Round-3 evidence set; run on the GPU box (via gpurun) from the repo root.  Outputs under gpurun_out/evidence3/ (copied into
# profiles/ as r03_* afterwards).  Needs heterofusionrcnn_amd/csrc/build_diag/libhfops_diag.so (scripts/probes/build_diag.sh)
# for the two steps that use diagnostic knobs; everything else runs on the product library.
set -o pipefail
OUT=$GRAFT_REPO_ROOT/gpurun_out/evidence3
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 10 700 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; rc=$?
echo "pytest exit $rc" >> $OUT/pytest_gpu.log; tail -3 $OUT/pytest_gpu.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit 1; fi
timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke exit $?"; tail -1 $OUT/smoke.log
timeout -k 10 900 python bench.py --steps 20 --warmup 3 > $OUT/bench.json 2> $OUT/bench.err; echo "bench exit $?"
python scripts/probes/show_bench.py $OUT/bench.json | cut -c1-300
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_bench -o bench -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-op-table --no-side-runs > $OUT/bench_prof.log 2>&1; echo "rocprof bench exit $?"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_bench_b1 -o bench_b1 -- python3 bench.py --frames-per-gpu 1 --steps 10 --warmup 3 --no-cpu-baseline --no-op-table --no-side-runs > $OUT/bench_b1_prof.log 2>&1; echo "rocprof bench B=1 exit $?"
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_roofline -o roofline -- python3 scripts/roofline_kernel.py > $OUT/roofline_trace.log 2>&1; echo "rocprof roofline exit $?"
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 scripts/roofline_kernel.py > $OUT/pmc_fetch.log 2>&1 && \
timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 scripts/roofline_kernel.py > $OUT/pmc_write.log 2>&1; echo "pmc exit $?"
python scripts/make_traffic_json.py $OUT/pmc_fetch $OUT/pmc_write $OUT/roofline_traffic.json $OUT/roofline
python scripts/parse_trace.py $OUT/prof_roofline qbp_cell > $OUT/roofline_kernel_durations.txt; cat $OUT/roofline_kernel_durations.txt
BQ_SHAPES=0,1,2,3,7,4,5 timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $OUT/prof_bq -o bq -- python3 scripts/probes/bq_sorted_timing.py > $OUT/bq_sorted_timing.txt 2>&1; echo "bq timing exit $?"
python scripts/probes/kernel_durations.py $OUT/prof_bq >> $OUT/bq_sorted_timing.txt; grep "^[0-9(]" $OUT/bq_sorted_timing.txt
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_two_stage -o two_stage -- python3 scripts/two_stage_profile.py > $OUT/two_stage.log 2>&1; echo "rocprof two-stage exit $?"; tail -1 $OUT/two_stage.log
HFOPS_LIBRARY=$GRAFT_REPO_ROOT/heterofusionrcnn_amd/csrc/build_diag/libhfops_diag.so timeout -k 10 200 python scripts/bev_nms_timing.py > $OUT/bev_nms_timing.json 2>&1; echo "bev_nms_timing exit $?"
(for st in 1 2 3 4 0; do for sp in 2 4; do echo "== HF_BQ_STOP=$st HF_BQ_SPLIT=$sp (cumulative phase exits of bq_build_kernel, 80 / 8 clouds)"; HFOPS_LIBRARY=$GRAFT_REPO_ROOT/heterofusionrcnn_amd/csrc/build_diag/libhfops_diag.so HF_BQ_STOP=$st HF_BQ_SPLIT=$sp BQ_SHAPES=0,1 timeout -k 10 100 rocprofv3 --kernel-trace --output-format csv -d $OUT/prof_bqst -o bq -- python3 scripts/probes/bq_sorted_timing.py > /dev/null 2>&1; python scripts/probes/kernel_durations.py $OUT/prof_bqst | grep bq_build; done; done) > $OUT/bq_build_phases.txt 2>&1; tail -4 $OUT/bq_build_phases.txt
echo done
