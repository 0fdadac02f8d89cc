#!/bin/bash
# Run on the GPU box (via gpurun) from the repo root: full GPU test suite, smoke, bench lines, rocprof kernel stats of the
# bench, kernel trace + PMC passes of the roofline kernel, its phase stamps, the bev_iou / NMS kernels (trace, phases, PMC).
# Outputs under gpurun_out/evidence/.  Build scripts/probes/libhfops_stamps.so first (scripts/probes/build_stamps.sh).
set -o pipefail
OUT=$GRAFT_REPO_ROOT/gpurun_out/evidence
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 10 500 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; rc=$?
echo "pytest exit $rc" >> $OUT/pytest_gpu.log; tail -3 $OUT/pytest_gpu.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit 1; fi
timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke exit $?"
timeout -k 10 400 python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; echo "bench exit $?"
cat $OUT/bench.json
timeout -k 10 300 python bench.py --steps 10 --warmup 3 --workload rpn_multiclass --no-op-table --no-cpu-baseline > $OUT/bench_rpn_multiclass.json 2>> $OUT/bench.err; echo "bench rpn_multiclass exit $?"
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --workload stack --no-op-table --no-cpu-baseline > $OUT/bench_stack.json 2>> $OUT/bench.err; echo "bench stack exit $?"
cat $OUT/bench_rpn_multiclass.json $OUT/bench_stack.json | cut -c1-400
if [ -f scripts/probes/libhfops_stamps.so ]; then timeout -k 10 100 python scripts/probes/qbp_stamps.py > $OUT/qbp_cell_phase_stamps.txt 2>&1; cat $OUT/qbp_cell_phase_stamps.txt; fi
timeout -k 10 100 python scripts/qbp_ab.py > $OUT/qbp_ab.json 2>&1; echo "qbp_ab exit $?"
timeout -k 10 100 python scripts/bev_nms_timing.py > $OUT/bev_nms_timing.json 2>&1; echo "bev_nms_timing exit $?"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_bench -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-op-table > $OUT/bench_prof.log 2>&1; echo "rocprof bench exit $?"
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_roofline -- python3 $GRAFT_REPO_ROOT/scripts/roofline_kernel.py > $OUT/roofline_trace.log 2>&1; echo "rocprof roofline exit $?"
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $GRAFT_REPO_ROOT/scripts/roofline_kernel.py > $OUT/pmc_fetch.log 2>&1 && \
timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $GRAFT_REPO_ROOT/scripts/roofline_kernel.py > $OUT/pmc_write.log 2>&1; echo "pmc exit $?"
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_bev_nms -- python3 $GRAFT_REPO_ROOT/scripts/bev_nms_kernels.py > $OUT/bev_nms_trace.log 2>&1; echo "rocprof bev/nms exit $?"
PHASES=1 timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $OUT/prof_bev_phases -- python3 $GRAFT_REPO_ROOT/scripts/bev_nms_timing.py > $OUT/bev_phases_trace.log 2>&1; echo "rocprof bev phases exit $?"
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_bev_fetch -- python3 $GRAFT_REPO_ROOT/scripts/bev_nms_kernels.py > $OUT/pmc_bev_fetch.log 2>&1 && \
timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_bev_write -- python3 $GRAFT_REPO_ROOT/scripts/bev_nms_kernels.py > $OUT/pmc_bev_write.log 2>&1; echo "pmc bev/nms exit $?"
cd $GRAFT_REPO_ROOT
python scripts/parse_pmc.py $OUT/pmc_fetch qbp_cell; python scripts/parse_pmc.py $OUT/pmc_write qbp_cell
python scripts/make_traffic_json.py $OUT/pmc_fetch $OUT/pmc_write $OUT/roofline_traffic.json $OUT/roofline
python scripts/parse_trace.py $OUT/prof_roofline qbp_cell > $OUT/roofline_kernel_durations.txt; cat $OUT/roofline_kernel_durations.txt
python scripts/probes/bev_parse_phases.py $OUT/prof_bev_phases > $OUT/bev_iou_phases.txt; cat $OUT/bev_iou_phases.txt
(python scripts/parse_pmc.py $OUT/pmc_bev_fetch bev_iou nms_; python scripts/parse_pmc.py $OUT/pmc_bev_write bev_iou nms_) > $OUT/bev_nms_pmc.txt; cat $OUT/bev_nms_pmc.txt
