#!/bin/bash
# Round-4 evidence set; run on the GPU box (via gpurun) from the repo root.  Outputs under gpurun_out/evidence4/ (copied into
# profiles/ as r04_* afterwards).
set -o pipefail
OUT=$GRAFT_REPO_ROOT/gpurun_out/evidence4
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 10 500 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; rc=$?
echo "pytest exit $rc" >> $OUT/pytest_gpu.log; tail -3 $OUT/pytest_gpu.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit 1; fi
timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke exit $?"; tail -1 $OUT/smoke.log
timeout -k 10 1000 python bench.py --steps 20 --warmup 3 > $OUT/bench.json 2> $OUT/bench.err; echo "bench exit $?"
python scripts/probes/show_bench.py $OUT/bench.json | cut -c1-400
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_bench -o bench -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-op-table --no-side-runs > $OUT/bench_prof.log 2>&1; echo "rocprof bench exit $?"
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_roofline -o roofline -- python3 scripts/roofline_kernel.py > $OUT/roofline_trace.log 2>&1; echo "rocprof roofline exit $?"
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 scripts/roofline_kernel.py > $OUT/pmc_fetch.log 2>&1 && \
timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 scripts/roofline_kernel.py > $OUT/pmc_write.log 2>&1; echo "pmc exit $?"
python scripts/make_traffic_json.py $OUT/pmc_fetch $OUT/pmc_write $OUT/roofline_traffic.json $OUT/roofline
python scripts/parse_trace.py $OUT/prof_roofline qbp_cell > $OUT/roofline_kernel_durations.txt; cat $OUT/roofline_kernel_durations.txt
BQ_SHAPES=0,1,2,3,7,4,5 timeout -k 10 200 python3 scripts/probes/bq_sorted_timing.py 2>/dev/null | grep "^[0-9]" > $OUT/bq_sorted_timing.txt; cat $OUT/bq_sorted_timing.txt | cut -c1-250
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_two_stage -o two_stage -- python3 scripts/two_stage_profile.py > $OUT/two_stage.log 2>&1; echo "rocprof two-stage exit $?"; tail -1 $OUT/two_stage.log
timeout -k 10 200 python scripts/bev_nms_timing.py > $OUT/bev_nms_timing.json 2>&1; echo "bev_nms_timing exit $?"; tail -3 $OUT/bev_nms_timing.json
# bev_iou ALU-bound fraction (stamped with the hash of bev_iou.hip; bench.py prints it only while the stamp matches)
rm -rf $OUT/bev_pmc $OUT/bev_tr
timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/bev_pmc -o p -- python3 scripts/bev_nms_kernels.py > /dev/null 2>&1 && \
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $OUT/bev_tr -o t -- python3 scripts/bev_nms_kernels.py > /dev/null 2>&1; echo "bev alu passes exit $?"
python3 scripts/make_alu_json.py $OUT/bev_pmc $OUT/bev_tr $OUT/bev_iou_alu.json
# one frame per GPU: kernel stats of the replayed step
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_f1 -o f1 -- python3 bench.py --frames-per-gpu 1 --steps 16 --warmup 3 --no-cpu-baseline --no-op-table --no-side-runs > $OUT/prof_f1.log 2>&1; echo "rocprof f1 exit $?"
python scripts/probes/step_breakdown.py $(find $OUT/prof_bench -name "*kernel_trace.csv" | head -1) > $OUT/step_breakdown.txt 2>&1; head -3 $OUT/step_breakdown.txt
rm -rf $OUT/bev_pmc/*/*agent_info.csv
# the per-rank shapes of a global batch of 8 on 2 / 4 GPUs (strong scaling, BASELINE config 4), replayed
for F in 4 2; do python bench.py --no-op-table --no-cpu-baseline --no-side-runs --frames-per-gpu $F --steps 32 2>/dev/null | python scripts/probes/show_bench.py - --short; done > $OUT/frames_per_gpu_4_2.txt; cat $OUT/frames_per_gpu_4_2.txt
timeout -k 10 200 python scripts/probes/bn_stream_timing.py 2>/dev/null > $OUT/bn_stream_timing.txt
echo done
