#!/bin/bash
# copy the summaries of gpurun_out/evidence4/ (scratch) into profiles/ (tracked) under their round-4 names
E=gpurun_out/evidence4; P=profiles
cp $E/bench.json $P/r04_bench.json
cp $E/pytest_gpu.log $P/r04_pytest_gpu.log
cp $E/smoke.log $P/r04_smoke.log
cp $E/prof_bench/bench_kernel_stats.csv $P/r04_bench_kernel_stats.csv
cp $E/prof_roofline/roofline_kernel_stats.csv $P/r04_roofline_kernel_stats.csv
cp $E/prof_two_stage/two_stage_kernel_stats.csv $P/r04_two_stage_kernel_stats.csv
cp $E/roofline_kernel_durations.txt $P/r04_roofline_kernel_durations.txt
cp $E/roofline_pmc_FETCH_SIZE.csv $P/r04_roofline_pmc_FETCH_SIZE.csv 2>/dev/null
cp $E/roofline_pmc_WRITE_SIZE.csv $P/r04_roofline_pmc_WRITE_SIZE.csv 2>/dev/null
cp $E/roofline_traffic.json $P/roofline_traffic.json
cp $E/bq_sorted_timing.txt $P/r04_bq_sorted_timing.txt
cp $E/bev_nms_timing.json $P/r04_bev_nms_timing.json
[ -f $E/bev_iou_alu.json ] && cp $E/bev_iou_alu.json $P/bev_iou_alu.json
[ -f $E/prof_f1/f1_kernel_stats.csv ] && cp $E/prof_f1/f1_kernel_stats.csv $P/r04_bench_1frame_kernel_stats.csv
[ -f $E/step_breakdown.txt ] && cp $E/step_breakdown.txt $P/r04_step_breakdown.txt
[ -f $E/frames_per_gpu_4_2.txt ] && cp $E/frames_per_gpu_4_2.txt $P/r04_frames_per_gpu_4_2.txt
[ -f $E/bn_stream_timing.txt ] && cp $E/bn_stream_timing.txt $P/r04_bn_stream_timing.txt
echo published
