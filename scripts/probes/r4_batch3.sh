#!/bin/bash
# round-4 measurement batch 3 (GPU box, repo root)
OUT=$PWD/gpurun_out/r4
mkdir -p $OUT
export TMPDIR=/tmp
B="python bench.py --no-op-table --no-cpu-baseline --no-side-runs"
echo "== two-stage, old xconv stores / new" > $OUT/batch3.log
HFOPS_LIBRARY=$PWD/build_variants/libhfops_xconv_old.so python scripts/two_stage_profile.py 2>&1 | tail -1 >> $OUT/batch3.log
python scripts/two_stage_profile.py 2>&1 | tail -1 >> $OUT/batch3.log
HFOPS_LIBRARY=$PWD/build_variants/libhfops_xconv_old.so python scripts/two_stage_profile.py 2>&1 | tail -1 >> $OUT/batch3.log
python scripts/two_stage_profile.py 2>&1 | tail -1 >> $OUT/batch3.log
echo "== with vgg" >> $OUT/batch3.log
timeout -k 10 300 $B --with-vgg --steps 6 2>$OUT/vgg.err | python scripts/probes/show_bench.py - --short >> $OUT/batch3.log 2>&1
echo "== 2 ranks rehearsed on one GPU (strong default + weak extra)" >> $OUT/batch3.log
timeout -k 10 500 python bench.py --gpus 2 --rehearse-on-one-gpu --steps 6 --warmup 2 --no-op-table --no-cpu-baseline --no-side-runs 2>$OUT/rehearse.err > $OUT/rehearse.json; python scripts/probes/show_bench.py $OUT/rehearse.json --short >> $OUT/batch3.log 2>&1
cat $OUT/batch3.log
# kernel trace of the one-frame step
rm -rf $OUT/prof_f1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_f1 -o f1 -- python3 bench.py --frames-per-gpu 1 --steps 16 --warmup 3 --no-cpu-baseline --no-op-table --no-side-runs > $OUT/prof_f1.log 2>&1; echo "rocprof f1 exit $?"
ls $OUT/prof_f1/*/ | head
# bev_iou ALU fraction
rm -rf $OUT/bev_pmc $OUT/bev_tr
timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/bev_pmc -o p -- python3 scripts/bev_nms_kernels.py > /dev/null 2>&1
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $OUT/bev_tr -o t -- python3 scripts/bev_nms_kernels.py > /dev/null 2>&1
python3 scripts/make_alu_json.py $OUT/bev_pmc $OUT/bev_tr $OUT/bev_iou_alu.json
bash scripts/probes/bq_query_pmc.sh
