#!/bin/bash
# round-4 measurement batch 2 (GPU box, repo root)
OUT=gpurun_out/r4
mkdir -p $OUT
B="python bench.py --no-op-table --no-cpu-baseline --no-side-runs --gemm-tuning off"
run() { name=$1; shift; echo "== $name: $*" >> $OUT/batch2.log; timeout -k 10 300 "$@" 2>>$OUT/batch2.err | python scripts/probes/show_bench.py - --short >> $OUT/batch2.log 2>&1; }
: > $OUT/batch2.log
HFOPS_LIBRARY=$PWD/build_variants/libhfops_nofold.so run f1_nofold $B --frames-per-gpu 1 --steps 32
run f1_fold $B --frames-per-gpu 1 --steps 32
HFOPS_LIBRARY=$PWD/build_variants/libhfops_nofold.so run f1_nofold_again $B --frames-per-gpu 1 --steps 32
run f1_fold_again $B --frames-per-gpu 1 --steps 32
HFOPS_LIBRARY=$PWD/build_variants/libhfops_nofold.so run f8_nofold $B --steps 20
run f8_fold $B --steps 20
cat $OUT/batch2.log
python scripts/probes/gemm_shapes.py rpn_multiclass 1 > $OUT/gemm_shapes_f1.txt 2>&1; head -45 $OUT/gemm_shapes_f1.txt
python scripts/probes/two_stage_gemm_shapes.py > $OUT/two_stage_gemm_shapes.txt 2>&1; head -12 $OUT/two_stage_gemm_shapes.txt
python scripts/two_stage_profile.py 2>&1 | tail -1
bash scripts/tune_gemms.sh
