#!/bin/bash
# one-frame-per-GPU (and 8-frame) train step under the round-4 switches; run on the GPU box from the repo root
OUT=gpurun_out/r4
mkdir -p $OUT
B="python bench.py --no-op-table --no-cpu-baseline --no-side-runs"
run() { name=$1; shift; echo "== $name: $*" >> $OUT/step_variants.log; timeout -k 10 300 "$@" 2>>$OUT/step_variants.err | python scripts/probes/show_bench.py - --short >> $OUT/step_variants.log 2>&1; }
: > $OUT/step_variants.log
run f1_torch_off   $B --frames-per-gpu 1 --steps 32 --optimizer torch --x-branch-stream off
run f1_hf_off      $B --frames-per-gpu 1 --steps 32 --optimizer hf --x-branch-stream off
run f1_hf_on       $B --frames-per-gpu 1 --steps 32 --optimizer hf --x-branch-stream on
run f8_hf_off      $B --steps 20 --optimizer hf --x-branch-stream off
run f8_hf_on       $B --steps 20 --optimizer hf --x-branch-stream on
export PYTORCH_TUNABLEOP_ENABLED=1 PYTORCH_TUNABLEOP_FILENAME=$OUT/tunableop_f1.csv PYTORCH_TUNABLEOP_MAX_TUNING_DURATION_MS=20 PYTORCH_TUNABLEOP_MAX_WARMUP_DURATION_MS=5 PYTORCH_TUNABLEOP_VERBOSE=0
run f1_hf_on_tunable $B --frames-per-gpu 1 --steps 32 --optimizer hf --x-branch-stream on
cat $OUT/step_variants.log
