#!/bin/bash
# Library-GEMM selections for the rpn_multiclass step at 1 / 2 / 4 / 8 frames per GPU (PyTorch TunableOp) -> heterofusionrcnn_amd/tuned_gemms.csv
# Run on the GPU box from the repo root; the merged file lands in gpurun_out/r4/tuned_gemms.csv (copy it into the package).
set -o pipefail
OUT=gpurun_out/r4
mkdir -p $OUT
for f in 1 2 4 8; do
  rm -f $OUT/tune_f$f.csv
  timeout -k 10 400 python bench.py --no-op-table --no-cpu-baseline --no-side-runs --frames-per-gpu $f --steps 8 --warmup 2 --gemm-tuning tune:$OUT/tune_f$f.csv 2>>$OUT/tune.err | python scripts/probes/show_bench.py - --short || exit 1
done
python - <<'PY'
import glob
head, rows = [], {}
for f in sorted(glob.glob("gpurun_out/r4/tune_f*.csv")):
    for line in open(f):
        line = line.rstrip("\n")
        if not line:
            continue
        if line.startswith("Validator"):
            if line not in head:
                head.append(line)
        else:
            key = ",".join(line.split(",")[:2])
            rows.setdefault(key, line)
open("gpurun_out/r4/tuned_gemms.csv", "w").write("\n".join(head + list(rows.values())) + "\n")
print("merged", len(rows), "shapes")
PY
