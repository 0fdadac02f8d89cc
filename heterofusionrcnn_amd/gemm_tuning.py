"""Library-GEMM selection for the train step's shapes.

The plain GEMMs of the step (pointwise convolutions and dense layers wider than the hand-written MFMA kernels cover, the split-K
weight gradients) go to rocBLAS / hipBLASLt through torch.mm / addmm / bmm.  The library's default heuristic picks 256 x 256
macro-tiles for most of them, which leaves the chip half empty at one frame per GPU (2 048 .. 16 384 rows): PyTorch's TunableOp
times the library's candidate solutions per shape and keeps the fastest.  `tuned_gemms.csv` next to this file holds the
selections for the rpn_multiclass step at 1 / 2 / 4 / 8 frames per GPU on MI355X (made by scripts/tune_gemms.sh on the GPU box);
TunableOp validates the library versions recorded in the file and silently falls back to the default heuristic on a mismatch or
for a shape the file does not hold, so loading it can only change WHICH library kernel runs, never the arithmetic (fp32 GEMM
either way; summation order inside a GEMM is the library's in both cases).

  enable()                  load the shipped selections, tuning off (what bench.py / graph_step users call before the first step)
  enable(tune=True, path=)  time candidates for every new shape and record them in `path` (the tuning run)
"""
import os

import torch

SHIPPED = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tuned_gemms.csv")


def enable(path=None, tune=False, max_tuning_ms=30, max_iterations=30):
    """returns the file in use, or None when there is nothing to load (no shipped file and tune=False)"""
    import torch.cuda.tunable as tunable
    path = path or SHIPPED
    if not tune and not os.path.exists(path):
        return None
    tunable.enable(True)
    tunable.set_filename(path, insert_device_ordinal=False)
    tunable.tuning_enable(bool(tune))
    if tune:
        tunable.set_max_tuning_duration(int(max_tuning_ms))
        tunable.set_max_tuning_iterations(int(max_iterations))
    return path


def disable():
    import torch.cuda.tunable as tunable
    tunable.enable(False)
