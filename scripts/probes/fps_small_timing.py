"""Diagnostic: FPS on many small clouds (the second stage's RoI clouds: 800 x 512 -> 128, 128 -> 32, 32 -> 8), workgroup kernel
against the one-wave-per-cloud kernel; device time per call."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import heterofusionrcnn_amd as hf
from bench import time_op
rng = np.random.default_rng(0)
for (b, n, m) in ((800, 512, 128), (800, 128, 32), (800, 32, 8), (8, 512, 128), (8, 256, 64), (8, 64, 16)):
    x = torch.from_numpy((rng.normal(0, 1, (b, n, 3)) * np.array([1.5, 0.8, 2.5])).astype(np.float32)).cuda()
    tp = time_op(lambda: hf.farthest_point_sample(m, x, kernel="plain"), iters=10, warm=2)
    tw = time_op(lambda: hf.farthest_point_sample(m, x, kernel="wave"), iters=10, warm=2)
    same = torch.equal(hf.farthest_point_sample(m, x, kernel="plain"), hf.farthest_point_sample(m, x, kernel="wave"))
    print("b %4d n %4d m %4d: workgroup kernel %7.1f us, one wave per cloud %7.1f us, equal %s" % (b, n, m, tp, tw, same), flush=True)
