"""Diagnostic: knn_point at the RCNN's shapes (800 RoI clouds of 512 / 128 / 32 points, rcnn_multiclass.config:157-186) and at the
RPN's, grid ring search against the tiled all-pairs kernel; device time per call."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import heterofusionrcnn_amd as hf
from bench import time_op
rng = np.random.default_rng(0)
for (b, n, m, k) in ((800, 512, 512, 4), (800, 512, 128, 8), (800, 128, 32, 12), (800, 32, 8, 12), (8, 16384, 16384, 8), (8, 16384, 4096, 8), (8, 4096, 1024, 8)):
    x1 = torch.from_numpy((rng.normal(0, 1, (b, n, 3)) * np.array([1.5, 0.8, 2.5])).astype(np.float32)).cuda()
    x2 = x1[:, :m].contiguous()
    tg = time_op(lambda: hf.knn_point(k, x1, x2), iters=10, warm=2)
    ta = time_op(lambda: hf.knn_point(k, x1, x2, all_pairs=True), iters=5, warm=1)
    vg, ig = hf.knn_point(k, x1, x2); va, ia = hf.knn_point(k, x1, x2, all_pairs=True)
    print("b %4d n %6d m %6d k %2d: grid %8.1f us, all pairs %8.1f us, equal %s" % (b, n, m, k, tg, ta, bool(torch.equal(ig, ia))), flush=True)
