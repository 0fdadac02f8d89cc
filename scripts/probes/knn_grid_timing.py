"""Diagnostic: the grid kNN / three_nn at the bench shapes (kitti_uniform clouds); device time per call."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import heterofusionrcnn_amd as hf
from bench import time_op, kitti_uniform
rng = np.random.default_rng(0)
x = torch.from_numpy(kitti_uniform(rng, 8, 16384)).cuda()
s4 = hf.gather_point(x, hf.farthest_point_sample(4096, x))
s1 = hf.gather_point(s4, hf.farthest_point_sample(1024, s4))
print("three_nn 16384 x 4096 x 8: %.1f us" % time_op(lambda: hf.three_nn(x, s4), iters=20, warm=3))
print("three_nn 4096 x 1024 x 8: %.1f us" % time_op(lambda: hf.three_nn(s4, s1), iters=20, warm=3))
for k in (8, 12, 16, 5, 17):
    print("knn k=%d 16384 data x 4096 queries: %.1f us" % (k, time_op(lambda: hf.knn_point(k, x, s4), iters=20, warm=3)))
    print("knn k=%d 16384 data x 16384 queries: %.1f us" % (k, time_op(lambda: hf.knn_point(k, x, x), iters=20, warm=3)))
    print("knn k=%d 4096 data x 16384 queries: %.1f us" % (k, time_op(lambda: hf.knn_point(k, s4, x), iters=20, warm=3)))
