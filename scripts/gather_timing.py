"""group_point (C=16, 64, 128) and three_interpolate (C=128, 256, 512) at the headline shapes: GB/s of algorithmic bytes."""
import os, sys, json
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import heterofusionrcnn_amd as hf
from bench import kitti_uniform, time_op, B, N0, KNN
rng = np.random.default_rng(0)
xyz = torch.from_numpy(kitti_uniform(rng, B, N0)).cuda()
new_xyz = hf.gather_point(xyz, hf.farthest_point_sample(4096, xyz))
idx, _ = hf.query_ball_point(0.5, KNN, xyz, new_xyz)
res = {}
for c in (16, 64, 128):
    f = torch.randn(B, N0, c, device="cuda")
    us = time_op(lambda: hf.group_point(f, idx), iters=30)
    res["group_point_c%d_us" % c] = round(us, 1)
    res["group_point_c%d_GBs" % c] = round((4 * B * N0 * c + 4 * B * 4096 * KNN + 4 * B * 4096 * KNN * c) / us / 1e3, 0)
dist, i3 = hf.three_nn(xyz, new_xyz)
w = torch.rand(B, N0, 3, device="cuda")
for c in (128, 256, 512):
    p = torch.randn(B, 4096, c, device="cuda")
    us = time_op(lambda: hf.three_interpolate(p, i3, w), iters=30)
    res["three_interpolate_c%d_us" % c] = round(us, 1)
    res["three_interpolate_c%d_GBs" % c] = round((4 * B * 4096 * c + 2 * 4 * 3 * B * N0 + 4 * B * N0 * c) / us / 1e3, 0)
print(json.dumps(res, indent=1))
