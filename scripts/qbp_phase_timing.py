"""Diagnostic: cumulative phase times of the slab ball-query kernel at the headline shape.
HF_QBP_STOP=n makes the kernel return after phase n (outputs are then invalid)."""
import os, sys, json
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import heterofusionrcnn_amd as hf
from bench import kitti_uniform, time_op

rng = np.random.default_rng(0)
xyz = torch.from_numpy(kitti_uniform(rng, 8, 16384)).cuda()
new_xyz = hf.gather_point(xyz, hf.farthest_point_sample(4096, xyz))
res = {}
for slabs in (os.environ.get("SLABS", "32").split(",")):
    os.environ["HF_QBP_SLABS"] = slabs
    for stop in (-1, -2, -3, 1, 2, 3, 4, 5, 0):
        os.environ["HF_QBP_STOP"] = str(stop)
        res["S%s_fused_stop%d" % (slabs, stop)] = round(time_op(lambda: hf.query_ball_group(0.5, 32, xyz, new_xyz, True), iters=50), 2)
    os.environ["HF_QBP_STOP"] = "0"
    res["S%s_qbp_only" % slabs] = round(time_op(lambda: hf.query_ball_point(0.5, 32, xyz, new_xyz), iters=50), 2)
print(json.dumps(res, indent=1))
