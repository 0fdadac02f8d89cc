"""Diagnostic: slab count / LDS geometry sweep of the slab ball-query kernel at the headline shape."""
import os, sys, json
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import heterofusionrcnn_amd as hf
from bench import kitti_uniform, time_op
rng = np.random.default_rng(0)
xyz = torch.from_numpy(kitti_uniform(rng, 8, 16384)).cuda()
new_xyz = hf.gather_point(xyz, hf.farthest_point_sample(4096, xyz))
ref = hf.query_ball_group(0.5, 32, xyz, new_xyz, True)
res = {}
for slabs, qcap, ccap in ((32, 256, 3072), (32, 192, 2048), (48, 128, 1536), (64, 128, 1280), (64, 96, 1024), (96, 64, 1024), (128, 64, 768)):
    os.environ.update(HF_QBP_SLABS=str(slabs), HF_QBP_QCAP=str(qcap), HF_QBP_CCAP=str(ccap))
    out = hf.query_ball_group(0.5, 32, xyz, new_xyz, True)
    ok = all(torch.equal(a, b) for a, b in zip(out, ref))
    us = time_op(lambda: hf.query_ball_group(0.5, 32, xyz, new_xyz, True), iters=100, warm=10)
    res["S%d_q%d_c%d" % (slabs, qcap, ccap)] = "%.2f us %s" % (us, "ok" if ok else "MISMATCH")
print(json.dumps(res, indent=1))
