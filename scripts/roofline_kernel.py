"""Launch ONLY the roofline kernel (fused ball query + group at the headline shape) plus two calibration
copies, for rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE need separate passes on gfx950)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import heterofusionrcnn_amd as hf
from bench import kitti_uniform, B, N0, KNN, SA

rng = np.random.default_rng(0)
xyz = torch.from_numpy(kitti_uniform(rng, B, N0)).cuda()
new_xyz = hf.gather_point(xyz, hf.farthest_point_sample(SA[0][0], xyz))
torch.cuda.synchronize()
# calibration: a 256 MiB float4 copy (known 256 MiB read + 256 MiB written)
src = torch.empty(64 * 1024 * 1024, dtype=torch.float32, device="cuda").normal_()
for _ in range(3):
    dst = src.clone()
torch.cuda.synchronize()
for _ in range(20):
    hf.query_ball_group(SA[0][1], KNN, xyz, new_xyz, True)
torch.cuda.synchronize()
print("done")
