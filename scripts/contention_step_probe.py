"""rocprofv3 --kernel-trace target: 10 train steps with fixed geometry, then 10 with FPS running on side streams.
The two phases are separated by a burst of 7 torch.arange kernels (marker)."""
import os, sys, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from heterofusionrcnn_amd import modules, farthest_point_sample, gather_point
from bench import kitti_uniform, B, N0, SA, FP
torch.manual_seed(0)
model = modules.PointnetSAFPStack(in_channel=1, sa=SA, fp=FP).cuda()
opt = torch.optim.Adam(model.parameters(), lr=1e-3, fused=True)
rng = np.random.default_rng(0)
xyz = torch.from_numpy(kitti_uniform(rng, B, N0)).cuda()
inten = torch.from_numpy(rng.uniform(-.5, .5, (B, N0, 1)).astype(np.float32)).cuda()
geo = model.geometry(xyz)
def step():
    opt.zero_grad(set_to_none=True)
    model(xyz, inten, geometry=geo).mean().backward(); opt.step()
for _ in range(3): step()
torch.cuda.synchronize()
for _ in range(10): step()
torch.cuda.synchronize()
for _ in range(7): torch.arange(1000, device='cuda')
torch.cuda.synchronize()
sides = [torch.cuda.Stream(), torch.cuda.Stream()]
for i in range(10):
    with torch.cuda.stream(sides[i % 2]):
        a = farthest_point_sample(4096, xyz); x1 = gather_point(xyz, a)
        b_ = farthest_point_sample(1024, x1); x2 = gather_point(x1, b_)
        farthest_point_sample(256, x2)
    step()
torch.cuda.synchronize()
