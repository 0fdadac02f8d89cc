"""One set-abstraction level, forward + backward, through the in-place route (the first MLP layer gathers its operand:
mlp.shared_mlp_grouped) or the materialised one (group_concat -> shared_mlp), for rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes
and a kernel-trace pass.  usage: sa_gather_traffic.py {inplace|materialised} [level]   (level 1 / 2 / 3 of BASELINE config 2)"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import heterofusionrcnn_amd as hf
from heterofusionrcnn_amd import modules
from bench import kitti_uniform
mode = sys.argv[1]
level = int(sys.argv[2]) if len(sys.argv) > 2 else 2
modules.GATHER_ON_LOAD = mode == "inplace"
# config 2: 16384 -> 4096 (r 0.5, mlp 32/32/64, in 1) -> 1024 (r 1.0, 64/64/128, in 64) -> 256 (r 2.0, 128/128/256, in 128)
shape = {1: (16384, 4096, 0.5, 1, [32, 32, 64]), 2: (4096, 1024, 1.0, 64, [64, 64, 128]), 3: (1024, 256, 2.0, 128, [128, 128, 256])}[level]
n, m, r, c, mlp = shape
rng = np.random.default_rng(0)
xyz = torch.from_numpy(kitti_uniform(rng, 8, n)).cuda()
pts = torch.randn(8, n, c, device="cuda", requires_grad=True)
torch.manual_seed(0)
sa = modules.PointnetSAModule(m, r, 32, c, mlp).cuda().train()
geom = sa.geometry(xyz)
for _ in range(3):
    _, f, _ = sa(xyz, pts, geom)
    f.square().sum().backward()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    _, f, _ = sa(xyz, pts, geom)
    f.square().sum().backward()
e1.record()
torch.cuda.synchronize()
print("%s level %d: %.1f us per forward+backward" % (mode, level, 1e3 * e0.elapsed_time(e1) / 10))
