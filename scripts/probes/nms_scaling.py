"""oriented_nms at several box counts (uniform boxes, threshold 0.8) for a rocprofv3 kernel trace: how the sweep kernel's duration
splits into a fixed part and a per-column-block part"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import heterofusionrcnn_amd as hf
from bench import rand_bev
rng = np.random.default_rng(3)
allb = rand_bev(rng, 9000)
for n in (9000, 4544, 2240, 1088, 64):
    nb = torch.from_numpy(allb[:n].copy()).cuda()
    for _ in range(4):
        hf.oriented_nms(nb, 0.8)
    torch.cuda.synchronize()
print("done")
