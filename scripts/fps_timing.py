"""Diagnostic: FPS kernels (plain with NT threads / bucketed) at the SA levels of the bench workload."""
import os, sys, json
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import heterofusionrcnn_amd as hf
from bench import kitti_uniform, time_op
rng = np.random.default_rng(0)
res = {}
for n, m in ((16384, 4096), (4096, 1024), (1024, 256), (512, 128)):
    xyz = torch.from_numpy(kitti_uniform(rng, 8, n)).cuda()
    ref = None
    for mode, nt in (("plain", 1024), ("plain", 512), ("plain", 256), ("bucket", 0)):
        if mode == "plain" and n > nt * (32 if nt <= 512 else 16):
            continue
        if nt == 256 and n > 4096:
            continue
        os.environ["HF_FPS"] = mode
        os.environ["HF_FPS_THREADS"] = str(nt)
        out = hf.farthest_point_sample(m, xyz)
        if ref is None:
            ref = out
        assert torch.equal(out, ref), (mode, nt, n)
        us = time_op(lambda: hf.farthest_point_sample(m, xyz), iters=3, warm=1)
        res["%s%s_%d_%d" % (mode, nt or "", n, m)] = "%.1f us  %.3f us/round" % (us, us / (m - 1))
print(json.dumps(res, indent=1))
