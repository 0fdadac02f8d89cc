"""Diagnostic: FPS kernels (plain / bucketed, 256 / 512 / 1024 threads) at the sampling levels of the bench workloads; every
variant must return the same indices."""
import os, sys, json
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import heterofusionrcnn_amd as hf
from bench import kitti_uniform, kitti_frustum, time_op
rng = np.random.default_rng(0)
res = {}
for cloud, gen in (("uniform", kitti_uniform), ("frustum", kitti_frustum)):
    for n, m in ((16384, 4096), (8192, 2048), (4096, 1024), (1024, 256)):
        xyz = torch.from_numpy(gen(rng, 8, n)).cuda()
        ref = None
        for mode, nt in (("plain", 1024), ("plain", 512), ("plain", 256), ("bucket", 1024), ("bucket", 512), ("auto", 0)):
            if mode == "plain" and (n > nt * (32 if nt <= 512 else 16) or (nt == 256 and n > 4096)):
                continue
            out = hf.farthest_point_sample(m, xyz, kernel=mode, threads=nt)
            if ref is None:
                ref = out
            assert torch.equal(out, ref), (mode, nt, n)
            us = time_op(lambda: hf.farthest_point_sample(m, xyz, kernel=mode, threads=nt), iters=3, warm=1)
            res["%s %s%s %d->%d" % (cloud, mode, nt or "", n, m)] = "%.1f us  %.3f us/round" % (us, us / (m - 1))
print(json.dumps(res, indent=1))
