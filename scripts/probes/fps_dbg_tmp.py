import sys, numpy as np, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import heterofusionrcnn_amd as hf
from bench import kitti_frustum
rng = np.random.default_rng(0)
for n, m in ((512, 128), (1024, 256), (2048, 512), (4096, 1024), (8192, 2048), (16384, 4096), (16384, 64), (12000, 3000)):
    xyz = torch.from_numpy(kitti_frustum(rng, 2, n)).cuda()
    ref = hf.farthest_point_sample(m, xyz, kernel="plain", threads=512 if n > 4096 else 256)
    for nt in (1024, 512):
        out = hf.farthest_point_sample(m, xyz, kernel="bucket", threads=nt)
        bad = (out != ref).nonzero()
        print(n, m, nt, "ok" if len(bad) == 0 else "first mismatch at %s of %d mismatches" % (bad[0].tolist(), len(bad)))
