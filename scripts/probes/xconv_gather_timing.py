"""Diagnostic: the X-Conv product with the neighbours' features read in place (hf_xconv_depthwise_gather) against the route it
replaces (concat_group -> hf_xconv_depthwise on the materialised F_*), forward and backward, at the layer shapes of
rpn_multiclass.config with 8 frames.  Run under rocprofv3 --kernel-trace for per-kernel durations."""
import os, sys, json
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import heterofusionrcnn_amd as hf
from heterofusionrcnn_amd import pointcnn as pc
from heterofusionrcnn_amd.grouping import concat_group, index_inverse, knn_point
from bench import kitti_frustum, time_op
rng = np.random.default_rng(0)
B, K = 8, 8
xyz = torch.from_numpy(kitti_frustum(rng, B, 16384)).cuda()
lvl = {16384: xyz}
for n in (4096, 1024):
    src = lvl[n * 4]
    lvl[n] = hf.gather_point(src, hf.farthest_point_sample(n, src))
res = {}
for name, n, p, c0, c1, m in (("dec5", 16384, 16384, 64, 256, 1), ("dec4", 4096, 16384, 64, 256, 1), ("enc1", 16384, 4096, 64, 256, 1),
                              ("dec3", 1024, 4096, 128, 512, 1), ("enc0", 16384, 16384, 64, 1, 4)):
    _, idx = knn_point(K, lvl[n], lvl[p])
    inv = index_inverse(idx, n)
    x = torch.randn(B, p, K, K, device="cuda", requires_grad=True)
    fd = torch.randn(B, p, K, c0, device="cuda", requires_grad=True)
    fts = torch.randn(B, n, c1, device="cuda", requires_grad=True)
    wd = torch.randn(K, c0 + c1, m, device="cuda", requires_grad=True)
    go = torch.randn(B, p, (c0 + c1) * m, device="cuda")
    def old():
        out = pc.xconv_depthwise(x, concat_group(fd, fts, idx, inv), wd)
        torch.autograd.grad(out, (x, fd, fts, wd), go)
    def new():
        out = pc.xconv_depthwise_gather(x, fd, fts, idx, wd, inv)
        torch.autograd.grad(out, (x, fd, fts, wd), go)
    res[name] = {"concat_route_us": round(time_op(old, iters=5, warm=2), 1), "gather_route_us": round(time_op(new, iters=5, warm=2), 1)}
    del x, fd, fts, wd, go
    torch.cuda.empty_cache()
print(json.dumps(res, indent=1))
