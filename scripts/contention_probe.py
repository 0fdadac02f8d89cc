"""Diagnostic: which main-stream kernels slow down while farthest point sampling (8 CUs busy) runs on a side stream?"""
import sys, json, torch, numpy as np
sys.path.insert(0, '.')
import heterofusionrcnn_amd as hf
from heterofusionrcnn_amd.mlp import _splitk_wgrad, linear_wgrad, BatchNormReLU
from bench import kitti_uniform
xyz = torch.from_numpy(kitti_uniform(np.random.default_rng(0), 8, 16384)).cuda()
side = torch.cuda.Stream(); side2 = torch.cuda.Stream()

def timed(fn, reps, with_fps):
    fn(); torch.cuda.synchronize()
    if with_fps:
        for s in (side, side2)[:with_fps]:
            with torch.cuda.stream(s):
                for _ in range(3): hf.farthest_point_sample(4096, xyz)   # ~13 ms of FPS per stream in the background
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3

import os
if os.environ.get('BLAS'): torch.backends.cuda.preferred_blas_library(os.environ['BLAS'])
print('blas', torch.backends.cuda.preferred_blas_library())
shapes = [(1048576, 4, 32), (1048576, 32, 32), (1048576, 32, 64), (262144, 67, 64), (262144, 64, 96), (262144, 96, 128),
          (65536, 131, 128), (65536, 128, 196), (65536, 196, 256), (8192, 384, 256), (8192, 256, 256), (32768, 320, 256),
          (32768, 256, 256), (131072, 129, 128), (131072, 128, 128)]
tot = {}
for rows, cin, cout in shapes:
    x = torch.randn(rows, cin, device='cuda'); w = torch.randn(cout, cin, device='cuda'); b = torch.randn(cout, device='cuda')
    dz = torch.randn(rows, cout, device='cuda'); z = torch.randn(rows, cout, device='cuda')
    bn = BatchNormReLU(cout).cuda()
    zz = z.clone().requires_grad_(True)
    def bn_fb():
        y = bn(zz); y.backward(dz)
    ops = {"fwd": lambda: torch.addmm(b, x, w.t()), "dgrad": lambda: dz @ w, "wgrad_bmm": lambda: _splitk_wgrad(dz, x),
           }
    row = {}
    reps = 20 if rows > 100000 else 40
    for name, fn in ops.items():
        t0 = timed(fn, reps, 0); t1 = timed(fn, reps, 1); t2 = timed(fn, reps, 2)
        row[name] = (t0, t1, t2)
        for i, t in enumerate((t0, t1, t2)): tot[(name, i)] = tot.get((name, i), 0) + t
    print(rows, cin, cout, {k: tuple(round(v) for v in vs) for k, vs in row.items()}, flush=True)
print({k: round(v) for k, v in tot.items()})
