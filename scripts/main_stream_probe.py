"""Diagnostic: train-step time with the geometry held fixed (no side-stream work at all) vs with the prefetcher."""
import os, sys, time, json
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from heterofusionrcnn_amd import modules
from heterofusionrcnn_amd.pipeline import GeometryPrefetcher
from bench import kitti_uniform, B, N0, SA, FP
torch.manual_seed(0)
model = modules.PointnetSAFPStack(in_channel=1, sa=SA, fp=FP).cuda()
opt = torch.optim.Adam(model.parameters(), lr=1e-3, fused=True)
rng = np.random.default_rng(0)
xyz = torch.from_numpy(kitti_uniform(rng, B, N0)).cuda()
inten = torch.from_numpy(rng.uniform(-.5, .5, (B, N0, 1)).astype(np.float32)).cuda()
geo_fixed = model.geometry(xyz)
def step(geo):
    opt.zero_grad(set_to_none=True)
    loss = model(xyz, inten, geometry=geo).mean()
    loss.backward(); opt.step()
def timeit(fn, k=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(k): fn()
    torch.cuda.synchronize(); return 1e3 * (time.perf_counter() - t0) / k
res = {"fixed_geometry_ms": timeit(lambda: step(geo_fixed))}
res["geometry_only_ms"] = timeit(lambda: model.geometry(xyz))
pf = GeometryPrefetcher(model.geometry, depth=2); pf.submit(xyz); pf.submit(xyz)
def pstep():
    geo = pf.get(); pf.submit(xyz); step(geo)
res["prefetch_depth2_ms"] = timeit(pstep)
# contention only: geometry of another batch computed on two free-running side streams, results discarded
sides = [torch.cuda.Stream(), torch.cuda.Stream()]
turn = [0]
def cstep():
    s = sides[turn[0] % 2]; turn[0] += 1
    with torch.cuda.stream(s):
        model.geometry(xyz)
    step(geo_fixed)
res["fixed_geometry_plus_free_running_side_ms"] = timeit(cstep)
torch.cuda.synchronize()
# only the FPS kernels on the side streams
from heterofusionrcnn_amd import farthest_point_sample, gather_point
def fps_only():
    a = farthest_point_sample(4096, xyz); x1 = gather_point(xyz, a)
    b_ = farthest_point_sample(1024, x1); x2 = gather_point(x1, b_)
    farthest_point_sample(256, x2)
def fstep():
    s = sides[turn[0] % 2]; turn[0] += 1
    with torch.cuda.stream(s):
        fps_only()
    step(geo_fixed)
res["fixed_geometry_plus_fps_only_side_ms"] = timeit(fstep)
torch.cuda.synchronize()
with torch.no_grad():
    res["forward_only_ms"] = timeit(lambda: model(xyz, inten, geometry=geo_fixed))
print(json.dumps({k: round(v, 3) for k, v in res.items()}))
