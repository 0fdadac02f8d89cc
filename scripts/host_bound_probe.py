"""Diagnostic: is the train step limited by host-side launch overhead?  Compare the time the Python loop needs
to ENQUEUE K steps with the time the device needs to finish them."""
import os, sys, time, json
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import heterofusionrcnn_amd as hf
from heterofusionrcnn_amd import modules
from heterofusionrcnn_amd.pipeline import GeometryPrefetcher
from bench import kitti_uniform, B, N0, SA, FP
torch.manual_seed(0)
model = modules.PointnetSAFPStack(in_channel=1, sa=SA, fp=FP).cuda()
opt = torch.optim.Adam(model.parameters(), lr=1e-3, fused=True)
rng = np.random.default_rng(0)
xyz = torch.from_numpy(kitti_uniform(rng, B, N0)).cuda()
inten = torch.from_numpy(rng.uniform(-.5, .5, (B, N0, 1)).astype(np.float32)).cuda()
pf = GeometryPrefetcher(model.geometry, depth=2)
pf.submit(xyz); pf.submit(xyz)
def step():
    geo = pf.get(); pf.submit(xyz)
    opt.zero_grad(set_to_none=True)
    loss = model(xyz, inten, geometry=geo).mean()
    loss.backward(); opt.step()
for _ in range(3): step()
torch.cuda.synchronize()
K = 20
t0 = time.perf_counter()
for _ in range(K): step()
t_enq = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print(json.dumps({"enqueue_ms_per_step": round(1e3 * t_enq / K, 3), "wall_ms_per_step": round(1e3 * t_all / K, 3)}))
