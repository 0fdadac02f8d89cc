"""A few launches of compute_bev_iou (70 000 x 64) and oriented_nms (9000 boxes, uniform and clustered) for
`rocprofv3 --kernel-trace --stats` / `--pmc` passes (scripts/collect_evidence.sh): short, so that a counter pass is cheap."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import heterofusionrcnn_amd as hf
from bench import rand_bev
rng = np.random.default_rng(3)
a = torch.from_numpy(rand_bev(rng, 70000)).cuda()
g = torch.from_numpy(rand_bev(rng, 64)).cuda()
nb = torch.from_numpy(rand_bev(rng, 9000)).cuda()
r2 = np.random.default_rng(4)
cl = np.repeat(rand_bev(r2, 300), 30, 0)
cl[:, [0, 2]] += r2.normal(0, 0.3, (9000, 1)).astype(np.float32)
cl[:, [1, 3]] += r2.normal(0, 0.3, (9000, 1)).astype(np.float32)
cl[:, 4] += r2.normal(0, 0.1, 9000).astype(np.float32)
nbc = torch.from_numpy(cl.astype(np.float32)).cuda()
for _ in range(12):
    hf.compute_bev_iou(a, g)
torch.cuda.synchronize()
for _ in range(6):
    hf.oriented_nms(nb, 0.8)
torch.cuda.synchronize()
for _ in range(6):
    hf.oriented_nms(nbc, 0.8)
torch.cuda.synchronize()
print("done")
