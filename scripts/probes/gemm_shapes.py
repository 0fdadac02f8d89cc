"""Diagnostic: every aten mm / addmm / bmm of one RPN train step (forward AND backward) with its shape and device time
(one event pair per call) -- which library GEMMs the step spends its time in."""
import os, sys, collections
import numpy as np, torch
from torch.utils._python_dispatch import TorchDispatchMode
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import heterofusionrcnn_amd as hf
from heterofusionrcnn_amd import rpn as rpn_mod
from bench import kitti_uniform, N0
wl = sys.argv[1] if len(sys.argv) > 1 else "rpn"
FR = int(sys.argv[2]) if len(sys.argv) > 2 else 8      # frames per step
cfg = rpn_mod.rpn_multiclass() if wl == "rpn_multiclass" else rpn_mod.rpn_cars_pointnet_paper()
model = rpn_mod.RpnModel(cfg).cuda()
opt = torch.optim.Adam(model.parameters(), lr=1e-3, fused=True)
rng = np.random.default_rng(0)
xyz = torch.from_numpy(kitti_uniform(rng, FR, N0)).cuda()
inten = torch.from_numpy(rng.uniform(-.5, .5, (FR, N0, 1)).astype(np.float32)).cuda()
gt_boxes, gt_cls = rpn_mod.synthetic_ground_truth(rng, FR, 12, cfg, ground_y=3.0)
lc, lr = rpn_mod.point_labels(xyz, torch.from_numpy(gt_boxes).cuda(), torch.from_numpy(gt_cls).cuda())
geo = model.geometry(xyz)
recs = []
class Rec(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = func.__name__ if hasattr(func, "__name__") else str(func)
        if name.split(".")[0] in ("mm", "addmm", "bmm", "baddbmm"):
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record(); out = func(*args, **(kwargs or {})); e1.record()
            ts = [tuple(a.shape) for a in args if isinstance(a, torch.Tensor)]
            st = ["T" if (isinstance(a, torch.Tensor) and a.dim() >= 2 and a.stride(-1) != 1) else "N" for a in args if isinstance(a, torch.Tensor)]
            recs.append((name.split(".")[0], tuple(ts), "".join(st), e0, e1))
            return out
        return func(*args, **(kwargs or {}))
def step():
    opt.zero_grad(set_to_none=True)
    seg, head = model(xyz, inten, geometry=geo)
    loss, _ = model.loss(xyz, seg, head, lc, lr)
    loss.backward(); opt.step()
for _ in range(2): step()
torch.cuda.synchronize()
with Rec():
    step()
torch.cuda.synchronize()
agg = collections.OrderedDict()
for name, ts, st, e0, e1 in recs:
    a = agg.setdefault((name, ts, st), [0, 0.0]); a[0] += 1; a[1] += e0.elapsed_time(e1) * 1e3
tot = sum(v[1] for v in agg.values())
print("library GEMM calls %d, total %.0f us" % (len(recs), tot))
def flops(name, ts):
    a, b = (ts[-2], ts[-1])
    m, k = a[-2], a[-1]
    n = b[-1]
    batch = a[0] if len(a) == 3 else 1
    return 2.0 * batch * m * k * n
fl = 0.0
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:60]:
    f = flops(k[0], k[1]) * v[0]
    print("%-8s %-52s %s x%d %8.0f us %7.1f TFLOP/s" % (k[0], str(k[1]), k[2], v[0], v[1], f / (v[1] * 1e-6) / 1e12))
print("all library GEMMs: %.1f GFLOP, %.1f TFLOP/s over their own time" % (sum(flops(k[0], k[1]) * v[0] for k, v in agg.items()) / 1e9,
      sum(flops(k[0], k[1]) * v[0] for k, v in agg.items()) / (tot * 1e-6) / 1e12))
