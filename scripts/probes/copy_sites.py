"""Diagnostic: where do the framework's copy kernels of one RPN train step come from?  Every aten clone / copy_ /
contiguous-producing call with its shape, device time and the Python line of this package that triggered it."""
import os, sys, collections, traceback
import numpy as np, torch
from torch.utils._python_dispatch import TorchDispatchMode
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import heterofusionrcnn_amd as hf
from heterofusionrcnn_amd import rpn as rpn_mod
from bench import kitti_uniform, N0
wl = sys.argv[1] if len(sys.argv) > 1 else "rpn_multiclass"
cfg = rpn_mod.rpn_multiclass() if wl == "rpn_multiclass" else rpn_mod.rpn_cars_pointnet_paper()
model = rpn_mod.RpnModel(cfg).cuda()
opt = torch.optim.Adam(model.parameters(), lr=1e-3, fused=True)
rng = np.random.default_rng(0)
xyz = torch.from_numpy(kitti_uniform(rng, 8, N0)).cuda()
inten = torch.from_numpy(rng.uniform(-.5, .5, (8, N0, 1)).astype(np.float32)).cuda()
gt_boxes, gt_cls = rpn_mod.synthetic_ground_truth(rng, 8, 12, cfg, ground_y=3.0)
lc, lr = rpn_mod.point_labels(xyz, torch.from_numpy(gt_boxes).cuda(), torch.from_numpy(gt_cls).cuda())
geo = model.geometry(xyz)
recs = []
class Rec(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func).split(".")[1] if "." in str(func) else str(func)
        if name in ("clone", "copy_", "_to_copy", "cat", "elu", "elu_backward", "native_dropout", "native_dropout_backward", "add", "mul", "sub"):
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record(); out = func(*args, **(kwargs or {})); e1.record()
            site = "?"
            for fr in reversed(traceback.extract_stack(limit=25)):
                if "heterofusionrcnn_amd" in fr.filename and "probes" not in fr.filename:
                    site = "%s:%d" % (os.path.basename(fr.filename), fr.lineno); break
            t = next((a for a in args if isinstance(a, torch.Tensor)), None)
            if t is None and args and isinstance(args[0], (list, tuple)): t = args[0][0]
            recs.append((name, tuple(t.shape) if t is not None else (), site, e0, e1))
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
for name, shape, site, e0, e1 in recs:
    a = agg.setdefault((name, site), [0, 0.0, shape]); a[0] += 1; a[1] += e0.elapsed_time(e1) * 1e3
tot = sum(v[1] for v in agg.values())
print("recorded calls %d, total %.0f us" % (len(recs), tot))
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:28]:
    print("%-24s %-22s x%-3d %8.0f us  e.g. %s" % (k[0], k[1], v[0], v[1], v[2]))
