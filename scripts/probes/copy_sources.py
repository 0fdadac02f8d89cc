"""Diagnostic: which lines of the package cause the framework's copy / elementwise / reduce launches of one RPN train step
(torch.profiler with stacks; aggregated by the innermost heterofusionrcnn_amd frame).  usage: copy_sources.py [frames]"""
import os, sys, collections
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from heterofusionrcnn_amd import rpn as rpn_mod
from heterofusionrcnn_amd.optim import MultiTensorAdam
from bench import kitti_uniform, N0
FR = int(sys.argv[1]) if len(sys.argv) > 1 else 1
cfg = rpn_mod.rpn_multiclass()
model = rpn_mod.RpnModel(cfg).cuda()
opt = MultiTensorAdam(model.parameters(), lr=1e-3)
rng = np.random.default_rng(0)
xyz = torch.from_numpy(kitti_uniform(rng, FR, N0)).cuda()
inten = torch.from_numpy(rng.uniform(-.5, .5, (FR, N0, 1)).astype(np.float32)).cuda()
gt_boxes, gt_cls = rpn_mod.synthetic_ground_truth(rng, FR, 12, cfg, ground_y=3.0)
lc, lr = rpn_mod.point_labels(xyz, torch.from_numpy(gt_boxes).cuda(), torch.from_numpy(gt_cls).cuda())
geo = model.geometry(xyz)
def step():
    opt.zero_grad(set_to_none=True)
    seg, head = model(xyz, inten, geometry=geo)
    loss, _ = model.loss(xyz, seg, head, lc, lr)
    loss.backward(); opt.step()
for _ in range(2): step()
torch.cuda.synchronize()
import traceback
from torch.utils._python_dispatch import TorchDispatchMode
SKIP = ("aten::view", "aten::_unsafe_view", "aten::reshape", "aten::t", "aten::transpose", "aten::permute", "aten::detach", "aten::slice",
        "aten::select", "aten::expand", "aten::unsqueeze", "aten::squeeze", "aten::as_strided", "aten::empty", "aten::alias", "aten::mm",
        "aten::addmm", "aten::bmm", "aten::empty_like", "aten::empty_strided", "aten::new_empty", "aten::split", "aten::unbind",
        "aten::_local_scalar_dense", "aten::lift_fresh", "aten::unfold", "aten::narrow", "aten::split_with_sizes", "aten::zeros", "aten::new_zeros")
agg = collections.OrderedDict()
class Rec(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = func._schema.name
        out = func(*args, **(kwargs or {}))
        if name in SKIP:
            return out
        t = next((a for a in args if isinstance(a, torch.Tensor)), None)
        if t is None or not t.is_cuda:
            return out
        frame = "(no package frame: autograd engine)"
        for f in reversed(traceback.extract_stack()):
            if "heterofusionrcnn_amd/" in f.filename:
                frame = "%s:%d %s" % (f.filename.split("heterofusionrcnn_amd/")[-1], f.lineno, f.name)
                break
        shp = tuple(t.shape)
        k = (name, frame, shp if name.startswith("aten::copy") or "clone" in name or "contiguous" in name else ())
        agg[k] = agg.get(k, 0) + 1
        return out
with Rec():
    step()
torch.cuda.synchronize()
print("aten ops on device tensors (views and GEMMs skipped): %d" % sum(agg.values()))
for k, v in sorted(agg.items(), key=lambda kv: (-kv[1], kv[0][0])):
    print("%-30s x%-3d %s %s" % (k[0], v, k[1], k[2] if k[2] else ""))
