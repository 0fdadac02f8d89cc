"""Diagnostic: every call through the C-ABI during one RPN train step (forward AND backward), grouped by entry point and leading
integer arguments (rows / channels / ...), with device time per group (one event pair per call; eager, so each figure includes the
gaps between the launches of one entry point) -- which shapes the step's small launches belong to.
usage: abi_call_shapes.py [frames] [name-filter]"""
import os, sys, collections, ctypes
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import heterofusionrcnn_amd as hf
from heterofusionrcnn_amd import rpn as rpn_mod, _lib
from heterofusionrcnn_amd.optim import MultiTensorAdam
from bench import kitti_uniform, N0
FR = int(sys.argv[1]) if len(sys.argv) > 1 else 1
FILT = sys.argv[2] if len(sys.argv) > 2 else ""
cfg = rpn_mod.rpn_multiclass()
model = rpn_mod.RpnModel(cfg).cuda()
opt = MultiTensorAdam(model.parameters(), lr=1e-3)
rng = np.random.default_rng(0)
xyz = torch.from_numpy(kitti_uniform(rng, FR, N0)).cuda()
inten = torch.from_numpy(rng.uniform(-.5, .5, (FR, N0, 1)).astype(np.float32)).cuda()
gt_boxes, gt_cls = rpn_mod.synthetic_ground_truth(rng, FR, 12, cfg, ground_y=3.0)
lc, lr = rpn_mod.point_labels(xyz, torch.from_numpy(gt_boxes).cuda(), torch.from_numpy(gt_cls).cuda())
geo = model.geometry(xyz)
real = _lib.lib()
recs, on = [], [False]
class Proxy:
    def __getattr__(self, name):
        fn = getattr(real, name)
        if not name.startswith("hf_") or "workspace" in name or name in ("hf_strerror", "hf_last_hip_error", "hf_adam_chunk"):
            return fn
        def call(*a):
            if not on[0]:
                return fn(*a)
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record(); r = fn(*a); e1.record()
            ints = []
            for v in a:
                if isinstance(v, int) and not isinstance(v, bool) and abs(v) < (1 << 40):
                    ints.append(v)
                else:
                    break
            recs.append((name, tuple(ints), e0, e1))
            return r
        return call
_lib._lib = Proxy()
def step():
    opt.zero_grad(set_to_none=True)
    seg, head = model(xyz, inten, geometry=geo)
    loss, _ = model.loss(xyz, seg, head, lc, lr)
    loss.backward(); opt.step()
for _ in range(2): step()
torch.cuda.synchronize()
on[0] = True
step()
torch.cuda.synchronize()
agg = collections.OrderedDict()
for name, ints, e0, e1 in recs:
    a = agg.setdefault((name, ints), [0, 0.0]); a[0] += 1; a[1] += e0.elapsed_time(e1) * 1e3
print("C-ABI calls %d, %.0f us (eager, event pairs)" % (len(recs), sum(v[1] for v in agg.values())))
byname = collections.OrderedDict()
for (name, ints), v in agg.items():
    b = byname.setdefault(name, [0, 0.0]); b[0] += v[0]; b[1] += v[1]
for name, v in sorted(byname.items(), key=lambda kv: -kv[1][1]):
    print("%-34s x%-4d %8.0f us" % (name, v[0], v[1]))
print()
for (name, ints), v in sorted(agg.items(), key=lambda kv: (kv[0][0], -kv[1][1])):
    if FILT in name:
        print("%-34s %-40s x%-3d %8.0f us  %6.1f us each" % (name, ints, v[0], v[1], v[1] / v[0]))
