"""Diagnostic: which nn.Linear shapes of a workload cost what (forward only, one event pair per call)."""
import os, sys, collections
import numpy as np, torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import heterofusionrcnn_amd as hf
from heterofusionrcnn_amd import rpn as rpn_mod
from bench import kitti_uniform, N0
wl = sys.argv[1] if len(sys.argv) > 1 else "rpn_multiclass"
cfg = rpn_mod.rpn_multiclass() if wl == "rpn_multiclass" else rpn_mod.rpn_cars_pointnet_paper()
model = rpn_mod.RpnModel(cfg).cuda()
rng = np.random.default_rng(0)
xyz = torch.from_numpy(kitti_uniform(rng, 8, N0)).cuda()
inten = torch.from_numpy(rng.uniform(-.5, .5, (8, N0, 1)).astype(np.float32)).cuda()
geo = model.geometry(xyz)
orig = F.linear
recs = []
def hooked(x, w, b=None):
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record(); y = orig(x, w, b); e1.record()
    recs.append((x.numel() // x.shape[-1], x.shape[-1], w.shape[0], e0, e1))
    return y
for it in range(2):
    recs.clear()
    F.linear = hooked
    torch.nn.functional.linear = hooked
    out = model(xyz, inten, geometry=geo)
    F.linear = orig
    torch.cuda.synchronize()
agg = collections.OrderedDict()
for rows, cin, cout, e0, e1 in recs:
    k = (rows, cin, cout)
    a = agg.setdefault(k, [0, 0.0]); a[0] += 1; a[1] += e0.elapsed_time(e1) * 1e3
tot = sum(v[1] for v in agg.values())
print("forward nn.Linear total %.0f us over %d calls" % (tot, len(recs)))
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:25]:
    rows, cin, cout = k
    mb = 4 * (rows * cin + rows * cout) / 1e6
    print("rows %8d cin %5d cout %5d  x%d  %8.0f us  (%.0f MB moved -> %.0f us at 5 TB/s; %.1f GFLOP)" % (rows, cin, cout, v[0], v[1], mb, mb / 5, 2e-9 * rows * cin * cout))
