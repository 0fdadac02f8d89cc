import os, sys, traceback
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import heterofusionrcnn_amd as hf
from heterofusionrcnn_amd import rpn as rpn_mod
from heterofusionrcnn_amd.pipeline import GeometryPrefetcher
from bench import kitti_uniform, N0
cfg = rpn_mod.rpn_cars_pointnet_paper()
model = rpn_mod.RpnModel(cfg).cuda()
opt = torch.optim.Adam(model.parameters(), lr=1e-3, fused=True)
rng = np.random.default_rng(0)
bsz = 8
xyz = torch.from_numpy(kitti_uniform(rng, bsz, N0)).cuda()
inten = torch.from_numpy(rng.uniform(-.5, .5, (bsz, N0, 1)).astype(np.float32)).cuda()
gt_boxes, gt_cls = rpn_mod.synthetic_ground_truth(rng, bsz, 12, cfg, ground_y=3.0)
label_cls, label_reg = rpn_mod.point_labels(xyz, torch.from_numpy(gt_boxes).cuda(), torch.from_numpy(gt_cls).cuda())
pf = GeometryPrefetcher(model.geometry, depth=2)
pf.submit(xyz); pf.submit(xyz)
def step(stage):
    geo = pf.get(); stage.append("get")
    pf.submit(xyz); stage.append("submit")
    opt.zero_grad(set_to_none=True)
    seg_logits, head = model(xyz, inten, geometry=geo); stage.append("fwd")
    loss, _ = model.loss(xyz, seg_logits, head, label_cls, label_reg); stage.append("loss")
    loss.backward(); stage.append("bwd")
    opt.step(); stage.append("opt")
for _ in range(3): step([])
torch.cuda.synchronize()
torch.cuda.set_sync_debug_mode("error")
st = []
try:
    step(st)
    print("no sync error; stages", st)
except Exception as e:
    print("sync after stages", st)
    traceback.print_exc(limit=12)
