"""Diagnostic: what bounds the replayed one-frame step.  The captured RPN step (rpn_multiclass, image feature map resident) timed
  A  graph replay alone, same static inputs                 (INVALID as a benchmark: no input refresh, no geometry)
  B  refresh of the 39 input slots + replay (same geometry)  (INVALID: geometry not recomputed)
  C  the bench's step: geometry of the next batches on side streams + refresh + replay
usage: replay_only.py [frames] [x-branch on|off]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from heterofusionrcnn_amd import rpn as rpn_mod, pointcnn as pointcnn_mod
from heterofusionrcnn_amd.optim import MultiTensorAdam
from heterofusionrcnn_amd.graph_step import TrainStep
from heterofusionrcnn_amd.pipeline import GeometryPrefetcher
from bench import kitti_frustum, N0, IMG_C, IMG_H, IMG_W, KITTI_P2
FR = int(sys.argv[1]) if len(sys.argv) > 1 else 1
pointcnn_mod.CONCURRENT_X_BRANCH = (sys.argv[2] if len(sys.argv) > 2 else "on") == "on"
pointcnn_mod.CONCURRENT_X_BRANCH_MIN_POINTS = int(os.environ.get("HF_X_MIN_POINTS", "0"))
ONLY_AB = os.environ.get("HF_ONLY_AB") == "1"
pointcnn_mod.CONCURRENT_LIFT_BRANCH_MAX_ROWS = int(os.environ.get("HF_LIFT", "131072"))
pointcnn_mod.CONCURRENT_RUN_AHEAD = os.environ.get("HF_AHEAD", "1") == "1"
print("lift branch on a side stream up to %s rows, run ahead %s" % (pointcnn_mod.CONCURRENT_LIFT_BRANCH_MAX_ROWS, pointcnn_mod.CONCURRENT_RUN_AHEAD))
rng = np.random.default_rng(0)
cfg = rpn_mod.rpn_multiclass(IMG_C)
model = rpn_mod.RpnModel(cfg).cuda()
xyz = torch.from_numpy(kitti_frustum(rng, FR, N0)).cuda()
inputs = {"xyz": xyz, "intensity": torch.from_numpy(rng.uniform(-.5, .5, (FR, N0, 1)).astype(np.float32)).cuda()}
gt_boxes, gt_cls = rpn_mod.synthetic_ground_truth(rng, FR, 12, cfg, ground_y=3.0)
inputs["label_cls"], inputs["label_reg"] = rpn_mod.point_labels(xyz, torch.from_numpy(gt_boxes).cuda(), torch.from_numpy(gt_cls).cuda())
inputs["img_fts"] = torch.randn(FR, IMG_H, IMG_W, IMG_C, device="cuda").requires_grad_(True)
inputs["calib"] = torch.from_numpy(KITTI_P2).cuda().repeat(FR, 1, 1).contiguous()
opt = MultiTensorAdam(list(model.parameters()), lr=1e-3, tf_epsilon=False)
train = TrainStep(model, opt, inputs, model.geometry(xyz), world=1, graph=True)
geo = model.geometry(xyz)
def timed(fn, n=40):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    th = time.perf_counter() - t0
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3, th / n * 1e3
print("frames %d, X branch on a side stream: %s" % (FR, pointcnn_mod.CONCURRENT_X_BRANCH))
print("A replay alone                    %.3f ms per step (host %.3f)" % timed(lambda: train.graph.replay()))
print("B slots refreshed + replay        %.3f ms per step (host %.3f)" % timed(lambda: train(geometry=geo)))
print("fork threshold (points): %d" % pointcnn_mod.CONCURRENT_X_BRANCH_MIN_POINTS)
if ONLY_AB:
    sys.exit(0)
for depth in (1, 2, 3):
    pre = GeometryPrefetcher(model.geometry, depth=depth, group=1)
    for _ in range(pre.capacity): pre.submit(xyz)
    def step():
        g = pre.get(); pre.submit(xyz); train(geometry=g)
    print("C geometry prefetched (depth %d)   %.3f ms per step (host %.3f)" % ((depth,) + timed(step)))
    while len(pre): pre.get()
    torch.cuda.synchronize()
def inline():
    train(geometry=model.geometry(xyz))
print("D geometry inline                 %.3f ms per step (host %.3f)" % timed(inline))
