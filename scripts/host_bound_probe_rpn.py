"""Diagnostic: how much of the RPN train step is host submission?  The same step at B = 8, 2, 1 frames: the number of
launches does not change with B, the device work does -- where the step time stops falling with B, the host is the limit.
Also reports host-synchronising calls inside a step (torch's sync debug mode)."""
import os, sys, time, json, warnings
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import heterofusionrcnn_amd as hf
from heterofusionrcnn_amd import rpn as rpn_mod
from heterofusionrcnn_amd.pipeline import GeometryPrefetcher
from bench import kitti_uniform, N0
torch.manual_seed(0)
out = {}
for bsz in (8, 2, 1):
    cfg = rpn_mod.rpn_cars_pointnet_paper()
    model = rpn_mod.RpnModel(cfg).cuda()
    opt = torch.optim.Adam(model.parameters(), lr=1e-3, fused=True)
    rng = np.random.default_rng(0)
    xyz = torch.from_numpy(kitti_uniform(rng, bsz, N0)).cuda()
    inten = torch.from_numpy(rng.uniform(-.5, .5, (bsz, N0, 1)).astype(np.float32)).cuda()
    gt_boxes, gt_cls = rpn_mod.synthetic_ground_truth(rng, bsz, 12, cfg, ground_y=3.0)
    label_cls, label_reg = rpn_mod.point_labels(xyz, torch.from_numpy(gt_boxes).cuda(), torch.from_numpy(gt_cls).cuda())
    pf = GeometryPrefetcher(model.geometry, depth=2)
    pf.submit(xyz); pf.submit(xyz)
    def step():
        geo = pf.get(); pf.submit(xyz)
        opt.zero_grad(set_to_none=True)
        seg_logits, head = model(xyz, inten, geometry=geo)
        loss, _ = model.loss(xyz, seg_logits, head, label_cls, label_reg)
        loss.backward(); opt.step()
    for _ in range(3): step()
    torch.cuda.synchronize()
    if bsz == 8:
        torch.cuda.set_sync_debug_mode("warn")
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            step()
        torch.cuda.set_sync_debug_mode("default")
        out["synchronising_calls_in_a_step"] = sorted({str(x.message)[:120] for x in w})
        torch.cuda.synchronize()
    K = 10
    t0 = time.perf_counter()
    for _ in range(K): step()
    t_enq = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    out["B%d" % bsz] = {"enqueue_ms_per_step": round(1e3 * t_enq / K, 3), "wall_ms_per_step": round(1e3 * t_all / K, 3)}
print(json.dumps(out, indent=1))
