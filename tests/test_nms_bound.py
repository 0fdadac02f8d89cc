"""The NMS mask kernel's third filter (csrc/bev_iou.hip: iou_surely_below) rejects a pair without clipping it when an upper bound on
the overlap area already puts the IoU below 0.98 x threshold.  Its soundness is a property of the formula, checked here on the CPU:
the kernel's fp32 expressions restated in numpy, against the oracle's IoU (bev_iou/bev_iou_g.cu:102-215 restated in
oracle/hf_oracle.c), on clusters of near-duplicate, aligned, touching and thin boxes -- a rejected pair never has iou > thresh."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402

f32 = np.float32


def _corners(b):
    """BoxPre.cor of csrc/bev_iou.hip (rot_center of the four corners), fp32"""
    x1, y1, x2, y2, ang = [b[:, i].astype(f32) for i in range(5)]
    cx, cy = (x1 + x2) / f32(2), (y1 + y2) / f32(2)
    cs, sn = np.cos(ang).astype(f32), np.sin(ang).astype(f32)
    out = []
    for (px, py) in ((x1, y1), (x2, y1), (x2, y2), (x1, y2)):
        out.append(np.stack([(px - cx) * cs + (py - cy) * sn + cx, -(px - cx) * sn + (py - cy) * cs + cy], -1).astype(f32))
    return np.stack(out, 1)


def _rejected(a, b, ca, cb, thresh):
    wa, ha, wb, hb = a[2] - a[0], a[3] - a[1], b[2] - b[0], b[3] - b[1]
    if not (wa > 0 and ha > 0 and wb > 0 and hb > 0):
        return False
    slack = f32(1e-3) + f32(1e-5) * max(np.abs(ca).max(), np.abs(cb).max())
    ub = np.inf
    for p, w, h in ((ca, wa, ha), (cb, wb, hb)):
        both = f32(1.0)
        for e in range(2):
            u = p[e + 1] - p[e]
            along, across = (w, h) if e == 0 else (h, w)
            pa, pb = ca @ u, cb @ u
            ov = max(min(pa.max(), pb.max()) - max(pa.min(), pb.min()), 0) / along + slack
            ub = min(ub, ov * (across + slack))
            both = both * ov
        ub = min(ub, both)
    den = wa * ha + wb * hb - ub
    return bool(den > 0 and ub < f32(0.98) * f32(thresh) * den)


def test_overlap_bound_never_rejects_a_pair_above_the_threshold():
    rng = np.random.default_rng(5)
    rejected = above = 0
    for trial in range(14):
        n = 40
        bx, bz = rng.uniform(-40, 40, 6), rng.uniform(0, 70, 6)
        k = rng.integers(0, 6, n)
        jit = rng.choice([0.0, 0.02, 0.1, 0.3, 1.0])
        cx, cz = bx[k] + rng.normal(0, jit, n), bz[k] + rng.normal(0, jit, n)
        l, w = np.clip(rng.normal(3.9, 0.4, n), 0.05, None), np.clip(rng.normal(1.6, 0.3, n), 0.05, None)
        if trial % 3:
            ry = rng.uniform(-np.pi, np.pi, 6)[k] + rng.normal(0, rng.choice([0, 0.01, 0.1, 0.5]), n)
        else:
            ry = rng.choice([0, np.pi / 2, np.pi, -np.pi / 2], n)          # axis-aligned: touching edges, shared corners
        boxes = np.stack([cx - l / 2, cz - w / 2, cx + l / 2, cz + w / 2, ry], 1).astype(f32)
        _, iou = oracle.compute_bev_iou(boxes, boxes)
        cor = _corners(boxes)
        for thresh in (0.8, 0.5, 0.1, 0.01):
            for i in range(n):
                for j in range(n):
                    rej = _rejected(boxes[i], boxes[j], cor[i], cor[j], thresh)
                    rejected += rej
                    above += iou[i, j] > thresh
                    assert not (rej and iou[i, j] > thresh), (thresh, float(iou[i, j]), boxes[i], boxes[j])
    assert rejected > 10000 and above > 1000, "the sweep must exercise both outcomes"
