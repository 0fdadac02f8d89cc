"""bev_iou ops -- same surface as the reference's bev_iou/bev_iou.py:13-39.

`bev_iou` is the BASELINE.json alias of compute_bev_iou."""
import torch

from . import _lib
from ._lib import check, dev_tensor, ptr, require, stream_ptr


def compute_bev_iou(proposals, gt_bboxes):
    """proposals (N,5), gt_bboxes (M,5) as [x1,y1,x2,y2,ry] -> (overlap_area (N,M), bev_iou (N,M)).
    Reference: bev_iou.py:13-24; non-differentiable."""
    require(proposals.dim() == 2 and proposals.shape[0] > 0 and proposals.shape[1] == 5,
            "ComputeIOU3D expects (N, 5) proposals shape")
    require(gt_bboxes.dim() == 2 and gt_bboxes.shape[0] > 0 and gt_bboxes.shape[1] == 5,
            "ComputeIOU3D expects (M, 5) gt_bboxes shape")
    a = dev_tensor(proposals.detach(), torch.float32, "proposals")
    b = dev_tensor(gt_bboxes.detach(), torch.float32, "gt_bboxes")
    na, nb = a.shape[0], b.shape[0]
    overlap = torch.empty((na, nb), dtype=torch.float32, device=a.device)
    iou = torch.empty((na, nb), dtype=torch.float32, device=a.device)
    check(_lib.lib().hf_compute_bev_iou(na, ptr(a), nb, ptr(b), ptr(overlap), ptr(iou), stream_ptr()),
          "compute_bev_iou")
    return overlap, iou


bev_iou = compute_bev_iou


def oriented_nms(boxes, thresh, return_count=False):
    """boxes (N,5) score-sorted [x1,y1,x2,y2,ry] -> keep_idx (N) int32: kept indices ascending, the
    tail padded with keep[0].  Reference: bev_iou.py:30-39 / OrientedNMSOp (bev_iou.cpp:27-124).
    Stays on the device: no host sweep, no synchronisation."""
    thresh = float(thresh)
    require(thresh >= 0, "Need nms_threshold >= 0, got %r" % thresh)
    require(boxes.dim() == 2 and boxes.shape[0] > 0 and boxes.shape[1] == 5,
            "OrientendNMS expects (N, 5) boxes shape")
    boxes = dev_tensor(boxes.detach(), torch.float32, "boxes")
    n = boxes.shape[0]
    L = _lib.lib()
    ws_bytes = L.hf_oriented_nms_workspace(n)
    ws = torch.empty((ws_bytes // 8,), dtype=torch.int64, device=boxes.device)
    keep = torch.empty((n,), dtype=torch.int32, device=boxes.device)
    num = torch.empty((1,), dtype=torch.int32, device=boxes.device)
    check(L.hf_oriented_nms(ptr(boxes), n, thresh, ptr(keep), ptr(num), ptr(ws), ws_bytes, stream_ptr()),
          "oriented_nms")
    return (keep, num) if return_count else keep


def oriented_nms_batched(boxes, thresh):
    """boxes (F,N,5), each frame score-sorted -> (keep (F,N) int32, num_kept (F) int32): oriented_nms for every
    frame of a batch in one launch pair (the reference loops frames with tf.map_fn, rpn_model.py:683-687)."""
    thresh = float(thresh)
    require(thresh >= 0, "Need nms_threshold >= 0, got %r" % thresh)
    require(boxes.dim() == 3 and boxes.shape[0] > 0 and boxes.shape[1] > 0 and boxes.shape[2] == 5,
            "oriented_nms_batched expects (F, N, 5) boxes shape")
    boxes = dev_tensor(boxes.detach(), torch.float32, "boxes")
    f, n, _ = boxes.shape
    L = _lib.lib()
    ws_bytes = f * L.hf_oriented_nms_workspace(n)
    ws = torch.empty((ws_bytes // 8,), dtype=torch.int64, device=boxes.device)
    keep = torch.empty((f, n), dtype=torch.int32, device=boxes.device)
    num = torch.empty((f,), dtype=torch.int32, device=boxes.device)
    check(L.hf_oriented_nms_batched(f, ptr(boxes), n, thresh, ptr(keep), ptr(num), ptr(ws), ws_bytes, stream_ptr()),
          "oriented_nms_batched")
    return keep, num


def nms_mask(boxes, thresh):
    """The raw suppression bit mask of oriented_nms_gpu (bev_iou.cpp:40): (N, ceil(N/64)) int64 words."""
    require(boxes.dim() == 2 and boxes.shape[0] > 0 and boxes.shape[1] == 5, "nms_mask expects (N, 5) boxes shape")
    boxes = dev_tensor(boxes.detach(), torch.float32, "boxes")
    n = boxes.shape[0]
    mask = torch.empty((n, (n + 63) // 64), dtype=torch.int64, device=boxes.device)
    check(_lib.lib().hf_nms_mask(ptr(boxes), ptr(mask), n, float(thresh), stream_ptr()), "nms_mask")
    return mask
