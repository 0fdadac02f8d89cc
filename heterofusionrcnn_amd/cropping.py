"""cropping op -- same surface as the reference's cropping/tf_cropping.py:13-36.

`crop_and_resize` is the BASELINE.json alias of pc_crop_and_sample."""
import torch

from . import _lib
from ._lib import check, dev_tensor, ptr, require, stream_ptr


def pc_crop_and_sample(pts, fts, intensities, mask, boxes, box_ind, resize):
    """pts (B,P,3), fts (B,P,C), intensities (B,P,1), mask (B,P) bool, boxes (N,3,8) corners,
    box_ind (N) int32, resize R ->
    (crop_pts (N,R,3), crop_fts (N,R,C), crop_intensity (N,R,1), crop_mask (N,R) bool,
     crop_ind (N,R) int32, non_empty_box_mask (N) bool).
    Reference: tf_cropping.py:13-33; NoGradient (:36).  Points of a box are taken in ascending
    point index (the reference's order is atomic arrival order)."""
    resize = int(resize)
    require(resize > 0, "PcCropAndSample expects positive resize")
    require(pts.dim() == 3 and pts.shape[1] > 0 and pts.shape[2] == 3, "PcCropAndSample expects (B, P, 3) pts shape")
    require(fts.dim() == 3 and fts.shape[1] > 0 and fts.shape[2] > 0, "PcCropAndSample expects (B, P, C) fts shape")
    require(intensities.dim() == 3 and intensities.shape[1] > 0 and intensities.shape[2] == 1,
            "PcCropAndSample expects (B, P, 1) intensities shape")
    require(pts.shape[0] == fts.shape[0] and pts.shape[1] == fts.shape[1],
            "PcCropAndSample expects pts & fts has same (B, P, ...) shape")
    require(intensities.shape[0] == pts.shape[0] and intensities.shape[1] == fts.shape[1],
            "PcCropAndSample expects intensities & fts has same (B, P, ...) shape")
    require(mask.dim() == 2 and tuple(mask.shape) == tuple(pts.shape[:2]), "PcCropAndSample expects (B, P) mask shape")
    require(boxes.dim() == 3 and boxes.shape[1] == 3 and boxes.shape[2] == 8, "boxes must be (N, 3, 8)")
    require(box_ind.dim() == 1 and box_ind.shape[0] == boxes.shape[0], "box_index has incompatible shape")
    pts = dev_tensor(pts.detach(), torch.float32, "pts")
    fts = dev_tensor(fts.detach(), torch.float32, "fts")
    intensities = dev_tensor(intensities.detach(), torch.float32, "intensities")
    mask = dev_tensor(mask, torch.bool, "mask")
    boxes = dev_tensor(boxes.detach(), torch.float32, "boxes")
    box_ind = dev_tensor(box_ind, torch.int32, "box_ind")
    bsz, npts, _ = pts.shape
    c, ic = fts.shape[2], intensities.shape[2]
    nb = boxes.shape[0]
    dev = pts.device
    crop_pts = torch.empty((nb, resize, 3), dtype=torch.float32, device=dev)
    crop_fts = torch.empty((nb, resize, c), dtype=torch.float32, device=dev)
    crop_int = torch.empty((nb, resize, ic), dtype=torch.float32, device=dev)
    crop_mask = torch.empty((nb, resize), dtype=torch.bool, device=dev)
    crop_ind = torch.empty((nb, resize), dtype=torch.int32, device=dev)
    non_empty = torch.empty((nb,), dtype=torch.bool, device=dev)
    check(_lib.lib().hf_pc_crop_and_sample(ptr(pts), ptr(fts), ptr(intensities), ptr(mask), ptr(boxes), ptr(box_ind),
                                           nb, bsz, npts, resize, c, ic, ptr(crop_pts), ptr(crop_fts), ptr(crop_int),
                                           ptr(crop_mask), ptr(crop_ind), ptr(non_empty), stream_ptr()),
          "pc_crop_and_sample")
    return crop_pts, crop_fts, crop_int, crop_mask, crop_ind, non_empty


crop_and_resize = pc_crop_and_sample


def pc_crop_and_sample_grad_fts(fts, box_ind, crop_ind, grad_crop_fts):
    """PcCropAndSampleGradFts op (tf_cropping.cpp:59-, kernel tf_cropping_g.cu:134-150): registered in the
    reference but unreachable from its Python (tf_cropping.py:36-52); exposed for completeness."""
    require(fts.dim() == 3 and grad_crop_fts.dim() == 3 and fts.shape[2] == grad_crop_fts.shape[2],
            "PcCropAndSampleGradFts expects fts and grad_crop_fts has same shape(2)")
    require(box_ind.shape[0] == crop_ind.shape[0] == grad_crop_fts.shape[0],
            "PcCropAndSampleGradFts expects box_ind, crop_ind and grad_crop_fts has same shape(0)")
    require(crop_ind.shape[1] == grad_crop_fts.shape[1],
            "PcCropAndSampleGradFts expects crop_ind and grad_crop_fts has same shape(1)")
    box_ind = dev_tensor(box_ind, torch.int32, "box_ind")
    crop_ind = dev_tensor(crop_ind, torch.int32, "crop_ind")
    grad_crop_fts = dev_tensor(grad_crop_fts.detach(), torch.float32, "grad_crop_fts")
    bsz, npts, c = fts.shape
    nb, resize = crop_ind.shape
    g = torch.empty((bsz, npts, c), dtype=torch.float32, device=grad_crop_fts.device)
    check(_lib.lib().hf_pc_crop_and_sample_grad_fts(ptr(box_ind), ptr(crop_ind), ptr(grad_crop_fts), nb, bsz, npts,
                                                    resize, c, ptr(g), stream_ptr()), "pc_crop_and_sample_grad_fts")
    return g
