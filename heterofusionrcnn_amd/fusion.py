"""The LiDAR -> image fusion step: project rectified-camera points into the image with the P2
calibration matrix and pick up the image feature vector under each point.

Reference: hf/core/projection.py:5-32 (tf_rect_to_image) and hf/core/models/rpn_model.py:227-235 (tf.cast to int32,
index tensor [b, v, u], tf.gather_nd).  Here it is one HIP kernel (csrc/glue.hip); a point that projects outside the
image receives zeros, as tf.gather_nd does on GPU.
"""
import torch

from . import _lib
from ._lib import check, dev_tensor, ptr, require, stream_ptr


def rect_to_image(pts3d, calib):
    """pts3d (B,N,3) rectified-camera points, calib (B,3,4) P2 -> (B,N,2) float pixel coordinates [x, y]
    (projection.py:5-32; plain torch: the fused op below never materialises it)"""
    require(pts3d.dim() == 3 and pts3d.shape[2] == 3 and calib.shape == (pts3d.shape[0], 3, 4),
            "rect_to_image expects (B,N,3) points and (B,3,4) calib")
    hom = torch.cat([pts3d, torch.ones_like(pts3d[..., :1])], dim=-1)
    pix = torch.einsum("bij,bnj->bni", calib, hom)
    return pix[..., :2] / pix[..., 2:3]


class _ProjectGather(torch.autograd.Function):
    @staticmethod
    def forward(ctx, img_fts, pts3d, calib):
        b, h, w, c = img_fts.shape
        p = pts3d.shape[1]
        out = torch.empty((b, p, c), dtype=torch.float32, device=img_fts.device)
        pix = torch.empty((b, p, 2), dtype=torch.int32, device=img_fts.device)
        check(_lib.lib().hf_project_gather(b, p, h, w, c, ptr(pts3d), ptr(calib), ptr(img_fts), ptr(out), ptr(pix),
                                           stream_ptr()), "project_gather")
        ctx.save_for_backward(pix)
        ctx.shape = (b, p, h, w, c)
        ctx.mark_non_differentiable(pix)
        return out, pix

    @staticmethod
    def backward(ctx, grad_out, _grad_pix):
        (pix,) = ctx.saved_tensors
        b, p, h, w, c = ctx.shape
        grad_out = grad_out.contiguous()
        g = torch.empty((b, h, w, c), dtype=torch.float32, device=grad_out.device)
        check(_lib.lib().hf_project_gather_grad(b, p, h, w, c, ptr(grad_out), ptr(pix), ptr(g), stream_ptr()),
              "project_gather_grad")
        return g, None, None


def project_gather(pts3d, calib, img_fts, return_pixels=False):
    """pts3d (B,P,3), calib (B,3,4), img_fts (B,H,W,C) -> (B,P,C): the image features under the projected points
    (rpn_model.py:227-235).  Gradient w.r.t. img_fts only (pixel indices are integers).  return_pixels also returns
    the int32 (B,P,2) [u, v] pixel of every point."""
    require(pts3d.dim() == 3 and pts3d.shape[2] == 3, "project_gather expects (B,P,3) points")
    require(img_fts.dim() == 4 and img_fts.shape[0] == pts3d.shape[0], "project_gather expects (B,H,W,C) image features")
    require(calib.shape == (pts3d.shape[0], 3, 4), "project_gather expects (B,3,4) calib")
    pts3d = dev_tensor(pts3d.detach(), torch.float32, "pts3d")
    calib = dev_tensor(calib.detach(), torch.float32, "calib")
    img_fts = dev_tensor(img_fts, torch.float32, "img_fts")
    out, pix = _ProjectGather.apply(img_fts, pts3d, calib)
    return (out, pix) if return_pixels else out


def fuse_point_image_features(pc_fts, proj_img_fts, method="concat", div=2.0):
    """hf/core/models/rpn_model.py:537-548: "mean" = (pc + img) / div (equal widths), "concat" = [pc, img]"""
    if method == "mean":
        require(pc_fts.shape[-1] == proj_img_fts.shape[-1], "mean fusion needs equal feature widths")
        return (pc_fts + proj_img_fts) / div
    require(method == "concat", "Invalid fusion method %r" % (method,))
    return torch.cat([pc_fts, proj_img_fts], dim=-1)
