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


def path_drop_masks(p_img, p_pc, random_values):
    """create_path_drop_masks (rpn_model.py:1130-1193, rcnn_model.py:1264-1327): three coin flips decide which branch a
    training step keeps.  random_values (3,) uniform in [0,1) on the device; returns a (2,) tensor [pc_mask, img_mask] of
    0.0 / 1.0 on the device -- no host read, so the decision rides along in a captured graph.
      img = r0 < p_img ; pc = r1 < p_pc ; both killed -> r2 > 0.5 keeps the image, r2 <= 0.5 keeps the points."""
    r = random_values
    img = (r[0] < p_img)
    pc = (r[1] < p_pc)
    both_dead = ~(img | pc)
    img = torch.where(both_dead, r[2] > 0.5, img)
    pc = torch.where(both_dead, r[2] <= 0.5, pc)
    return torch.stack([pc, img]).to(torch.float32)


class _FuseConcat(torch.autograd.Function):
    """hf_fuse_concat (+ grad): [pc * mask_pc | img * mask_img] in one pass"""

    @staticmethod
    def forward(ctx, a, b, masks):
        c1, c2 = a.shape[-1], b.shape[-1]
        a2, b2 = a.reshape(-1, c1).contiguous(), b.reshape(-1, c2).contiguous()
        out = torch.empty((a2.shape[0], c1 + c2), dtype=torch.float32, device=a.device)
        check(_lib.lib().hf_fuse_concat(a2.shape[0], c1, c2, ptr(a2), ptr(b2), ptr(masks), ptr(out), stream_ptr()), "fuse_concat")
        ctx.save_for_backward(masks) if masks is not None else ctx.save_for_backward()
        ctx.shapes = (tuple(a.shape), tuple(b.shape))
        return out.reshape(*a.shape[:-1], c1 + c2)

    @staticmethod
    def backward(ctx, g):
        masks = ctx.saved_tensors[0] if ctx.saved_tensors else None
        sa, sb = ctx.shapes
        c1, c2 = sa[-1], sb[-1]
        g = g.reshape(-1, c1 + c2).contiguous()
        ga = torch.empty((g.shape[0], c1), dtype=torch.float32, device=g.device) if ctx.needs_input_grad[0] else None
        gb = torch.empty((g.shape[0], c2), dtype=torch.float32, device=g.device) if ctx.needs_input_grad[1] else None
        check(_lib.lib().hf_fuse_concat_grad(g.shape[0], c1, c2, ptr(g), ptr(masks), ptr(ga), ptr(gb), stream_ptr()),
              "fuse_concat_grad")
        return (ga.reshape(sa) if ga is not None else None), (gb.reshape(sb) if gb is not None else None), None


def fuse_point_image_features(pc_fts, proj_img_fts, method="concat", div=2.0, masks=None):
    """hf/core/models/rpn_model.py:515-548: optional path drop (masks = path_drop_masks(...): [pc_mask, img_mask]), then
    "mean" = (pc + img) / div (equal widths; div = img_mask + pc_mask under path drop) or "concat" = [pc, img]"""
    if method == "mean":
        require(pc_fts.shape[-1] == proj_img_fts.shape[-1], "mean fusion needs equal feature widths")
        if masks is not None:
            return (pc_fts * masks[0] + proj_img_fts * masks[1]) / (masks[0] + masks[1])
        return (pc_fts + proj_img_fts) / div
    require(method == "concat", "Invalid fusion method %r" % (method,))
    if pc_fts.is_cuda and pc_fts.dtype == torch.float32 and proj_img_fts.dtype == torch.float32:
        return _FuseConcat.apply(pc_fts, proj_img_fts, masks)
    if masks is not None:
        pc_fts, proj_img_fts = pc_fts * masks[0], proj_img_fts * masks[1]
    return torch.cat([pc_fts, proj_img_fts], dim=-1)
