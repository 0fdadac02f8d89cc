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


def project_boxes_to_image(boxes_3d, calib, image_hw):
    """hf/core/projection.py:35-96 (tf_project_to_image_space) for every frame at once: boxes_3d (B,n,7) [x,y,z,l,w,h,ry]
    (y = bottom), calib (B,3,4), image_hw = (h, w) -> (corners (B,n,4) [x1,y1,x2,y2] in pixels, the same normalised by
    [w,h,w,h]): the 8 corners of a box are projected and the 2-D box is their bounding rectangle."""
    from .modules import box_3d_to_box_8co
    b, n, _ = boxes_3d.shape
    corners = box_3d_to_box_8co(boxes_3d.reshape(-1, 7)).transpose(1, 2).reshape(b, n * 8, 3)    # (B, n*8, 3)
    hom = torch.cat([corners, torch.ones_like(corners[..., :1])], dim=-1)
    pix = torch.einsum("bij,bnj->bni", calib, hom)
    pix = (pix / pix[..., 2:3])[..., :2].reshape(b, n, 8, 2)
    lo, hi = pix.min(dim=2).values, pix.max(dim=2).values
    box = torch.cat([lo, hi], dim=-1)
    h, w = image_hw
    return box, box / torch.tensor([w, h, w, h], dtype=box.dtype, device=box.device)


def image_crop_and_resize(img_fts, boxes_norm_yxyx, box_ind, crop_size, extrapolation_value=0.0):
    """tf.image.crop_and_resize (bilinear) as the RCNN uses it on the image feature map (rcnn_model.py:494-500): img_fts
    (B,H,W,C), boxes (N,4) [y1,x1,y2,x2] normalised, box_ind (N) -> (N,crop,crop,C).  Sample i of a crop sits at
    y1 (H-1) + i (y2-y1)(H-1)/(crop-1); a sample outside [0, H-1] x [0, W-1] takes the extrapolation value.
    Plain torch gathers (N * crop^2 * 4 rows of C floats); TensorFlow is not importable here: parity unpinned beyond the
    documented formula (tests/test_rcnn.py checks it against a loop)."""
    b, h, w, c = img_fts.shape
    n = boxes_norm_yxyx.shape[0]
    ch = cw = int(crop_size)
    y1, x1, y2, x2 = boxes_norm_yxyx.unbind(-1)
    steps_y = torch.arange(ch, device=img_fts.device, dtype=img_fts.dtype)
    steps_x = torch.arange(cw, device=img_fts.device, dtype=img_fts.dtype)
    if ch > 1:
        ys = y1[:, None] * (h - 1) + steps_y[None] * ((y2 - y1)[:, None] * (h - 1) / (ch - 1))
    else:
        ys = (0.5 * (y1 + y2) * (h - 1))[:, None]
    if cw > 1:
        xs = x1[:, None] * (w - 1) + steps_x[None] * ((x2 - x1)[:, None] * (w - 1) / (cw - 1))
    else:
        xs = (0.5 * (x1 + x2) * (w - 1))[:, None]
    oky = (ys >= 0) & (ys <= h - 1)
    okx = (xs >= 0) & (xs <= w - 1)
    y0f, x0f = torch.floor(ys), torch.floor(xs)
    ly, lx = (ys - y0f)[:, :, None, None], (xs - x0f)[:, None, :, None]
    y0 = y0f.long().clamp(0, h - 1)
    yb = torch.ceil(ys).long().clamp(0, h - 1)
    x0 = x0f.long().clamp(0, w - 1)
    xb = torch.ceil(xs).long().clamp(0, w - 1)
    bi = box_ind.long()[:, None, None]
    g = lambda yy, xx: img_fts[bi, yy[:, :, None], xx[:, None, :]]             # (N, ch, cw, C)
    top = g(y0, x0) + (g(y0, xb) - g(y0, x0)) * lx
    bot = g(yb, x0) + (g(yb, xb) - g(yb, x0)) * lx
    out = top + (bot - top) * ly
    ok = (oky[:, :, None] & okx[:, None, :])[..., None]
    return torch.where(ok, out, torch.full_like(out, extrapolation_value))
