"""The second stage (RCNN) on the HIP ops, as hf/core/models/rcnn_model.py builds it for hf/configs/rcnn_multiclass.config.

  proposals -> image boxes                 rcnn_model.py:419-452   projection.tf_project_to_image_space, reorder to [y1,x1,y2,x2]
  expand by the pooling context            :455-471                sizes + 2 c, (bottom) y + c
  RoI pooling on the cloud                 :475-486                box corners -> pc_crop_and_sample(resize 512)   [HIP op]
  RoI pooling on the image feature map     :491-500                tf.image.crop_and_resize(7 x 7)
  local spatial features                   :502-541                canonical transform, intensity, foreground mask, distance
  mlp 256 / 256 on them (pf.dense)         :543-561
  PointCNN on [crop_fts | mlp]             :563-567, config :157-186   4 xconv layers (K 4 / 8 / 12 / 12; 512 -> 128 -> 32 -> 8 points)
  path drop + 'flat_concat' fusion         :571-595
  classification head (fc 256/256 -> K+1)  :599-634
  refinement head (fc 256/256 -> K x 46)   :638-664                bin-based residuals around the proposal
  decode + per-frame oriented NMS          :670-778                tf_decode(proposal centre, proposal heading), class of the best
                                                                   foreground score, empty RoIs dropped, NMS 0.01, 100 boxes

Training losses of the second stage (:780-1000) are not part of BASELINE config 5 (inference) and are not built.
TensorFlow cannot be imported here: the layer semantics follow rpn.py / pointcnn.py (pf.dense = linear -> ELU -> BatchNorm);
parity unpinned against reference outputs, pinned against the text by tests/test_rcnn.py.
"""
import math
from dataclasses import dataclass, field
from typing import Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import box_codec, modules
from .cropping import pc_crop_and_sample
from .fusion import image_crop_and_resize, path_drop_masks, project_boxes_to_image
from .pointcnn import Dense, PointCnnBackbone, PointCnnConfig


def rcnn_pointcnn_config(in_channel):
    """rcnn_multiclass.config:157-186: four xconv layers, no decoder, no fc (the heads have their own)"""
    return PointCnnConfig(xconv=((4, 1, -1, 512), (8, 1, 128, 512), (12, 1, 32, 1024), (12, 1, 8, 1024)), xdconv=(), fc=(),
                          in_channel=in_channel, with_x=True, with_global=True)


@dataclass
class RcnnConfig:
    """The values of hf/configs/rcnn_multiclass.config that shape the second stage."""
    num_classes: int = 3
    roi_crop_size: int = 512                                               # rcnn_proposal_roi_crop_size
    roi_img_crop_size: int = 7                                             # rcnn_proposal_roi_img_crop_size
    nms_size: int = 100
    nms_iou_thresh: float = 0.01
    xz_search_range: Tuple[float, ...] = (1.5, 0.75, 0.75)
    xz_bin_len: Tuple[float, ...] = (0.5, 0.25, 0.25)
    theta_search_range: float = 0.25                                       # fraction of pi
    theta_bin_len_deg: float = 10.0
    pooling_context_length: float = 1.0
    fusion: str = "flat_concat"                                            # | 'mean_concat'
    path_drop: Tuple[float, float] = (0.9, 0.9)
    mlp: Tuple[Tuple[int, float], ...] = ((256, 0.5), (256, 0.5))          # on the local spatial features
    fc: Tuple[Tuple[int, float], ...] = ((256, 0.5), (256, 0.5))           # each head
    rpn_fts_channels: int = 256 + 32                                       # PL_RPN_FTS (rcnn_model.py:177-181)
    img_channels: int = 32
    img_hw: Tuple[int, int] = (360, 1200)
    use_intensity: bool = True
    cluster_sizes: Tuple[Tuple[float, float, float], ...] = ((3.88, 1.63, 1.53), (0.84, 0.66, 1.76), (1.76, 0.6, 1.73))
    bev_extent_z: float = 70.0                                             # bev_extents[1, 1]: normalises the sensor distance
    pointcnn: PointCnnConfig = None

    def __post_init__(self):
        if self.pointcnn is None:
            self.pointcnn = rcnn_pointcnn_config(self.rpn_fts_channels + self.mlp[-1][0])

    @property
    def num_bin_xz(self):
        return int(2 * self.xz_search_range[0] / self.xz_bin_len[0])       # rcnn_model.py:111-112

    @property
    def r_theta(self):
        return self.theta_search_range * math.pi                           # :114

    @property
    def delta_theta(self):
        return self.theta_bin_len_deg * math.pi / 180.0                    # :115

    @property
    def num_bin_theta(self):
        return int(2 * self.r_theta / self.delta_theta)                    # :116

    @property
    def head_width(self):
        return 4 * self.num_bin_xz + 2 * self.num_bin_theta + 4


def canonical_transform(pts, boxes_3d):
    """rcnn_model.py:207-235: translate to the box centre, rotate by -ry about y.  pts (N,R,3), boxes (N,7)"""
    shift = pts - boxes_3d[:, None, 0:3]
    ry = -boxes_3d[:, 6]
    c, s = torch.cos(ry)[:, None], torch.sin(ry)[:, None]
    x = c * shift[:, :, 0] + s * shift[:, :, 2]
    z = -s * shift[:, :, 0] + c * shift[:, :, 2]
    return torch.stack([x, shift[:, :, 1], z], dim=2)


def expand_proposals(proposals, context):
    """rcnn_model.py:455-471: sizes grow by 2*context, the (bottom-centre) y moves down by context"""
    out = proposals.clone()
    out[:, 1] = proposals[:, 1] + context
    out[:, 3:6] = proposals[:, 3:6] + 2 * context
    return out


class _FcStack(nn.Module):
    """pf.dense + tf.layers.dropout per configured layer"""

    def __init__(self, cin, layers):
        super().__init__()
        self.layers = nn.ModuleList()
        self.rates = []
        for (w, rate) in layers:
            self.layers.append(Dense(cin, w))
            self.rates.append(rate)
            cin = w
        self.out_channel = cin

    def forward(self, x):
        for layer, rate in zip(self.layers, self.rates):
            x = layer(x, dropout=rate if self.training else 0.0)
        return x


class RcnnModel(nn.Module):
    def __init__(self, cfg: RcnnConfig = None):
        super().__init__()
        self.cfg = cfg = cfg or RcnnConfig()
        local_c = 3 + (1 if cfg.use_intensity else 0) + 1 + 1              # canonical xyz, intensity, mask, distance
        self.mlp = _FcStack(local_c, cfg.mlp)
        assert cfg.pointcnn.in_channel == cfg.rpn_fts_channels + self.mlp.out_channel
        self.encoder = PointCnnBackbone(cfg.pointcnn)
        r = [p if p > 0 else cfg.roi_crop_size for (_, _, p, _) in cfg.pointcnn.xconv][-1]
        self.roi_points = r
        if cfg.fusion == "flat_concat":
            fuse_c = r * self.encoder.out_channel + cfg.roi_img_crop_size ** 2 * cfg.img_channels
        else:
            fuse_c = self.encoder.out_channel + cfg.img_channels
        self.cls_fc = _FcStack(fuse_c, cfg.fc)
        self.cls_logits = nn.Linear(self.cls_fc.out_channel, cfg.num_classes + 1)        # with_bn=False, activation=None
        nn.init.xavier_normal_(self.cls_logits.weight)
        nn.init.zeros_(self.cls_logits.bias)
        self.reg_fc = _FcStack(fuse_c, cfg.fc)
        self.reg_out = Dense(self.reg_fc.out_channel, cfg.head_width * cfg.num_classes, activation=False)

    # ------------------------------------------------------------------ RoI pooling
    def roi_pool(self, xyz, rpn_fts, intensity, fg_mask, proposals, img_fts, calib):
        """-> dict of the pooled tensors; proposals (B,n,7)"""
        cfg = self.cfg
        b, n, _ = proposals.shape
        flat = proposals.reshape(-1, 7).contiguous()
        box_ind = torch.arange(b, device=xyz.device, dtype=torch.int32).repeat_interleave(n)
        _, box2d_norm = project_boxes_to_image(proposals, calib, cfg.img_hw)                 # (B,n,4) [x1,y1,x2,y2]
        yxyx = box2d_norm.reshape(-1, 4)[:, [1, 0, 3, 2]]                                   # reorder_projected_boxes
        boxes8 = modules.box_3d_to_box_8co(expand_proposals(flat, cfg.pooling_context_length)).contiguous()
        crop_pts, crop_fts, crop_int, crop_mask, crop_ind, non_empty = pc_crop_and_sample(
            xyz, rpn_fts.contiguous(), intensity, fg_mask, boxes8, box_ind, cfg.roi_crop_size)
        img_rois = image_crop_and_resize(img_fts, yxyx, box_ind, cfg.roi_img_crop_size)            # (N,7,7,C1)
        return {"flat": flat, "box_ind": box_ind, "crop_pts": crop_pts, "crop_fts": crop_fts, "crop_int": crop_int,
                "crop_mask": crop_mask, "crop_ind": crop_ind, "non_empty": non_empty, "img_rois": img_rois, "boxes8": boxes8,
                "yxyx": yxyx}

    def local_features(self, pool):
        """rcnn_model.py:502-541"""
        crop_pts = pool["crop_pts"]
        pts_ct = canonical_transform(crop_pts, pool["flat"]).contiguous()
        dist = torch.sqrt(crop_pts[..., 0] ** 2 + crop_pts[..., 1] ** 2 + crop_pts[..., 2] ** 2) / self.cfg.bev_extent_z - 0.5
        parts = [pts_ct]
        if self.cfg.use_intensity:
            parts.append(pool["crop_int"])
        parts += [pool["crop_mask"].unsqueeze(-1).float(), dist.unsqueeze(-1)]
        return pts_ct, torch.cat(parts, dim=-1)

    # ------------------------------------------------------------------ forward
    def forward(self, xyz, rpn_fts, intensity, fg_mask, proposals, img_fts, calib):
        """-> (cls_logits (N,K+1), reg_output (N,K,D), pool dict), N = B n RoIs"""
        cfg = self.cfg
        pool = self.roi_pool(xyz, rpn_fts, intensity, fg_mask, proposals, img_fts, calib)
        pts_ct, local = self.local_features(pool)
        merged = torch.cat([pool["crop_fts"], self.mlp(local)], dim=-1)                      # (N,R,2C)
        pc_rois = self.encoder(pts_ct, merged)                                               # (N,r,C')
        img_rois = pool["img_rois"]
        if self.training and tuple(cfg.path_drop) != (1.0, 1.0):
            masks = path_drop_masks(cfg.path_drop[0], cfg.path_drop[1], torch.rand(3, device=xyz.device))
            pc_rois, img_rois = pc_rois * masks[0], img_rois * masks[1]
        nroi = pc_rois.shape[0]
        if cfg.fusion == "flat_concat":
            fuse = torch.cat([pc_rois.reshape(nroi, -1), img_rois.reshape(nroi, -1)], dim=-1)
        else:
            fuse = torch.cat([pc_rois.mean(dim=1), img_rois.mean(dim=(1, 2))], dim=-1)
        cls_logits = self.cls_logits(self.cls_fc(fuse))
        reg = self.reg_out(self.reg_fc(fuse)).reshape(nroi, cfg.num_classes, cfg.head_width)
        return cls_logits, reg, pool

    @torch.no_grad()
    def detect(self, xyz, rpn_fts, intensity, fg_mask, proposals, img_fts, calib):
        """Test-mode outputs of the second stage for a batch of frames: per frame the boxes that survive the oriented NMS
        (threshold 0.01, at most nms_size), their scores and classes.  rcnn_model.py:670-778: decode around the proposal
        (its centre and heading), the class with the best foreground score, empty RoIs dropped before the NMS.  All frames go
        through ONE batched device NMS and one host read of the counts (the reference loops frames with tf.map_fn)."""
        from .bev_iou import oriented_nms_batched
        cfg = self.cfg
        b, n, _ = proposals.shape
        cls_logits, reg, pool = self.forward(xyz, rpn_fts, intensity, fg_mask, proposals, img_fts, calib)
        prob = torch.softmax(cls_logits, dim=-1)
        score, fg_cls = prob[:, 1:].max(dim=-1)
        flat = pool["flat"]
        boxes = box_codec.decode_head(reg, flat[:, 0:3].contiguous(), flat[:, 6].contiguous(), cfg.cluster_sizes, cfg.num_bin_xz,
                                      cfg.num_bin_xz, cfg.num_bin_theta, cfg.xz_search_range, cfg.xz_bin_len, cfg.r_theta,
                                      cfg.delta_theta, cls=fg_cls)
        non_empty = pool["non_empty"].view(b, n)
        score, boxes, fg_cls = score.view(b, n), boxes.view(b, n, 7), fg_cls.view(b, n)
        # empty RoIs leave the list (tf.boolean_mask, :731-733): here they sort behind every real box and are parked far away
        # from everything (they can neither suppress nor be suppressed), then dropped from the keep list
        key = torch.where(non_empty, score, torch.full_like(score, -1.0))
        order = torch.sort(key, dim=1, descending=True, stable=True).indices
        bev = torch.gather(modules.boxes3d_to_bev(boxes), 1, order.unsqueeze(-1).expand(-1, -1, 5))
        ne_sorted = torch.gather(non_empty, 1, order)
        park = torch.arange(n, device=xyz.device, dtype=bev.dtype)[None, :, None] * 1.0e3 + 1.0e6
        parked = torch.cat([park + torch.zeros_like(bev[..., :2]), park + 1.0 + torch.zeros_like(bev[..., :2]), torch.zeros_like(bev[..., :1])], dim=-1)
        bev = torch.where(ne_sorted.unsqueeze(-1), bev, parked)
        keep, num = oriented_nms_batched(bev.contiguous(), cfg.nms_iou_thresh)
        n_real = ne_sorted.sum(dim=1)
        # kept entries are in score order: the real ones come first, so the first min(#kept real, nms_size) are the answer
        kept_real = (torch.gather(ne_sorted, 1, keep.long()) & (torch.arange(n, device=xyz.device)[None] < num[:, None])).sum(dim=1)
        counts = torch.minimum(kept_real, torch.full_like(kept_real, cfg.nms_size)).tolist()
        ind = torch.gather(order, 1, keep.long())
        out = []
        for i in range(b):
            sel = ind[i, :counts[i]]
            out.append({"boxes": boxes[i, sel], "scores": score[i, sel], "classes": fg_cls[i, sel] + 1})
        return out, {"pool": pool, "cls_prob": prob, "boxes_all": boxes, "order": order, "keep": keep, "num": num, "n_real": n_real}
