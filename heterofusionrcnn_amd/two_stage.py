"""Two-stage inference flow (BASELINE config 5) on the HIP ops, in miniature and with random weights.

What is restated from the reference is the DATA FLOW between the ops, not its network bodies:

  RPN forward -> per-point scores + boxes            hf/core/models/rpn_model.py:585-642
  (optional) LiDAR -> image fusion: project, gather  rpn_model.py:227-235, 537-548, 860 (fusion.project_gather)
  top-k pre_nms_size, oriented NMS, nms_size         rpn_model.py:645-687, model_util.py:101-142
  expand proposals by the pooling context            hf/core/models/rcnn_model.py:462-476
  box corners -> pc_crop_and_sample(resize)          rcnn_model.py:478-489
  canonical transform of the cropped points          rcnn_model.py:207-235
  RCNN forward on (N_roi, R, .) -> scores + boxes    rcnn_model.py:505-730
  per-frame oriented NMS (thresh 0.01, size 100)     rcnn_model.py:731-778

Frames are independent: `run_sharded` splits them rank-strided over the process group and rank 0 collects
the per-frame detections (no tensor collective), as the reference's single-process evaluator would
(hf/experiments/run_inference.py:152) if it were sharded.
"""
import math

import torch
import torch.nn as nn

from . import box_codec, dp, modules
from .bev_iou import oriented_nms_batched
from .cropping import pc_crop_and_sample
from .fusion import fuse_point_image_features, project_gather


def canonical_transform(pts, boxes_3d):
    """rcnn_model.py:207-235: translate to the box centre, rotate by -ry about y.  pts (N,R,3), boxes (N,7)"""
    shift = pts - boxes_3d[:, None, 0:3]
    ry = -boxes_3d[:, 6]
    c, s = torch.cos(ry)[:, None], torch.sin(ry)[:, None]
    x = c * shift[:, :, 0] + s * shift[:, :, 2]
    z = -s * shift[:, :, 0] + c * shift[:, :, 2]
    return torch.stack([x, shift[:, :, 1], z], dim=2)


def expand_proposals(proposals, context):
    """rcnn_model.py:462-476: sizes grow by 2*context, the (bottom-centre) y moves down by context"""
    out = proposals.clone()
    out[:, 1] = proposals[:, 1] + context
    out[:, 3:6] = proposals[:, 3:6] + 2 * context
    return out


class BoxHead(nn.Module):
    """Per-point / per-RoI score and a bin-based box around a reference position: the output layout and the decoding
    of the reference's heads (rpn_model.py:570-642; rcnn_model.py decodes the same way around the proposal, with its
    heading as ref_theta), one class.  The fc stack in front of it is a single Linear here (random weights)."""

    def __init__(self, cin, mean_size=(3.9, 1.6, 1.5), search_range=3.0, bin_len=0.5, theta_range=0.25, theta_bins=12):
        super().__init__()
        self.nb_xz = int(2 * search_range / bin_len)           # NUM_BIN_X = NUM_BIN_Z (rpn_model.py:115-116)
        self.nb_theta = theta_bins
        self.ss, self.deltas = [search_range], [bin_len]
        self.r = theta_range * math.pi                          # rpn_model.py:118-119
        self.delta_theta = 2 * self.r / theta_bins
        self.cls = nn.Linear(cin, 1)
        self.reg = nn.Linear(cin, 4 * self.nb_xz + 2 * self.nb_theta + 4)
        self.register_buffer("mean_size", torch.tensor([list(mean_size)]))

    def forward(self, feats, ref_xyz, ref_theta=0):
        score = torch.sigmoid(self.cls(feats)).squeeze(-1)
        boxes = box_codec.decode_head(self.reg(feats), ref_xyz, ref_theta, self.mean_size, self.nb_xz, self.nb_xz,
                                      self.nb_theta, self.ss, self.deltas, self.r, self.delta_theta)
        return score, boxes.squeeze(-2)


class TwoStageDetector(nn.Module):
    def __init__(self, pre_nms_size=9000, rpn_nms_thresh=0.8, rpn_nms_size=100, roi_crop_size=512, context=1.0,
                 rcnn_nms_thresh=0.01, rcnn_nms_size=100, rpn_feat=128, img_channels=0):
        """img_channels > 0: the image branch's feature map (B,H,W,img_channels) and the P2 calibration are passed to
        forward(); the features under the projected points are concatenated to the point features for the RPN head
        and for the RCNN crop (the reference's "concat" fusion and its saved output_fts)."""
        super().__init__()
        self.img_channels = img_channels
        self.pre_nms_size, self.rpn_nms_thresh, self.rpn_nms_size = pre_nms_size, rpn_nms_thresh, rpn_nms_size
        self.roi_crop_size, self.context = roi_crop_size, context
        self.rcnn_nms_thresh, self.rcnn_nms_size = rcnn_nms_thresh, rcnn_nms_size
        self.rpn = modules.PointnetSAFPStack(in_channel=1)
        assert self.rpn.out_channel == rpn_feat
        self.rpn_head = BoxHead(rpn_feat + img_channels)
        # cropped features (point + image) + intensity + foreground mask (rcnn_model.py:178-182, 505-520)
        cin = rpn_feat + img_channels + 1 + 1
        self.rcnn_sa1 = modules.PointnetSAModule(128, 0.4, 16, cin, [128, 128])
        self.rcnn_sa2 = modules.PointnetSAModule(32, 0.8, 16, 128, [128, 256])
        self.rcnn_sa3 = modules.PointnetSAModule(None, None, None, 256, [256, 512], group_all=True)
        self.rcnn_head = BoxHead(512)

    @torch.no_grad()
    def rpn_stage(self, xyz, intensity, geometry=None, img_fts=None, calib=None):
        """geometry: self.rpn.geometry(xyz) computed ahead (pipeline.GeometryPrefetcher), or None to do it inline"""
        feats = self.rpn(xyz, intensity, geometry=geometry)                # (B,N,C)
        if self.img_channels:
            assert img_fts is not None and calib is not None and img_fts.shape[-1] == self.img_channels
            feats = fuse_point_image_features(feats, project_gather(xyz, calib, img_fts), "concat")
        scores, boxes = self.rpn_head(feats, xyz)                          # (B,N), (B,N,7)
        k = min(self.pre_nms_size, xyz.shape[1])
        top_s, top_i = torch.topk(scores, k, dim=1)                        # rpn_model.py:647-655, sorted descending
        top_b = torch.gather(boxes, 1, top_i.unsqueeze(-1).expand(-1, -1, 7))
        # tf.map_fn(sb_nms_fn) over the frames (rpn_model.py:683-687) as ONE batched device NMS; the tail of each
        # keep row repeats keep[0] (bev_iou.cpp:110-112), which is what fixed_num_proposal_nms relies on
        keep, _ = oriented_nms_batched(modules.boxes3d_to_bev(top_b).contiguous(), self.rpn_nms_thresh)
        ind = keep[:, :self.rpn_nms_size].long()
        proposals = torch.gather(top_b, 1, ind.unsqueeze(-1).expand(-1, -1, 7))
        return feats, proposals, torch.gather(top_s, 1, ind), scores

    @torch.no_grad()
    def rcnn_stage(self, xyz, feats, intensity, point_scores, proposals):
        b, n_prop, _ = proposals.shape
        flat = proposals.reshape(-1, 7)
        box_ind = torch.arange(b, device=xyz.device, dtype=torch.int32).repeat_interleave(n_prop)
        boxes8 = modules.box_3d_to_box_8co(expand_proposals(flat, self.context)).contiguous()
        fg_mask = point_scores > 0.5
        crop_pts, crop_fts, crop_int, crop_mask, _, non_empty = pc_crop_and_sample(
            xyz, feats.contiguous(), intensity, fg_mask, boxes8, box_ind, self.roi_crop_size)
        pts_ct = canonical_transform(crop_pts, flat).contiguous()
        pts_in = torch.cat([crop_fts, crop_int, crop_mask.unsqueeze(-1).float()], dim=-1)
        x1, f1, _ = self.rcnn_sa1(pts_ct, pts_in)
        x2, f2, _ = self.rcnn_sa2(x1, f1)
        _, f3, _ = self.rcnn_sa3(x2, f2)
        score, refined = self.rcnn_head(f3.squeeze(1), flat[:, 0:3].contiguous(), flat[:, 6].contiguous())
        score = score * non_empty.float()                                   # empty RoIs carry no evidence
        # model_util.py:101-142 / rcnn_model.py:731-778 for all frames at once: sort by score, ONE batched oriented
        # NMS, drop the keep[0] padding (= keep the first `num` entries), one host read of the counts
        score, refined = score.view(b, n_prop), refined.view(b, n_prop, 7)
        order = torch.sort(score, dim=1, descending=True, stable=True).indices
        bev = torch.gather(modules.boxes3d_to_bev(refined), 1, order.unsqueeze(-1).expand(-1, -1, 5))
        keep, num = oriented_nms_batched(bev.contiguous(), self.rcnn_nms_thresh)
        ind = torch.gather(order, 1, keep.long())
        counts = torch.clamp(num, max=self.rcnn_nms_size).tolist()
        detections = []
        for i in range(b):
            sel = ind[i, :counts[i]]
            detections.append({"boxes": refined[i, sel], "scores": score[i, sel]})
        return detections

    @torch.no_grad()
    def forward(self, xyz, intensity, geometry=None, img_fts=None, calib=None):
        feats, proposals, _, point_scores = self.rpn_stage(xyz, intensity, geometry, img_fts, calib)
        return self.rcnn_stage(xyz, feats, intensity, point_scores, proposals)


def run_sharded(model, frames_xyz, frames_intensity, ctx, frames_per_batch=8):
    """frames_* are lists of per-frame tensors (all ranks hold the same list, as a shared dataset would);
    each rank runs its rank-strided shard and rank 0 receives {frame_id: detections} from everyone."""
    mine = dp.shard_frames(len(frames_xyz), ctx.rank, ctx.world)
    batches = [mine[i:i + frames_per_batch] for i in range(0, len(mine), frames_per_batch)]

    def upload(ids):
        return (torch.stack([frames_xyz[j] for j in ids]).to(ctx.device),
                torch.stack([frames_intensity[j] for j in ids]).to(ctx.device))

    # on a GPU the sampling / grouping geometry of the RPN stack for batch i+1 runs on a side stream while batch i
    # goes through the two stages (FPS alone is ~5.6 ms of one CU per cloud)
    prefetch = None
    if ctx.device.type == "cuda" and batches:
        from .pipeline import GeometryPrefetcher
        prefetch = GeometryPrefetcher(model.rpn.geometry, device=ctx.device, depth=2)
    out, staged = {}, []
    for ids in batches[:2]:
        staged.append(upload(ids))
        if prefetch is not None:
            prefetch.submit(staged[-1][0])
    for bi, ids in enumerate(batches):
        xyz, inten = staged.pop(0)
        geo = prefetch.get() if prefetch is not None else None
        if bi + 2 < len(batches):
            staged.append(upload(batches[bi + 2]))
            if prefetch is not None:
                prefetch.submit(staged[-1][0])
        for j, det in zip(ids, model(xyz, inten, geometry=geo)):
            out[j] = {k: v.cpu() for k, v in det.items()}
    gathered = dp.gather_objects(out, ctx)
    if ctx.rank != 0:
        return None
    merged = {}
    for part in gathered:
        merged.update(part)
    return merged
