"""Two-stage inference (BASELINE config 5) on the HIP ops: the RPN of rpn_multiclass.config hands its proposals, fused
per-point features, foreground mask and intensities to the RCNN of rcnn_multiclass.config.

  RPN in test mode -> scores, classes, decoded boxes      rpn.RpnModel.propose       hf/core/models/rpn_model.py:455-700
  top-k 9000, oriented NMS 0.8, 100 proposals per frame   (one batched device NMS)    rpn_model.py:645-700, model_util.py:101-142
  what the first stage saves for the second               rpn_fts, fg_mask, intensity rpn_model.py:845-866 (SAVE_RPN_*)
  RoI pooling, local features, PointCNN, heads            rcnn.RcnnModel.forward      hf/core/models/rcnn_model.py:402-664
  decode, drop empty RoIs, oriented NMS 0.01, 100 boxes   rcnn.RcnnModel.detect       rcnn_model.py:670-778

The reference runs the two stages as two programs with the hand-off on disk (hf/experiments/run_inference.py, kitti_io has
the file formats); here they are chained on the device.  Frames are independent: `run_sharded` splits them rank-strided over
the process group and rank 0 collects the per-frame detections (no tensor collective).  Weights are random in tests and the
bench (no checkpoint ships with the reference); the image branch's VGG pyramid is not run -- its feature map is an input.
"""
import torch
import torch.nn as nn

from . import dp
from .rcnn import RcnnConfig, RcnnModel, canonical_transform, expand_proposals  # noqa: F401  (re-exported)
from .rpn import RpnModel, rpn_multiclass


class TwoStageDetector(nn.Module):
    def __init__(self, rpn_cfg=None, rcnn_cfg=None, pre_nms_size=9000, rpn_nms_thresh=0.8, rpn_post_nms_size=100):
        """defaults = rpn_multiclass.config:27-29 (test settings) + rcnn_multiclass.config"""
        super().__init__()
        self.rpn_cfg = rpn_cfg or rpn_multiclass(32)
        self.rcnn_cfg = rcnn_cfg or RcnnConfig(img_channels=self.rpn_cfg.img_channels)
        self.pre_nms_size, self.rpn_nms_thresh, self.rpn_post_nms_size = pre_nms_size, rpn_nms_thresh, rpn_post_nms_size
        self.rpn = RpnModel(self.rpn_cfg)
        self.rcnn = RcnnModel(self.rcnn_cfg)
        assert self.rcnn_cfg.rpn_fts_channels == self.rpn.backbone.out_channel + self.rpn_cfg.img_channels

    def geometry(self, xyz):
        """the coordinate-only half of the RPN (sampling, neighbour tables): pipeline.GeometryPrefetcher runs it ahead"""
        return self.rpn.geometry(xyz)

    @torch.no_grad()
    def rpn_stage(self, xyz, intensity, img_fts, calib, geometry=None):
        return self.rpn.propose(xyz, intensity, geometry, img_fts, calib, self.pre_nms_size, self.rpn_nms_thresh,
                                self.rpn_post_nms_size)

    @torch.no_grad()
    def rcnn_stage(self, xyz, intensity, img_fts, calib, rpn_out):
        return self.rcnn.detect(xyz, rpn_out["rpn_fts"], intensity, rpn_out["fg_mask"], rpn_out["proposals"], img_fts, calib)

    @torch.no_grad()
    def forward(self, xyz, intensity, img_fts, calib, geometry=None, return_debug=False):
        rpn_out = self.rpn_stage(xyz, intensity, img_fts, calib, geometry)
        dets, dbg = self.rcnn_stage(xyz, intensity, img_fts, calib, rpn_out)
        return (dets, {"rpn": rpn_out, "rcnn": dbg}) if return_debug else dets


def run_sharded(model, frames, ctx, frames_per_batch=8):
    """frames: list of dicts {xyz (P,3), intensity (P,1), img_fts (H,W,C), calib (3,4)} (all ranks hold the same list, as a
    shared dataset would); each rank runs its rank-strided shard and rank 0 receives {frame_id: detections} from everyone."""
    mine = dp.shard_frames(len(frames), ctx.rank, ctx.world)
    batches = [mine[i:i + frames_per_batch] for i in range(0, len(mine), frames_per_batch)]

    def upload(ids):
        return {k: torch.stack([frames[j][k] for j in ids]).to(ctx.device) for k in ("xyz", "intensity", "img_fts", "calib")}

    # on a GPU the sampling / neighbour geometry of the RPN for batch i+1 runs on a side stream while batch i goes through
    # the two stages (FPS alone is ~5 ms of one CU per cloud)
    prefetch = None
    if ctx.device.type == "cuda" and batches:
        from .pipeline import GeometryPrefetcher
        prefetch = GeometryPrefetcher(model.geometry, device=ctx.device, depth=2)
    out, staged = {}, []
    for ids in batches[:2]:
        staged.append(upload(ids))
        if prefetch is not None:
            prefetch.submit(staged[-1]["xyz"])
    for bi, ids in enumerate(batches):
        cur = staged.pop(0)
        geo = prefetch.get() if prefetch is not None else None
        if bi + 2 < len(batches):
            staged.append(upload(batches[bi + 2]))
            if prefetch is not None:
                prefetch.submit(staged[-1]["xyz"])
        for j, det in zip(ids, model(cur["xyz"], cur["intensity"], cur["img_fts"], cur["calib"], geometry=geo)):
            out[j] = {k: v.cpu() for k, v in det.items()}
    gathered = dp.gather_objects(out, ctx)
    if ctx.rank != 0:
        return None
    merged = {}
    for part in gathered:
        merged.update(part)
    return merged
