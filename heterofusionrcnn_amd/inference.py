"""KITTI files in, KITTI result files out: the on-disk formats (kitti_io.py) wired to the two-stage detector (SURVEY.md 8f rank 4,
BASELINE config 5 "Two-stage RPN -> tf_cropping -> RCNN inference on KITTI val, frames sharded across 8 GPUs").

  frame loading     velodyne + calib + image -> rect-frame cloud inside the image, 16384-point sample, image resized to
                    360 x 1200 and mean-subtracted          hf/datasets/kitti/kitti_dataset.py:300-410, img_feature_extractor.py:8-30
  image branch      ImgVggPyr: the VGG pyramid of hf/core/feature_extractors/img_vgg_pyramid.py:31-175 as a plain torch module
                    (a stock convolutional network on MIOpen, no custom op; OUTSIDE the hot path: bench.py's step takes its
                    output as an input).  Here so that the flow runs from files to files.
  detection         two_stage.TwoStageDetector (rank-strided frame shards, geometry prefetched one batch ahead)
  result writing    score threshold (eval_config kitti_score_threshold 0.1), 3-D boxes projected into the image, boxes that
                    leave the image or cover more than 80 % of it dropped, the rest truncated, KITTI label lines with alpha
                    -10, three decimals                      hf/core/evaluator_utils.py:18-166, box_3d_projector.py:88-163

Weights are random unless the caller loads its own (no checkpoint ships with the reference): the test checks the plumbing
(files -> tensors -> detector -> files, frame ids, sharding), not detection quality.
"""
import os

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import dp, kitti_io

IMAGE_MEAN = (92.8403, 97.7996, 93.5843)        # img_feature_extractor.py:9-11 (R, G, B)
CLASSES = ("Car", "Pedestrian", "Cyclist")      # rpn_multiclass.config dataset_config.classes


class _ConvBNReLU(nn.Sequential):
    """slim.conv2d(3x3, SAME, normalizer_fn=slim.batch_norm, activation relu)"""

    def __init__(self, cin, cout):
        super().__init__(nn.Conv2d(cin, cout, 3, padding=1, bias=False), nn.BatchNorm2d(cout, eps=1e-3, momentum=0.001), nn.ReLU(inplace=True))


class _UpConvBNReLU(nn.Sequential):
    """slim.conv2d_transpose(3x3, stride 2, SAME): doubles height and width"""

    def __init__(self, cin, cout):
        super().__init__(nn.ConvTranspose2d(cin, cout, 3, stride=2, padding=1, output_padding=1, bias=False),
                         nn.BatchNorm2d(cout, eps=1e-3, momentum=0.001), nn.ReLU(inplace=True))


class ImgVggPyr(nn.Module):
    """img_vgg_pyramid.py:31-175 with rpn_multiclass.config:120-128 (vgg_conv1 [2,32], conv2 [2,64], conv3 [3,128], conv4 [3,256]):
    encoder with three 2x2 max-pools, decoder of three (upconv, concat with the encoder level, 3x3 fusion conv) stages back
    to full resolution.  (B,H,W,3) image -> (B,H,W,32) feature map, channel-last on both sides like the reference."""

    def __init__(self, conv=((2, 32), (2, 64), (3, 128), (3, 256))):
        super().__init__()
        def block(cin, n, c):
            layers = []
            for _ in range(n):
                layers.append(_ConvBNReLU(cin, c))
                cin = c
            return nn.Sequential(*layers)
        c1, c2, c3, c4 = (c for _, c in conv)
        self.conv1, self.conv2 = block(3, conv[0][0], c1), block(c1, conv[1][0], c2)
        self.conv3, self.conv4 = block(c2, conv[2][0], c3), block(c3, conv[3][0], c4)
        self.upconv3, self.fusion3 = _UpConvBNReLU(c4, c3), _ConvBNReLU(2 * c3, c2)
        self.upconv2, self.fusion2 = _UpConvBNReLU(c2, c2), _ConvBNReLU(2 * c2, c1)
        self.upconv1, self.fusion1 = _UpConvBNReLU(c1, c1), _ConvBNReLU(2 * c1, c1)
        self.out_channel = c1

    def forward(self, image):
        x = image.permute(0, 3, 1, 2)
        x = x - torch.tensor(IMAGE_MEAN, device=x.device, dtype=x.dtype).view(1, 3, 1, 1)      # preprocess_input
        conv1 = self.conv1(x)
        conv2 = self.conv2(F.max_pool2d(conv1, 2))
        conv3 = self.conv3(F.max_pool2d(conv2, 2))
        conv4 = self.conv4(F.max_pool2d(conv3, 2))
        f3 = self.fusion3(torch.cat([conv3, self.upconv3(conv4)], dim=1))
        f2 = self.fusion2(torch.cat([conv2, self.upconv2(f3)], dim=1))
        f1 = self.fusion1(torch.cat([conv1, self.upconv1(f2)], dim=1))
        return f1.permute(0, 2, 3, 1).contiguous()


def load_kitti_frame(dataset_dir, name, rng, num_points=16384, img_hw=(360, 1200)):
    """One frame of <dataset_dir>/{velodyne,calib,image_2}/<name>.* -> dict of numpy arrays: xyz (P,3), intensity (P,1), image
    (h,w,3) float32 resized as the reference resizes it (kitti_dataset.py:398), calib (3,4) = P2 RESCALED WITH THE IMAGE as the
    reference does (kitti_dataset.py:399-400, 525-526: row 0 *= img_w / w0, row 1 *= img_h / h0) -- the matrix the detector
    projects with (project_gather, project_boxes_to_image, crop_and_resize all work in resized pixels); calib_orig (3,4) = P2 as
    read, for the result files (boxes are written in ORIGINAL-image pixels, evaluator_utils.py:88-166), image_size (w, h) of the
    original image"""
    from PIL import Image      # the one image reader of this package: only this file-level flow needs it
    calib = kitti_io.read_calib(os.path.join(dataset_dir, "calib", name + ".txt"))
    img = Image.open(os.path.join(dataset_dir, "image_2", name + ".png")).convert("RGB")
    w0, h0 = img.size
    pts, inten = kitti_io.load_point_cloud(os.path.join(dataset_dir, "velodyne", name + ".bin"), calib, (h0, w0))
    sample, _ = kitti_io.sample_points(pts, inten, num_points, rng)
    image = np.asarray(img.resize((img_hw[1], img_hw[0]), Image.BILINEAR), dtype=np.float32)
    p2 = calib["p2"].astype(np.float32)
    return {"name": name, "xyz": sample[:, :3].copy(), "intensity": sample[:, 3:4].copy(), "image": image,
            "calib": rescale_p2(p2, (w0, h0), (img_hw[1], img_hw[0])), "calib_orig": p2, "image_size": (w0, h0)}


def rescale_p2(p2, size_wh, new_size_wh):
    """P2 for an image resized from size_wh to new_size_wh: u' = u * w'/w, v' = v * h'/h (kitti_dataset.py:399-400)"""
    out = np.array(p2, dtype=np.float32, copy=True)
    out[0, :] *= np.float32(new_size_wh[0] / size_wh[0])
    out[1, :] *= np.float32(new_size_wh[1] / size_wh[1])
    return out


def project_box3d_to_image(box_3d, p2, image_size):
    """box_3d_projector.project_to_image_space(truncate=True, discard_before_truncation=True) (:88-163): the bounding rectangle of
    the projected corners; None when it lies outside the image or is wider / taller than 80 % of it; else truncated to the image"""
    x, y, z, l, w, h, ry = [float(v) for v in box_3d]
    c, s = np.cos(ry), np.sin(ry)
    xs = np.array([l / 2, l / 2, -l / 2, -l / 2, l / 2, l / 2, -l / 2, -l / 2])
    zs = np.array([w / 2, -w / 2, -w / 2, w / 2, w / 2, -w / 2, -w / 2, w / 2])
    ys = np.array([0, 0, 0, 0, -h, -h, -h, -h])
    corners = np.stack([c * xs + s * zs + x, ys + y, -s * xs + c * zs + z])            # obj_utils.compute_box_corners_3d
    uv = kitti_io.project_to_image(corners.T, p2)
    box = np.array([uv[:, 0].min(), uv[:, 1].min(), uv[:, 0].max(), uv[:, 1].max()])
    iw, ih = image_size
    if box[0] > iw or box[1] > ih or box[2] < 0 or box[3] < 0:
        return None
    if box[2] - box[0] > 0.8 * iw or box[3] - box[1] > 0.8 * ih:
        return None
    return np.array([max(box[0], 0), max(box[1], 0), min(box[2], iw), min(box[3], ih)])


def write_frame_results(path, det, p2, image_size, score_threshold=0.1, classes=CLASSES):
    """one KITTI result file from a frame's detections {boxes (n,7), scores (n), classes (n) in 1..K} (evaluator_utils.py:88-166)"""
    boxes, scores, cls = (np.asarray(det[k].cpu() if torch.is_tensor(det[k]) else det[k]) for k in ("boxes", "scores", "classes"))
    keep = scores >= round(score_threshold, 3)
    types, b2, b3, sc = [], [], [], []
    for bx, s_, c_ in zip(boxes[keep], scores[keep], cls[keep]):
        img_box = project_box3d_to_image(bx, p2, image_size)
        if img_box is None:
            continue
        types.append(classes[int(c_) - 1])
        b2.append(np.round(img_box, 3))
        b3.append(np.round(bx, 3))
        sc.append(round(float(s_), 3))
    kitti_io.write_kitti_results(path, types, np.asarray(b2).reshape(-1, 4), np.asarray(b3).reshape(-1, 7), sc)
    return len(types)


@torch.no_grad()
def run_kitti_inference(detector, img_net, dataset_dir, names, out_dir, ctx, frames_per_batch=8, seed=0, score_threshold=0.1):
    """The file-to-file flow of BASELINE config 5 on this rank's shard of `names` (rank-strided, dp.shard_frames): load, image
    branch, two stages, result files <out_dir>/<name>.txt.  Returns {name: number of boxes written} gathered on rank 0."""
    from .pipeline import GeometryPrefetcher
    os.makedirs(out_dir, exist_ok=True)
    mine = [names[i] for i in dp.shard_frames(len(names), ctx.rank, ctx.world)]
    batches = [mine[i:i + frames_per_batch] for i in range(0, len(mine), frames_per_batch)]
    rng = np.random.default_rng(seed + ctx.rank)

    def load(batch):
        frames = [load_kitti_frame(dataset_dir, n, rng, img_hw=detector.rcnn_cfg.img_hw) for n in batch]
        dev = {k: torch.from_numpy(np.stack([f[k] for f in frames])).to(ctx.device) for k in ("xyz", "intensity", "image", "calib")}
        return frames, dev

    prefetch = GeometryPrefetcher(detector.geometry, device=ctx.device, depth=2) if ctx.device.type == "cuda" and batches else None
    staged, written = [], {}
    for batch in batches[:2]:
        staged.append(load(batch))
        if prefetch is not None:
            prefetch.submit(staged[-1][1]["xyz"])
    for bi, batch in enumerate(batches):
        frames, dev = staged.pop(0)
        geo = prefetch.get() if prefetch is not None else None
        if bi + 2 < len(batches):
            staged.append(load(batches[bi + 2]))
            if prefetch is not None:
                prefetch.submit(staged[-1][1]["xyz"])
        img_fts = img_net(dev["image"])
        dets = detector(dev["xyz"], dev["intensity"], img_fts, dev["calib"], geometry=geo)
        for f, det in zip(frames, dets):
            written[f["name"]] = write_frame_results(os.path.join(out_dir, f["name"] + ".txt"), det, f["calib_orig"], f["image_size"],
                                                     score_threshold)
    gathered = dp.gather_objects(written, ctx)
    if ctx.rank != 0:
        return None
    merged = {}
    for part in gathered:
        merged.update(part)
    return merged
