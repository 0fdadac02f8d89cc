"""RPN train step on the HIP ops (BASELINE config 4): point backbone -> foreground segmentation head ->
bin-based box head -> targets -> focal + softmax + smooth-L1 losses, as hf/core/models/rpn_model.py builds it.

  backbone            hf/core/feature_extractors/pointnet.py:22-153 (SA / SA-MSG down, FP up, conv1d+BN+ReLU fc layers)
  seg head            rpn_model.py:455-476   dense(K+1), no BN, no activation; softmax
  fusion              rpn_model.py:515-548   'mean' / 'concat' with the image features under the projected points
  box head            rpn_model.py:552-581   dense(C)+ELU+BN, dropout, ... ; dense((2 NBX + 2 NBZ + 2 NBT + 4) K) + BN
  train mode          rpn_model.py:588-590   no decoding, no NMS
  targets             rpn_model.py:733-796   tf_encode around the point, per-class gathers, one-hot bins
  losses              rpn_model.py:1040-1128 + hf/core/losses.py:131-226
  per-point labels    hf/datasets/kitti/kitti_dataset.py:416-440 (point in box -> class + that box)

The network bodies are torch (the GEMMs go to hipBLASLt / the fused MFMA nodes of mlp.py); every sampling / grouping /
interpolation / encoding step is a HIP op of this package.  TensorFlow cannot be imported here, so the layer semantics
(`pf.dense` = linear -> ELU -> BatchNorm(momentum 0.99, eps 1e-3), pointfly.py:371-497) are restated from the text:
parity unpinned against reference outputs; the loss is pinned by a literal op-by-op restatement in tests/test_rpn.py.

The image branch (VGG pyramid, hf/core/feature_extractors/img_vgg_pyramid.py) is outside the path: the
model takes the image FEATURE MAP as an input and does the reference's projection + gather + fusion on it.
"""
import math
from dataclasses import dataclass, field
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import box_codec
from .fusion import fuse_point_image_features, path_drop_masks, project_gather
from .modules import PointnetFPModule, PointnetSAModule, SharedMLPLayer
from .grouping import group_concat, group_point, query_ball_group
from .sampling import farthest_point_sample, gather_point
from .mlp import BatchNormReLU, shared_mlp, linear_nobias, linear_narrow


# ------------------------------------------------------------------------------------------------ configuration
@dataclass
class SAScale:
    radius: float
    nsample: int
    mlp: Tuple[int, ...]


@dataclass
class SALevel:
    npoint: int
    scales: Tuple[SAScale, ...]          # one scale = pointnet_sa_module, several = pointnet_sa_module_msg


@dataclass
class RpnConfig:
    """The values of one reference config file that shape the RPN train step."""
    name: str
    num_classes: int = 1
    pc_sample_pts: int = 16384
    in_channel: int = 1                                                    # rpn_use_intensity_feature
    sa: Tuple[SALevel, ...] = ()
    fp: Tuple[Tuple[int, ...], ...] = ()                                   # deepest first, pointnet.py:108-128
    backbone_fc: Tuple[Tuple[int, float], ...] = ((256, 0.5), (256, 0.5))  # (C, dropout keep_prob), pointnet.py:131-151
    rpn_fc: Tuple[Tuple[int, float], ...] = ((512, 0.5), (512, 0.5))       # (C, dropout rate), rpn_model.py:556-568
    xz_search_range: Tuple[float, ...] = (3.0,)                            # Ss, one per class
    xz_bin_len: Tuple[float, ...] = (0.5,)                                 # DELTAs
    theta_search_range: float = 1.0                                        # fraction of pi
    theta_bin_num: int = 12
    cluster_sizes: Tuple[Tuple[float, float, float], ...] = ((3.9, 1.6, 1.5),)   # mean (l, w, h) per class
    backbone: str = "pointnet"                                             # 'pointnet' (sa / fp / backbone_fc) | 'pointcnn'
    pointcnn: Optional[object] = None                                      # pointcnn.PointCnnConfig for backbone 'pointcnn'
    fusion: str = "none"                                                   # 'mean' | 'concat' | 'none' (no image branch)
    img_channels: int = 0
    path_drop: Tuple[float, float] = (1.0, 1.0)                            # keep probabilities [image, points]; (1, 1) = off
    seg_loss_weight: float = 100.0
    cls_loss_weight: float = 1.0
    reg_loss_weight: float = 1.0

    @property
    def num_bin_xz(self):
        return int(2 * self.xz_search_range[0] / self.xz_bin_len[0])       # rpn_model.py:115-116

    @property
    def r_theta(self):
        return self.theta_search_range * math.pi                           # rpn_model.py:118

    @property
    def delta_theta(self):
        return 2 * self.r_theta / self.theta_bin_num                       # rpn_model.py:119

    @property
    def head_width(self):
        return 2 * self.num_bin_xz + 2 * self.num_bin_xz + 2 * self.theta_bin_num + 4


def rpn_cars_pointnet_paper() -> RpnConfig:
    """hf/configs/rpn_cars_pointnet_paper.config:61-152 (MSG set abstraction, 4 FP modules, cars only; its 'mean' fusion
    needs the image branch: here fusion='none' unless the caller supplies a feature map)."""
    s = SAScale
    return RpnConfig(
        name="rpn_cars_pointnet_paper",
        sa=(SALevel(4096, (s(0.1, 16, (16, 16, 32)), s(0.5, 32, (32, 32, 64)))),
            SALevel(1024, (s(0.5, 16, (64, 64, 128)), s(1.0, 32, (64, 96, 128)))),
            SALevel(512, (s(1.0, 16, (128, 196, 256)), s(2.0, 32, (128, 196, 256)))),
            SALevel(64, (s(2.0, 16, (256, 256, 512)), s(4.0, 32, (256, 384, 512))))),
        # the config lists each FP mlp with its INPUT width first (1536 = 1024 + 512, ...); only the outputs are layers
        fp=((512, 512), (512, 512), (256, 256), (128, 128)))


def rpn_stack_config2() -> RpnConfig:
    """The synthetic single-scale SA+FP stack of BASELINE.json configs[1] with the RPN heads on top."""
    s = SAScale
    return RpnConfig(
        name="sa_fp_stack_config2",
        sa=(SALevel(4096, (s(0.5, 32, (32, 32, 64)),)), SALevel(1024, (s(1.0, 32, (64, 96, 128)),)),
            SALevel(256, (s(2.0, 32, (128, 196, 256)),))),
        fp=((256, 256), (256, 256), (128, 128)))


def rpn_multiclass(img_channels: int = 0) -> RpnConfig:
    """hf/configs/rpn_multiclass.config: the PointCNN extractor (:62-118), three classes with their own search ranges
    (:20-36), 'concat' fusion (:21) with the image branch's features and path drop 0.9 / 0.9 (:57) when a feature map is
    supplied (img_channels > 0; the VGG pyramid of :120-128 ends in vgg_conv1[1] = 32 channels at full resolution)."""
    from .pointcnn import PointCnnConfig
    import dataclasses
    base = RpnConfig(name="rpn_multiclass", backbone="pointcnn", pointcnn=PointCnnConfig(),
                     fusion="concat" if img_channels else "none", img_channels=img_channels,
                     path_drop=(0.9, 0.9) if img_channels else (1.0, 1.0))
    heads = rpn_multiclass_heads(base)
    return dataclasses.replace(heads, name="rpn_multiclass")


def rpn_multiclass_heads(base: RpnConfig) -> RpnConfig:
    """The head / loss / class settings of hf/configs/rpn_multiclass.config:20-36,193-198,256-258 on a given backbone
    (Car, Pedestrian, Cyclist; per-class search ranges; 'concat' fusion)."""
    import dataclasses
    return dataclasses.replace(base, name=base.name + "+multiclass_heads", num_classes=3,
                               xz_search_range=(3.0, 1.5, 1.5), xz_bin_len=(0.5, 0.25, 0.25),
                               cluster_sizes=((3.88, 1.63, 1.53), (0.84, 0.66, 1.76), (1.76, 0.6, 1.73)))


# ------------------------------------------------------------------------------------------------ backbone
class _SALevelModule(nn.Module):
    """One set-abstraction level: FPS once, then every scale's ball query + shared MLP + max (pointnet_util.py:104-286).
    Single-scale levels concatenate [xyz, features] (:58-60), multi-scale levels [features, xyz] (:264)."""

    def __init__(self, level: SALevel, in_channel: int):
        super().__init__()
        self.npoint = level.npoint
        self.scales = level.scales
        self.msg = len(level.scales) > 1
        self.mlps = nn.ModuleList()
        cout = 0
        for sc in level.scales:
            layers, cin = [], in_channel + 3
            for w in sc.mlp:
                layers.append(SharedMLPLayer(cin, w))
                cin = w
            self.mlps.append(nn.Sequential(*layers))
            cout += cin
        self.out_channel = cout

    def geometry(self, xyz):
        with torch.no_grad():
            new_xyz = gather_point(xyz, farthest_point_sample(self.npoint, xyz))
            per_scale = []
            for sc in self.scales:
                idx, _, grouped_xyz = query_ball_group(sc.radius, sc.nsample, xyz, new_xyz, center=True)
                per_scale.append((idx, grouped_xyz))
        return new_xyz, per_scale

    def forward(self, xyz, points, geom=None):
        new_xyz, per_scale = geom if geom is not None else self.geometry(xyz)
        outs = []
        from . import modules as _modules
        from .mlp import GATHER_MIN_CFEAT, grouped_mlp_fusable, shared_mlp_grouped
        for (idx, grouped_xyz), mlp in zip(per_scale, self.mlps):
            live = all(l.bn.training for l in mlp) and torch.is_grad_enabled() or not any(l.bn.training for l in mlp)
            if _modules.GATHER_ON_LOAD and xyz.is_cuda and live and grouped_mlp_fusable(list(mlp), points, idx, GATHER_MIN_CFEAT):
                # the first layer reads the neighbourhoods in place: no (B,M,K,C+3) tensor
                outs.append(shared_mlp_grouped(list(mlp), points, idx, grouped_xyz, xyz_first=not self.msg).reshape(idx.shape[0], idx.shape[1], -1))
                continue
            grouped = group_concat(points, idx, grouped_xyz, xyz_last=self.msg)    # one pass, rows padded to 4 columns
            bsz, npt, k, cin = grouped.shape
            outs.append(shared_mlp(mlp, grouped.reshape(-1, cin), pool_k=k).reshape(bsz, npt, -1))
        return new_xyz, outs[0] if len(outs) == 1 else torch.cat(outs, dim=-1)


class PointnetBackbone(nn.Module):
    """hf/core/feature_extractors/pointnet.py: SA levels down, FP levels up (the 'original structure' wiring,
    :108-128), then conv1d + BN + ReLU layers with dropout between them (:131-151)."""

    def __init__(self, cfg: RpnConfig):
        super().__init__()
        self.sa = nn.ModuleList()
        chans = [cfg.in_channel]
        for level in cfg.sa:
            m = _SALevelModule(level, chans[-1])
            self.sa.append(m)
            chans.append(m.out_channel)
        self.fp = nn.ModuleList()
        c = chans[-1]
        for i, widths in enumerate(cfg.fp):                    # deepest first
            skip = chans[len(cfg.sa) - 1 - i]
            m = PointnetFPModule(c + skip, list(widths))
            self.fp.append(m)
            c = m.out_channel
        fcs = []
        self.fc_keep = []
        for (w, keep) in cfg.backbone_fc:
            fcs.append(SharedMLPLayer(c, w))
            self.fc_keep.append(keep)
            c = w
        self.fc = nn.ModuleList(fcs)
        self.out_channel = c

    def geometry(self, xyz):
        """everything that depends on the coordinates only (runs ahead of the features: pipeline.GeometryPrefetcher)"""
        sa, xyzs = [], [xyz]
        for m in self.sa:
            g = m.geometry(xyzs[-1])
            sa.append(g)
            xyzs.append(g[0])
        fp = []
        for i in range(len(self.fp)):
            d = len(self.sa) - 1 - i
            fp.append(PointnetFPModule.geometry(xyzs[d], xyzs[d + 1]))
        return {"sa": sa, "fp": fp}

    def forward(self, xyz, points, geometry=None):
        geometry = geometry if geometry is not None else self.geometry(xyz)
        xyzs, feats = [xyz], [points]
        for li, m in enumerate(self.sa):
            nx, nf = m(xyzs[-1], feats[-1], geometry["sa"][li])
            xyzs.append(nx)
            feats.append(nf)
        up = feats[-1]
        for i, m in enumerate(self.fp):
            d = len(self.sa) - 1 - i
            up = m(xyzs[d], xyzs[d + 1], feats[d], up, geometry["fp"][i])
        b, n, c = up.shape
        x = up.reshape(-1, c)
        for li, layer in enumerate(self.fc):
            x = layer(x)
            if li != len(self.fc) - 1:
                x = F.dropout(x, p=1.0 - self.fc_keep[li], training=self.training)   # tf_util.dropout(keep_prob), :146-151
        return x.reshape(b, n, -1)


# ------------------------------------------------------------------------------------------------ heads
class DenseEluBN(nn.Module):
    """pointfly.dense (pointfly.py:480-497): linear (no bias under BN) -> activation -> tf.layers.batch_normalization
    (momentum 0.99, epsilon 1e-3, pointfly.py:371-380).  activation=None for the output layer (rpn_model.py:571-578)."""

    def __init__(self, cin, cout, activation=True):
        super().__init__()
        self.linear = nn.Linear(cin, cout, bias=False)
        nn.init.xavier_normal_(self.linear.weight)                 # glorot_normal_initializer
        # the HIP statistics / apply / backward passes, with the ELU applied on load inside them
        self.bn = BatchNormReLU(cout, eps=1e-3, momentum=0.01, relu=False, elu_in=activation)
        self.activation = activation

    def forward(self, x, dropout=0.0):
        """dropout: the rate of the tf.layers.dropout that follows this layer (training mode: fused into the BatchNorm passes)"""
        return self.bn(linear_nobias(x, self.linear.weight), dropout)


class RpnHeads(nn.Module):
    def __init__(self, cfg: RpnConfig, cin: int):
        super().__init__()
        self.cfg = cfg
        k = cfg.num_classes
        self.seg = nn.Linear(cin, k + 1)                           # with_bn=False, activation=None: plain dense with bias
        nn.init.xavier_normal_(self.seg.weight)
        nn.init.zeros_(self.seg.bias)
        c = cin if cfg.fusion != "concat" else cin + cfg.img_channels
        layers = []
        self.drop = []
        for (w, rate) in cfg.rpn_fc:
            layers.append(DenseEluBN(c, w))
            self.drop.append(rate)
            c = w
        self.fc = nn.ModuleList(layers)
        self.out = DenseEluBN(c, cfg.head_width * k, activation=False)

    def forward(self, pc_fts, proj_img_fts=None, return_fused=False):
        b, p, c = pc_fts.shape
        seg_logits = linear_narrow(pc_fts, self.seg.weight, self.seg.bias)   # (B,P,K+1)
        x = pc_fts
        if self.cfg.fusion != "none":
            masks = None
            if self.training and tuple(self.cfg.path_drop) != (1.0, 1.0):      # rpn_model.py:515-535
                masks = path_drop_masks(self.cfg.path_drop[0], self.cfg.path_drop[1], torch.rand(3, device=pc_fts.device))
            x = fuse_point_image_features(pc_fts, proj_img_fts, self.cfg.fusion, masks=masks)
        x = x.reshape(b * p, -1)
        fused = x
        for layer, rate in zip(self.fc, self.drop):
            x = layer(x, dropout=rate if self.training else 0.0)       # dense -> tf.layers.dropout(rate), fused into the BatchNorm
        out = self.out(x).reshape(b, p, self.cfg.num_classes, self.cfg.head_width)
        return (seg_logits, out, fused.reshape(b, p, -1)) if return_fused else (seg_logits, out)


def parse_rpn_output(out, nbx, nbz, nbt):
    """rpn_model.py:870-935: (B,P,K,D) -> bin_x logits, res_x, bin_z logits, res_z, bin_theta logits, res_theta, res_y, res_size"""
    o = 0
    parts = []
    for w in (nbx, nbx, nbz, nbz, nbt, nbt, 1, 3):
        parts.append(out[..., o:o + w])
        o += w
    parts[6] = parts[6].squeeze(-1)
    return parts


# ------------------------------------------------------------------------------------------------ labels and targets
def point_labels(xyz, boxes_3d, box_cls):
    """Per-point class and box (the RPN's label_segs / label_regs): a point takes the FIRST ground-truth box that contains
    it (kitti_dataset.py:416-440 loops over the objects and overwrites nothing once set), class 0 = background.
    xyz (B,P,3); boxes_3d (B,G,7) [x,y,z,l,w,h,ry] with y the bottom (camera y down); box_cls (B,G) in 1..K, 0 = padding."""
    rel = xyz.unsqueeze(2) - boxes_3d[:, None, :, 0:3]                       # (B,P,G,3)
    c, s = torch.cos(boxes_3d[..., 6])[:, None], torch.sin(boxes_3d[..., 6])[:, None]
    lx = c * rel[..., 0] - s * rel[..., 2]                                   # rotate by -ry about y into the box frame
    lz = s * rel[..., 0] + c * rel[..., 2]
    l, w, h = boxes_3d[:, None, :, 3], boxes_3d[:, None, :, 4], boxes_3d[:, None, :, 5]
    inside = (lx.abs() <= l / 2) & (lz.abs() <= w / 2) & (rel[..., 1] <= 0) & (rel[..., 1] >= -h) & (box_cls[:, None, :] > 0)
    first = torch.argmax(inside.to(torch.int8), dim=2)                       # first True (0 when none)
    any_in = inside.any(dim=2)
    label_cls = torch.where(any_in, torch.gather(box_cls, 1, first), torch.zeros_like(first))
    label_reg = torch.gather(boxes_3d, 1, first.unsqueeze(-1).expand(-1, -1, 7)) * any_in.unsqueeze(-1)
    return label_cls, label_reg


def rpn_targets(cfg: RpnConfig, xyz, label_cls, label_reg):
    """rpn_model.py:733-776: mean sizes of the labelled class, tf_encode around the point (one HIP kernel), then the row of
    the labelled class.  Background points get class index 0: they are masked out of every box loss."""
    k = cfg.num_classes
    cls0 = torch.clamp(label_cls.long() - 1, min=0)                          # (B,P)
    sizes = box_codec.const_f32(xyz.device, cfg.cluster_sizes)   # uploaded once: no host-to-device copy inside the step
    mean_sizes = sizes[cls0]                                                 # (B,P,3)
    bin_x, res_x, bin_z, res_z, bin_t, res_t, res_y, res_size = box_codec.encode(
        xyz, 0, label_reg, mean_sizes, cfg.xz_search_range, cfg.xz_bin_len, cfg.r_theta, cfg.delta_theta, k)
    pick = lambda t: torch.gather(t, 2, cls0.unsqueeze(-1)).squeeze(-1)
    return {"cls0": cls0, "bin_x": pick(bin_x).long(), "res_x": pick(res_x), "bin_z": pick(bin_z).long(), "res_z": pick(res_z),
            "bin_theta": bin_t.long(), "res_theta": res_t, "res_y": res_y, "res_size": res_size}


def rpn_loss(cfg: RpnConfig, seg_logits, head, label_cls, targets):
    """rpn_model.py:1040-1128 with hf/core/losses.py: focal loss on the segmentation softmax (alpha 0.25, gamma 2, summed,
    x seg_loss_weight, / (B P)); softmax cross-entropy of the x / z / theta bins and smooth-L1 of the five residual groups
    over the foreground points (summed, / number of foreground points; zero when there is none).  No boolean_mask: the
    foreground mask multiplies, so shapes are static and nothing synchronises with the host."""
    b, p, k1 = seg_logits.shape
    nbx, nbt = cfg.num_bin_xz, cfg.theta_bin_num
    fg = (label_cls > 0)
    fgf = fg.to(seg_logits.dtype)
    # segmentation: -alpha (1 - p_t)^gamma log(p_t) on the true class, p clipped to [1e-7, 1 - 1e-7]
    prob = torch.softmax(seg_logits, dim=-1)
    pt = torch.gather(prob, 2, label_cls.long().unsqueeze(-1)).squeeze(-1).clamp(1e-7, 1.0 - 1e-7)
    seg = (0.25 * (1.0 - pt) ** 2 * (-torch.log(pt))).sum() * cfg.seg_loss_weight / float(b * p)
    # the labelled class's row of the head
    cls0 = targets["cls0"]
    row = torch.gather(head, 2, cls0[:, :, None, None].expand(-1, -1, 1, head.shape[-1])).squeeze(2)   # (B,P,D)
    bx, rx, bz, rz, bt, rt, ry, rs = parse_rpn_output(row, nbx, nbx, nbt)
    nfg = fgf.sum()
    denom = torch.clamp(nfg, min=1.0)

    def ce(logits, target):
        return (F.cross_entropy(logits.reshape(b * p, -1), target.reshape(-1), reduction="none") * fgf.reshape(-1)).sum()

    cls = (ce(bx, targets["bin_x"]) + ce(bz, targets["bin_z"]) + ce(bt, targets["bin_theta"])) * cfg.cls_loss_weight / denom
    take = lambda res, bins: torch.gather(res, 2, bins.unsqueeze(-1)).squeeze(-1)        # residual of the TRUE bin (:778-786)

    def sl1(pred, target):
        d = (pred - target).abs()
        v = torch.where(d < 1, 0.5 * d * d, d - 0.5)
        if v.dim() == 3:
            v = v.sum(-1)
        return (v * fgf).sum()

    reg = (sl1(take(rx, targets["bin_x"]), targets["res_x"]) + sl1(take(rz, targets["bin_z"]), targets["res_z"]) +
           sl1(take(rt, targets["bin_theta"]), targets["res_theta"]) + sl1(ry, targets["res_y"]) +
           sl1(rs, targets["res_size"])) * cfg.reg_loss_weight / denom
    return seg + cls + reg, {"segmentation": seg.detach(), "bin_classification": cls.detach(), "regression": reg.detach(),
                             "num_foreground": nfg.detach()}


class _RpnLossFused(torch.autograd.Function):
    """hf_rpn_loss_fwd / hf_rpn_loss_bwd: targets picked for the labelled class, focal + cross-entropy + smooth-L1 terms and
    their gradients in two passes over the head (the op-by-op form above is ~80 framework kernels and a dozen passes).
    Returns [segmentation, bin classification, regression, #foreground, total]; only the total carries a gradient."""

    @staticmethod
    def forward(ctx, seg_logits, head, label, enc, k, nbx, nbt, weights):
        from . import _lib
        from ._lib import check, ptr, stream_ptr
        L = _lib.lib()
        rows = label.numel()
        out5 = torch.empty((5,), dtype=torch.float32, device=head.device)
        nbytes = L.hf_rpn_loss_workspace()
        ws = torch.empty((nbytes // 4,), dtype=torch.float32, device=head.device)
        seg_logits, head = seg_logits.contiguous(), head.contiguous()
        args = [ptr(t) for t in (seg_logits, head, label) + tuple(enc)]
        check(L.hf_rpn_loss_fwd(rows, k, nbx, nbt, *args, weights[0], weights[1], weights[2], ptr(out5), ptr(ws), nbytes, stream_ptr()),
              "rpn_loss_fwd")
        ctx.save_for_backward(seg_logits, head, label, out5, *enc)
        ctx.meta = (rows, k, nbx, nbt, weights)
        return out5

    @staticmethod
    def backward(ctx, g):
        from . import _lib
        from ._lib import check, ptr, stream_ptr
        seg_logits, head, label, out5 = ctx.saved_tensors[:4]
        enc = ctx.saved_tensors[4:]
        rows, k, nbx, nbt, weights = ctx.meta
        up = g[4:5].contiguous()
        grad_seg, grad_head = torch.empty_like(seg_logits), torch.empty_like(head)
        args = [ptr(t) for t in (seg_logits, head, label) + tuple(enc)]
        check(_lib.lib().hf_rpn_loss_bwd(rows, k, nbx, nbt, *args, weights[0], weights[1], weights[2], ptr(out5), ptr(up), ptr(grad_seg),
                                         ptr(grad_head), stream_ptr()), "rpn_loss_bwd")
        return grad_seg, grad_head, None, None, None, None, None, None


def rpn_loss_fused(cfg: RpnConfig, xyz, seg_logits, head, label_cls, label_reg):
    """the same loss as rpn_targets + rpn_loss through the two HIP passes; -> (loss, parts)"""
    k = cfg.num_classes
    lab = label_cls.to(torch.int32).contiguous()
    cls0 = torch.clamp(label_cls.long() - 1, min=0)
    mean_sizes = box_codec.const_f32(xyz.device, cfg.cluster_sizes)[cls0]
    enc = box_codec.encode(xyz, 0, label_reg, mean_sizes, cfg.xz_search_range, cfg.xz_bin_len, cfg.r_theta, cfg.delta_theta, k)
    bin_x, res_x, bin_z, res_z, bin_t, res_t, res_y, res_size = enc
    out5 = _RpnLossFused.apply(seg_logits, head, lab, (bin_x, res_x, bin_z, res_z, bin_t, res_t, res_y, res_size), k, cfg.num_bin_xz,
                               cfg.theta_bin_num, (float(cfg.seg_loss_weight), float(cfg.cls_loss_weight), float(cfg.reg_loss_weight)))
    d = out5.detach()
    return out5[4], {"segmentation": d[0], "bin_classification": d[1], "regression": d[2], "num_foreground": d[3]}


# ------------------------------------------------------------------------------------------------ model
class RpnModel(nn.Module):
    def __init__(self, cfg: RpnConfig):
        super().__init__()
        self.cfg = cfg
        if cfg.backbone == "pointcnn":
            from .pointcnn import PointCnnBackbone
            self.backbone = PointCnnBackbone(cfg.pointcnn)
        else:
            self.backbone = PointnetBackbone(cfg)
        self.heads = RpnHeads(cfg, self.backbone.out_channel)

    def geometry(self, xyz):
        return self.backbone.geometry(xyz)

    def forward(self, xyz, intensity, geometry=None, img_fts=None, calib=None, taps=None):
        """xyz (B,P,3), intensity (B,P,1) -> seg logits (B,P,K+1), head (B,P,K,D).  img_fts (B,H,W,C) + calib (B,3,4) feed
        the fusion when the config has one.  taps (a list, PointCNN backbone only): receives (output, detached copy) of every encoder
        layer -- the tensors that separate the encoder from everything after it (graph_step.TrainStep cuts the backward pass there)."""
        pc_fts = self.backbone(xyz, intensity, geometry, taps=taps) if taps is not None else self.backbone(xyz, intensity, geometry)
        proj = None
        if self.cfg.fusion != "none":
            proj = project_gather(xyz, calib, img_fts)
        return self.heads(pc_fts, proj)

    @torch.no_grad()
    def propose(self, xyz, intensity, geometry=None, img_fts=None, calib=None, pre_nms_size=9000, nms_thresh=0.8, post_nms_size=100):
        """The RPN in test mode (rpn_model.py:455-476, 593-700, 845-866): segmentation softmax -> per-point score (the best
        foreground class) and class, the bin-based head decoded around every point for its predicted class, the
        pre_nms_size best points, oriented BEV NMS, post_nms_size proposals per frame (fixed_num_proposal_nms: the keep
        list is padded with its first entry, bev_iou.cpp:110-112).  Also returns what the reference saves for the second
        stage: the fused per-point features, the foreground mask and the point scores."""
        from .bev_iou import oriented_nms_batched
        from .modules import boxes3d_to_bev
        cfg = self.cfg
        pc_fts = self.backbone(xyz, intensity, geometry)
        proj = project_gather(xyz, calib, img_fts) if cfg.fusion != "none" else None
        seg_logits, head, fused = self.heads(pc_fts, proj, return_fused=True)
        prob = torch.softmax(seg_logits, dim=-1)
        scores, fg_cls = prob[..., 1:].max(dim=-1)                              # seg_scores, seg_fg_preds
        fg_mask = prob.argmax(dim=-1) > 0                                       # seg_preds > 0 (test mode, :505-507)
        boxes = box_codec.decode_head(head, xyz, 0, cfg.cluster_sizes, cfg.num_bin_xz, cfg.num_bin_xz, cfg.theta_bin_num,
                                      cfg.xz_search_range, cfg.xz_bin_len, cfg.r_theta, cfg.delta_theta, cls=fg_cls)
        k = min(pre_nms_size, xyz.shape[1])
        top_s, top_i = torch.topk(scores, k, dim=1)                             # tf.nn.top_k(sorted=True), :647-655
        top_b = torch.gather(boxes, 1, top_i.unsqueeze(-1).expand(-1, -1, 7))
        keep, num = oriented_nms_batched(boxes3d_to_bev(top_b).contiguous(), nms_thresh)
        ind = keep[:, :post_nms_size].long()
        return {"proposals": torch.gather(top_b, 1, ind.unsqueeze(-1).expand(-1, -1, 7)), "proposal_scores": torch.gather(top_s, 1, ind),
                "num_before_padding": torch.clamp(num, max=post_nms_size), "rpn_fts": fused, "fg_mask": fg_mask, "point_scores": scores,
                "pre_nms_boxes": top_b, "pre_nms_scores": top_s, "keep": keep, "num_kept": num}

    def loss(self, xyz, seg_logits, head, label_cls, label_reg, fused=None):
        """fused (default: on the device in fp32): the two-pass HIP loss; else the op-by-op torch form it is tested against"""
        if fused is None:
            fused = head.is_cuda and head.dtype == torch.float32 and self.cfg.num_classes + 1 <= 8 and self.cfg.num_bin_xz <= 32 \
                and self.cfg.theta_bin_num <= 32
        if fused:
            return rpn_loss_fused(self.cfg, xyz, seg_logits, head, label_cls, label_reg)
        with torch.no_grad():
            targets = rpn_targets(self.cfg, xyz, label_cls, label_reg)
        return rpn_loss(self.cfg, seg_logits, head, label_cls, targets)


class RpnWithImageBranch(nn.Module):
    """The whole rpn_multiclass step of hf/core/models/rpn_model.py:126-127,223: the VGG pyramid (inference.ImgVggPyr, a stock
    convolutional network on the vendor library -- no custom op, outside SURVEY 8's kernel scope) in front of the fusion, trained with
    the RPN.  `img_fts` of forward() is then the IMAGE (B,H,W,3); everything else is RpnModel's interface."""

    def __init__(self, rpn: "RpnModel", img_net: nn.Module):
        super().__init__()
        self.rpn, self.img_net = rpn, img_net
        self.cfg = rpn.cfg

    @property
    def backbone(self):
        return self.rpn.backbone

    def geometry(self, xyz):
        return self.rpn.geometry(xyz)

    def forward(self, xyz, intensity, geometry=None, img_fts=None, calib=None, **kw):
        return self.rpn(xyz, intensity, geometry=geometry, img_fts=self.img_net(img_fts), calib=calib, **kw)

    def loss(self, *args, **kw):
        return self.rpn.loss(*args, **kw)


def synthetic_ground_truth(rng, batch, boxes_per_frame, cfg: RpnConfig, extent=((-40.0, 40.0), (0.0, 70.0)), ground_y=1.6):
    """(B,G,7) boxes and (B,G) classes in the KITTI camera frame: bottoms on the ground plane y = ground_y (camera y points
    down, the LiDAR sits ~1.6 m above the road), sizes around the class means, any heading."""
    k = cfg.num_classes
    cls = rng.integers(1, k + 1, (batch, boxes_per_frame))
    mean = np.asarray(cfg.cluster_sizes, np.float32)[cls - 1]
    size = np.clip(mean * (1.0 + 0.1 * rng.standard_normal(mean.shape)), 0.3, None)
    x = rng.uniform(extent[0][0], extent[0][1], (batch, boxes_per_frame))
    z = rng.uniform(extent[1][0], extent[1][1], (batch, boxes_per_frame))
    y = np.full_like(x, ground_y)
    ry = rng.uniform(-np.pi, np.pi, (batch, boxes_per_frame))
    boxes = np.concatenate([np.stack([x, y, z], -1), size, ry[..., None]], -1).astype(np.float32)
    return boxes, cls.astype(np.int64)
