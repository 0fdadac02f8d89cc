"""Host-side composition of the ops: the callers on either side of the hot path.

Restates, in torch on top of the HIP ops, the Python glue of the reference that defines HOW the
ops are chained:

  sample_and_group, PointnetSAModule, PointnetSAModuleMSG, PointnetFPModule
        hf/core/feature_extractors/pointnet_util.py:24-66,104-220,223-286,289-330
  boxes3d_to_bev, box3d_iou, oriented_nms_3d, sb_nms
        hf/core/compute_iou.py:7-80, hf/core/models/model_util.py:101-142
  box_3d_to_box_8co
        hf/core/box_8c_encoder.py:101-185

The MLPs between the ops are 1x1 convolutions in the reference (tf_util.conv2d: xavier weights,
zero bias, batch norm with decay 0.9 / epsilon 1e-3, ReLU; tf_util.py:127-203,554-581); on
channel-last tensors that is a Linear over the last dimension, which torch runs as a rocBLAS /
hipBLASLt GEMM.  Tensors stay channel-last (B, M, K, C) end to end, the layout the HIP gather
kernels read and write coalesced.
"""
import torch
import torch.nn as nn

from .grouping import group_concat, group_point, knn_point, query_ball_group, query_ball_point
from .interpolate import (INVERSE_MAX_KNOWN, three_interpolate, three_interpolate_concat, three_nn,
                          three_nn_inverse)
from .sampling import farthest_point_sample, gather_point
from .mlp import GATHER_MIN_CFEAT, BatchNormReLU, grouped_mlp_fusable, linear_bn_relu, shared_mlp, shared_mlp_grouped

# the first layer of a set-abstraction MLP gathers its operand in place (mlp.shared_mlp_grouped); False keeps the
# materialised group_concat route (tests compare the two)
GATHER_ON_LOAD = True
from .bev_iou import compute_bev_iou as _compute_bev_iou, oriented_nms as _oriented_nms


class _TallSkinnyLinear(torch.autograd.Function):
    """y = x @ W^T + b for x of shape (R, Cin) with R >> Cin, Cout (R = B*M*K grouped points).

    Forward and dgrad are ordinary GEMMs.  The weight gradient dW = g^T x is a (Cout x R) @ (R x Cin)
    product whose OUTPUT is a single small tile: the BLAS picks a one-workgroup kernel that walks the
    whole R = 10^6 reduction alone (measured 0.57-1.1 ms per layer on MI355X, 8 ms per train step).
    Here the reduction is split over R in chunks (a batched GEMM that fills the chip) and the partial
    (chunks, Cout, Cin) products are summed -- same sum, different association."""

    CHUNK = 4096

    @staticmethod
    def forward(ctx, x, weight, bias):
        ctx.save_for_backward(x, weight)
        return torch.addmm(bias, x, weight.t())

    @staticmethod
    def backward(ctx, g):
        x, weight = ctx.saved_tensors
        g = g.contiguous()
        gx = g @ weight if ctx.needs_input_grad[0] else None
        r = x.shape[0]
        chunk = _TallSkinnyLinear.CHUNK
        main = (r // chunk) * chunk
        gw = None
        if main:
            parts = torch.bmm(g[:main].view(-1, chunk, g.shape[1]).transpose(1, 2), x[:main].view(-1, chunk, x.shape[1]))
            gw = parts.sum(dim=0)
        if main < r:
            tail = g[main:].t() @ x[main:]
            gw = tail if gw is None else gw + tail
        return gx, gw, g.sum(dim=0)


def tall_skinny_linear(x, weight, bias):
    """x (R, Cin) -> (R, Cout); the split-K path only where the plain wgrad GEMM degenerates"""
    if x.shape[0] >= 32 * _TallSkinnyLinear.CHUNK and weight.shape[0] * weight.shape[1] <= 512 * 512:
        return _TallSkinnyLinear.apply(x, weight, bias)
    return torch.nn.functional.linear(x, weight, bias)


class SharedMLPLayer(nn.Module):
    """tf_util.conv2d(..., [1,1], bn=True) on a channel-last tensor: Linear + bias, BN, ReLU"""

    def __init__(self, cin, cout, bn=True, bn_decay=0.9, relu=True):
        super().__init__()
        self.fc = nn.Linear(cin, cout, bias=True)
        nn.init.xavier_uniform_(self.fc.weight)     # tf.contrib.layers.xavier_initializer (tf_util.py:42)
        nn.init.zeros_(self.fc.bias)                # tf.constant_initializer(0.0) (tf_util.py:188)
        # batch norm (moments over every axis but channels, tf_util.py:571) and ReLU as ONE fused HIP op
        self.bn = BatchNormReLU(cout, eps=1e-3, momentum=1.0 - bn_decay, relu=relu) if bn else None
        self.relu = relu

    def forward(self, x):
        shape = x.shape
        rows = x.reshape(-1, shape[-1])
        if self.bn is not None:
            y = linear_bn_relu(rows, self.fc.weight, self.fc.bias, self.bn)  # one fused autograd node
        else:
            y = tall_skinny_linear(rows, self.fc.weight, self.fc.bias)
            if self.relu:
                y = torch.relu(y)
        return y.reshape(*shape[:-1], y.shape[-1])


def _mlp(cin, widths, bn, bn_decay):
    layers = []
    for w in widths:
        layers.append(SharedMLPLayer(cin, w, bn=bn, bn_decay=bn_decay))
        cin = w
    return nn.Sequential(*layers), cin


def sample_and_group(npoint, radius, nsample, xyz, points, knn=False, use_xyz=True):
    """pointnet_util.py:24-66.  Returns (new_xyz (B,npoint,3), new_points (B,npoint,nsample,3+C),
    idx (B,npoint,nsample), grouped_xyz (B,npoint,nsample,3) centred on new_xyz).
    Concat order is [grouped_xyz, grouped_points] (:58-60)."""
    new_xyz = gather_point(xyz, farthest_point_sample(npoint, xyz))
    if knn:
        _, idx = knn_point(nsample, xyz, new_xyz)
        grouped_xyz = group_point(xyz, idx) - new_xyz.unsqueeze(2)
    else:
        # query_ball_point + group_point(xyz) + centring in one launch
        idx, _, grouped_xyz = query_ball_group(radius, nsample, xyz, new_xyz, center=True)
    if points is not None:
        grouped_points = group_point(points, idx)
        new_points = torch.cat([grouped_xyz, grouped_points], dim=-1) if use_xyz else grouped_points
    else:
        new_points = grouped_xyz
    return new_xyz, new_points, idx, grouped_xyz


def sample_and_group_all(xyz, points, use_xyz=True):
    """pointnet_util.py:69-101: one group holding every point, centroid (0,0,0)"""
    b, n, _ = xyz.shape
    new_xyz = torch.zeros((b, 1, 3), dtype=xyz.dtype, device=xyz.device)
    idx = torch.arange(n, dtype=torch.int32, device=xyz.device).reshape(1, 1, n).repeat(b, 1, 1)
    grouped_xyz = xyz.reshape(b, 1, n, 3)
    if points is not None:
        new_points = torch.cat([xyz, points], dim=2) if use_xyz else points
        new_points = new_points.unsqueeze(1)
    else:
        new_points = grouped_xyz
    return new_xyz, new_points, idx, grouped_xyz


class PointnetSAModule(nn.Module):
    """pointnet_sa_module, pointnet_util.py:104-220 (NHWC path)."""

    def __init__(self, npoint, radius, nsample, in_channel, mlp, mlp2=None, group_all=False, bn=True, bn_decay=0.9,
                 pooling="max", knn=False, use_xyz=True):
        super().__init__()
        self.npoint, self.radius, self.nsample = npoint, radius, nsample
        self.group_all, self.pooling, self.knn, self.use_xyz = group_all, pooling, knn, use_xyz
        cin = in_channel + (3 if (use_xyz or in_channel == 0) else 0)
        self.mlp, cout = _mlp(cin, mlp, bn, bn_decay)
        if pooling == "max_and_avg":
            cout *= 2
        self.mlp2, cout = _mlp(cout, mlp2, bn, bn_decay) if mlp2 is not None else (None, cout)
        self.out_channel = cout

    def geometry(self, xyz):
        """The feature-independent half of sample_and_group: (new_xyz, idx, centred grouped_xyz).
        Depends on coordinates only, so it can run ahead of the MLPs (GeometryPrefetcher)."""
        with torch.no_grad():
            new_xyz = gather_point(xyz, farthest_point_sample(self.npoint, xyz))
            if self.knn:
                _, idx = knn_point(self.nsample, xyz, new_xyz)
                grouped_xyz = group_point(xyz, idx) - new_xyz.unsqueeze(2)
            else:
                idx, _, grouped_xyz = query_ball_group(self.radius, self.nsample, xyz, new_xyz, center=True)
        return new_xyz, idx, grouped_xyz

    def forward(self, xyz, points, geom=None):
        if self.group_all:
            new_xyz, new_points, idx, grouped_xyz = sample_and_group_all(xyz, points, self.use_xyz)
        elif geom is not None or not xyz.requires_grad:
            # coordinates are inputs in every reference model: the feature-independent half runs without autograd
            new_xyz, idx, grouped_xyz = geom if geom is not None else self.geometry(xyz)
            if (GATHER_ON_LOAD and self.pooling == "max" and (self.use_xyz or points is None) and xyz.is_cuda
                    and grouped_mlp_fusable(self.mlp, points, idx, GATHER_MIN_CFEAT)
                    and (all(l.bn.training for l in self.mlp) and torch.is_grad_enabled() or not any(l.bn.training for l in self.mlp))):
                # idx -> gather -> centre -> MLP -> max as one chain; the grouped (B,M,K,C+3) tensor of
                # pointnet_util.py:58-60 is never written (the first layer reads the neighbourhoods in place)
                bsz, npt, k = idx.shape
                new_points = shared_mlp_grouped(self.mlp, points, idx, grouped_xyz, xyz_first=True).reshape(bsz, npt, 1, -1)
                if self.mlp2 is not None:
                    new_points = self.mlp2(new_points)
                return new_xyz, new_points.squeeze(2), idx
            if points is not None and self.use_xyz:
                # [xyz, features] rows written once, padded to a multiple of 4 columns (shared_mlp pads the weight)
                new_points = group_concat(points, idx, grouped_xyz)
            elif points is not None:
                new_points = group_point(points, idx)
            else:
                new_points = grouped_xyz
        else:
            new_xyz, new_points, idx, grouped_xyz = sample_and_group(self.npoint, self.radius, self.nsample, xyz,
                                                                     points, self.knn, self.use_xyz)
        bsz, npt, k, cin = new_points.shape
        last = self.mlp[-1]
        fuse_pool = (self.pooling == "max" and last.bn is not None and last.relu and k <= 255 and new_points.is_cuda)
        if fuse_pool:
            # the whole MLP and the max over the K grouped points as one node (mlp.shared_mlp)
            pooled = shared_mlp(self.mlp, new_points.reshape(-1, cin), pool_k=k)
            new_points = pooled.reshape(bsz, npt, 1, -1)
        else:
            new_points = shared_mlp(self.mlp, new_points.reshape(-1, cin)).reshape(bsz, npt, k, -1)
        if fuse_pool:
            pass
        elif self.pooling == "max":
            new_points = new_points.max(dim=2, keepdim=True).values
        elif self.pooling == "avg":
            new_points = new_points.mean(dim=2, keepdim=True)
        elif self.pooling == "weighted_avg":
            dists = torch.norm(grouped_xyz, dim=-1, keepdim=True)
            w = torch.exp(-dists * 5)
            w = w / w.sum(dim=2, keepdim=True)
            new_points = (new_points * w).sum(dim=2, keepdim=True)
        elif self.pooling == "max_and_avg":
            new_points = torch.cat([new_points.mean(dim=2, keepdim=True),
                                    new_points.max(dim=2, keepdim=True).values], dim=-1)
        else:
            raise ValueError("unknown pooling %r" % self.pooling)
        if self.mlp2 is not None:
            new_points = self.mlp2(new_points)
        return new_xyz, new_points.squeeze(2), idx


class PointnetSAModuleMSG(nn.Module):
    """pointnet_sa_module_msg, pointnet_util.py:223-286.  Concat order inside a scale is
    [grouped_points, grouped_xyz] (:264) -- the opposite of the single-scale module."""

    def __init__(self, npoint, radius_list, nsample_list, in_channel, mlp_list, bn=True, bn_decay=0.9, use_xyz=True):
        super().__init__()
        self.npoint, self.radius_list, self.nsample_list, self.use_xyz = npoint, radius_list, nsample_list, use_xyz
        cin = in_channel + (3 if (use_xyz or in_channel == 0) else 0)
        mlps, cout = [], 0
        for widths in mlp_list:
            m, c = _mlp(cin, widths, bn, bn_decay)
            mlps.append(m)
            cout += c
        self.mlps = nn.ModuleList(mlps)
        self.out_channel = cout

    def forward(self, xyz, points):
        new_xyz = gather_point(xyz, farthest_point_sample(self.npoint, xyz))
        outs = []
        for radius, nsample, mlp in zip(self.radius_list, self.nsample_list, self.mlps):
            idx, _, grouped_xyz = query_ball_group(radius, nsample, xyz, new_xyz, center=True)
            if (GATHER_ON_LOAD and (self.use_xyz or points is None) and not xyz.requires_grad and xyz.is_cuda
                    and grouped_mlp_fusable(mlp, points, idx, GATHER_MIN_CFEAT)
                    and (all(l.bn.training for l in mlp) and torch.is_grad_enabled() or not any(l.bn.training for l in mlp))):
                outs.append(shared_mlp_grouped(mlp, points, idx, grouped_xyz, xyz_first=False).reshape(idx.shape[0], idx.shape[1], -1))
                continue
            if points is not None and self.use_xyz and not xyz.requires_grad:
                grouped = group_concat(points, idx, grouped_xyz, xyz_last=True)   # [features, xyz, 0-pad], one pass
            elif points is not None:
                grouped = group_point(points, idx)
                if self.use_xyz:
                    grouped = torch.cat([grouped, grouped_xyz], dim=-1)
            else:
                grouped = grouped_xyz
            bsz, npt, k, cin = grouped.shape
            # the scale's MLP and the max over its K grouped points as one node (mlp.shared_mlp)
            outs.append(shared_mlp(mlp, grouped.reshape(-1, cin), pool_k=k).reshape(bsz, npt, -1))
        return new_xyz, torch.cat(outs, dim=-1)


def three_nn_weights(dist):
    """pointnet_util.py:304-307: dist = max(dist, 1e-10); w = (1/dist) / sum(1/dist)"""
    d = torch.clamp(dist, min=1e-10)
    inv = 1.0 / d
    return inv / inv.sum(dim=2, keepdim=True)


class PointnetFPModule(nn.Module):
    """pointnet_fp_module, pointnet_util.py:289-330: xyz1 dense (B,N,3), xyz2 sparse (B,M,3),
    points1 (B,N,C1) or None, points2 (B,M,C2) -> (B,N,mlp[-1]); concat [interpolated, points1] (:311-313)."""

    def __init__(self, in_channel, mlp, bn=True, bn_decay=0.9):
        super().__init__()
        self.mlp, self.out_channel = _mlp(in_channel, mlp, bn, bn_decay)

    @staticmethod
    def geometry(xyz1, xyz2):
        """three_nn + inverse-distance weights: coordinates only"""
        with torch.no_grad():
            dist, idx = three_nn(xyz1, xyz2)
            m = xyz2.shape[1]
            inverse = three_nn_inverse(idx, m) if m <= INVERSE_MAX_KNOWN else None
            return idx, three_nn_weights(dist), inverse

    def forward(self, xyz1, xyz2, points1, points2, geom=None):
        idx, weight, inverse = geom if geom is not None else self.geometry(xyz1, xyz2)
        if inverse is not None and self.training and torch.is_grad_enabled():
            # [interpolated, skip] rows written once, padded to a multiple of 4 columns (shared_mlp pads the weight)
            new_points = three_interpolate_concat(points2, points1, idx, weight, inverse)
        else:
            interpolated = three_interpolate(points2, idx, weight, inverse if torch.is_grad_enabled() else None)
            new_points = torch.cat([interpolated, points1], dim=2) if points1 is not None else interpolated
        bsz, npt, cin = new_points.shape
        return shared_mlp(self.mlp, new_points.reshape(-1, cin)).reshape(bsz, npt, -1)


class PointnetSAFPStack(nn.Module):
    """The synthetic SA+FP stack of BASELINE.json config 2 / SURVEY.md 8d: SA levels
    (npoint, radius, nsample, mlp) then FP levels back to the input resolution.  Layer wiring
    follows hf/core/feature_extractors/pointnet.py (SA down, FP up with skip links)."""

    def __init__(self, in_channel=1,
                 sa=((4096, 0.5, 32, (32, 32, 64)), (1024, 1.0, 32, (64, 96, 128)), (256, 2.0, 32, (128, 196, 256))),
                 fp=((256, 256), (256, 256), (128, 128))):
        super().__init__()
        self.sa = nn.ModuleList()
        chans = [in_channel]
        c = in_channel
        for npoint, radius, nsample, mlp in sa:
            m = PointnetSAModule(npoint, radius, nsample, c, list(mlp))
            self.sa.append(m)
            c = m.out_channel
            chans.append(c)
        self.fp = nn.ModuleList()
        for level, mlp in enumerate(fp):                       # deepest first
            skip = chans[len(sa) - 1 - level]
            m = PointnetFPModule(c + skip, list(mlp))
            self.fp.append(m)
            c = m.out_channel
        self.out_channel = c

    def geometry(self, xyz):
        """Every sampling / grouping / neighbour result of one pass: a function of xyz alone.
        Returns {"sa": [(new_xyz, idx, grouped_xyz) per level], "fp": [(idx3, weight, inverse index) per level, deepest first]}."""
        sa, xyzs = [], [xyz]
        for m in self.sa:
            g = m.geometry(xyzs[-1])
            sa.append(g)
            xyzs.append(g[0])
        fp = []
        for level in range(len(self.fp)):
            i = len(self.sa) - 1 - level
            fp.append(PointnetFPModule.geometry(xyzs[i], xyzs[i + 1]))
        return {"sa": sa, "fp": fp}

    def forward(self, xyz, points, geometry=None):
        xyzs, feats = [xyz], [points]
        for li, m in enumerate(self.sa):
            nx, nf, _ = m(xyzs[-1], feats[-1], geometry["sa"][li] if geometry is not None else None)
            xyzs.append(nx)
            feats.append(nf)
        up = feats[-1]
        for level, m in enumerate(self.fp):
            i = len(self.sa) - 1 - level
            up = m(xyzs[i], xyzs[i + 1], feats[i], up, geometry["fp"][level] if geometry is not None else None)
        return up


# ---------------------------------------------------------------- IoU / NMS adapters (hf/core/compute_iou.py)
def boxes3d_to_bev(boxes3d):
    """compute_iou.py:7-20: (...,7)[x,y,z,l,w,h,ry] -> (...,5)[x - l/2, z - w/2, x + l/2, z + w/2, ry]"""
    cu, cv = boxes3d[..., 0], boxes3d[..., 2]
    half_l, half_w = boxes3d[..., 3] / 2, boxes3d[..., 4] / 2
    return torch.stack([cu - half_l, cv - half_w, cu + half_l, cv + half_w, boxes3d[..., 6]], dim=-1)


def box3d_iou(boxes_a, boxes_b):
    """compute_iou.py:23-64: BEV overlap x height overlap (y points down: a box spans y-h .. y)."""
    overlaps_bev, iou_2d = _compute_bev_iou(boxes3d_to_bev(boxes_a), boxes3d_to_bev(boxes_b))
    a_min, a_max = (boxes_a[:, 1] - boxes_a[:, 5]).reshape(-1, 1), boxes_a[:, 1].reshape(-1, 1)
    b_min, b_max = (boxes_b[:, 1] - boxes_b[:, 5]).reshape(1, -1), boxes_b[:, 1].reshape(1, -1)
    overlaps_h = torch.clamp(torch.minimum(a_max, b_max) - torch.maximum(a_min, b_min), min=0)
    overlaps_3d = overlaps_bev * overlaps_h
    vol_a = (boxes_a[:, 3] * boxes_a[:, 4] * boxes_a[:, 5]).reshape(-1, 1)
    vol_b = (boxes_b[:, 3] * boxes_b[:, 4] * boxes_b[:, 5]).reshape(1, -1)
    iou_3d = overlaps_3d / torch.clamp(vol_a + vol_b - overlaps_3d, min=1e-7)
    return iou_3d, iou_2d


def oriented_nms_3d(boxes, scores, thresh):
    """compute_iou.py:67-80: to BEV, sort by score (descending), oriented NMS, map back."""
    boxes_bev = boxes3d_to_bev(boxes)
    sorted_idxs = torch.sort(scores, descending=True, stable=True).indices
    keep = _oriented_nms(boxes_bev[sorted_idxs].contiguous(), thresh)
    return sorted_idxs[keep.long()].to(torch.int32)


def sb_nms(boxes, scores, nms_iou_thresh, nms_size, fixed_num_proposal_nms=True):
    """model_util.py:101-142 (single frame): NMS, truncate to nms_size, optionally drop the
    duplicated padding, pad with -1.  Returns (indices (nms_size,), number before padding)."""
    ind = oriented_nms_3d(boxes, scores, nms_iou_thresh)[:nms_size]
    if not fixed_num_proposal_nms:
        # tf.unique keeps first occurrences in order; the padding repeats keep[0]
        keep = torch.ones_like(ind, dtype=torch.bool)
        keep[1:] = ind[1:] != ind[0]
        ind = ind[keep]
    n = ind.shape[0]
    if nms_size > n:
        ind = torch.cat([ind, torch.full((nms_size - n,), -1, dtype=ind.dtype, device=ind.device)])
    return ind, n


def box_3d_to_box_8co(boxes_3d):
    """box_8c_encoder.py:101-185: (N,7)[x,y,z,l,w,h,ry] -> (N,3,8) corners, order P1..P8 of :21-37."""
    x, y, z, l, w, h, ry = [boxes_3d[:, i] for i in range(7)]
    s, c = torch.sin(ry), torch.cos(ry)
    hl, hw = l / 2, w / 2
    xc = torch.stack([hl, hl, -hl, -hl, hl, hl, -hl, -hl], dim=1)
    zero = torch.zeros_like(h)
    yc = torch.stack([zero, zero, zero, zero, -h, -h, -h, -h], dim=1)
    zc = torch.stack([hw, -hw, -hw, hw, hw, -hw, -hw, hw], dim=1)
    X = c[:, None] * xc + s[:, None] * zc + x[:, None]
    Y = yc + y[:, None]
    Z = -s[:, None] * xc + c[:, None] * zc + z[:, None]
    return torch.stack([X, Y, Z], dim=1)
