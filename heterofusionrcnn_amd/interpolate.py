"""interpolate ops -- same surface as the reference's interpolate/tf_interpolate.py:11-49."""
import torch

from . import _lib
from ._lib import check, dev_tensor, ptr, require, stream_ptr


def three_nn(xyz1, xyz2):
    """xyz1 (B,N,3) unknown, xyz2 (B,M,3) known -> dist (B,N,3) SQUARED distances, idx (B,N,3) int32.
    Reference: tf_interpolate.py:11-23; non-differentiable."""
    require(xyz1.dim() == 3 and xyz1.shape[2] == 3, "ThreeNN expects (b,n,3) unknown shape")
    require(xyz2.dim() == 3 and xyz2.shape[2] == 3, "ThreeNN expects (b,m,3) known shape")
    require(xyz1.shape[0] == xyz2.shape[0], "ThreeNN expects the same batch size")
    require(xyz2.shape[1] > 0, "ThreeNN expects at least one known point")
    xyz1 = dev_tensor(xyz1.detach(), torch.float32, "xyz1")
    xyz2 = dev_tensor(xyz2.detach(), torch.float32, "xyz2")
    b, n, _ = xyz1.shape
    m = xyz2.shape[1]
    dist = torch.empty((b, n, 3), dtype=torch.float32, device=xyz1.device)
    idx = torch.empty((b, n, 3), dtype=torch.int32, device=xyz1.device)
    L = _lib.lib()
    nbytes = L.hf_three_nn_workspace(b, m)
    if nbytes:  # grid ring-search kernels (the k = 3 case of knn_point); larger clouds take the all-pairs kernel
        ws = torch.empty((nbytes // 4,), dtype=torch.float32, device=xyz1.device)
        check(L.hf_three_nn_sorted(b, n, m, ptr(xyz1), ptr(xyz2), ptr(dist), ptr(idx), ptr(ws), nbytes, stream_ptr()),
              "three_nn")
    else:
        check(L.hf_three_nn(b, n, m, ptr(xyz1), ptr(xyz2), ptr(dist), ptr(idx), stream_ptr()), "three_nn")
    return dist, idx


def three_nn_all_pairs(xyz1, xyz2):
    """the reference's O(n*m) scan (ThreeNNGpuOp as written), kept as its own entry point"""
    xyz1 = dev_tensor(xyz1.detach(), torch.float32, "xyz1")
    xyz2 = dev_tensor(xyz2.detach(), torch.float32, "xyz2")
    b, n, _ = xyz1.shape
    m = xyz2.shape[1]
    dist = torch.empty((b, n, 3), dtype=torch.float32, device=xyz1.device)
    idx = torch.empty((b, n, 3), dtype=torch.int32, device=xyz1.device)
    check(_lib.lib().hf_three_nn(b, n, m, ptr(xyz1), ptr(xyz2), ptr(dist), ptr(idx), stream_ptr()), "three_nn")
    return dist, idx


INVERSE_MAX_KNOWN = 8192


def three_nn_inverse(idx, m):
    """idx (B,N,3) int32 from three_nn over M known points -> (offsets (B,M+1), entries (B,3N)) int32: for every
    known point the flat (unknown*3 + slot) positions that reference it, ascending.  Coordinates-only, like
    three_nn itself, so it can be prepared ahead of the features (pipeline.GeometryPrefetcher)."""
    idx = dev_tensor(idx, torch.int32, "idx")
    require(idx.dim() == 3 and idx.shape[2] == 3, "three_nn_inverse expects (b,n,3) idx shape")
    require(0 < m <= INVERSE_MAX_KNOWN, "three_nn_inverse expects 0 < m <= %d" % INVERSE_MAX_KNOWN)
    b, n, _ = idx.shape
    offsets = torch.empty((b, m + 1), dtype=torch.int32, device=idx.device)
    entries = torch.empty((b, 3 * n), dtype=torch.int32, device=idx.device)
    check(_lib.lib().hf_three_nn_inverse(b, n, m, ptr(idx), ptr(offsets), ptr(entries), stream_ptr()), "three_nn_inverse")
    return offsets, entries


class _ThreeInterpolateInv(torch.autograd.Function):
    """three_interpolate whose gradient gathers over a prepared inverse index instead of scattering atomics"""

    @staticmethod
    def forward(ctx, points, idx, weight, offsets, entries):
        b, m, c = points.shape
        n = idx.shape[1]
        out = torch.empty((b, n, c), dtype=torch.float32, device=points.device)
        check(_lib.lib().hf_three_interpolate_cl(b, m, c, n, ptr(points), ptr(idx), ptr(weight), ptr(out),
                                                 stream_ptr()), "three_interpolate")
        ctx.save_for_backward(weight, offsets, entries)
        ctx.shape = (b, m, c, n)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        weight, offsets, entries = ctx.saved_tensors
        b, m, c, n = ctx.shape
        grad_out = grad_out.contiguous()
        g = torch.empty((b, m, c), dtype=torch.float32, device=grad_out.device)
        check(_lib.lib().hf_three_interpolate_cl_grad_gather(b, n, c, m, ptr(grad_out), ptr(weight), ptr(offsets),
                                                             ptr(entries), ptr(g), stream_ptr()),
              "three_interpolate_grad_gather")
        return g, None, None, None, None


class _InterpolateConcat(torch.autograd.Function):
    @staticmethod
    def forward(ctx, points, skip, idx, weight, offsets, entries, width):
        b, m, c = points.shape
        n = idx.shape[1]
        c1 = skip.shape[2] if skip is not None else 0
        out = torch.empty((b, n, width), dtype=torch.float32, device=points.device)
        check(_lib.lib().hf_three_interpolate_concat(b, m, c, n, c1, width, ptr(points), ptr(idx), ptr(weight), ptr(skip),
                                                     ptr(out), stream_ptr()), "three_interpolate_concat")
        ctx.save_for_backward(weight, offsets, entries)
        ctx.shape = (b, m, c, n, c1, width)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        weight, offsets, entries = ctx.saved_tensors
        b, m, c, n, c1, width = ctx.shape
        grad_out = grad_out.contiguous()
        g = None
        if ctx.needs_input_grad[0]:
            g = torch.empty((b, m, c), dtype=torch.float32, device=grad_out.device)
            check(_lib.lib().hf_three_interpolate_concat_grad(b, n, c, m, width, ptr(grad_out), ptr(weight), ptr(offsets),
                                                              ptr(entries), ptr(g), stream_ptr()),
                  "three_interpolate_concat_grad")
        gskip = grad_out[:, :, c:c + c1] if (c1 and ctx.needs_input_grad[1]) else None
        return g, gskip, None, None, None, None, None


def three_interpolate_concat(points, skip, idx, weight, inverse, width=None):
    """[three_interpolate(points, idx, weight), skip, zero padding] as one (B,N,width) tensor: the concat of
    pointnet_fp_module (pointnet_util.py:311-313) without the interpolated temporary.  width defaults to C + C1
    rounded up to a multiple of 4.  `inverse` = three_nn_inverse(idx, M); gradients w.r.t. points and skip."""
    points = dev_tensor(points, torch.float32, "points")
    idx = dev_tensor(idx, torch.int32, "idx")
    weight = dev_tensor(weight.detach(), torch.float32, "weight")
    require(points.dim() == 3 and idx.dim() == 3 and idx.shape[2] == 3 and weight.shape == idx.shape,
            "ThreeInterpolate expects (b,m,c) points and (b,n,3) idx / weight")
    c1 = 0
    if skip is not None:
        skip = dev_tensor(skip, torch.float32, "skip")
        require(skip.dim() == 3 and skip.shape[:2] == idx.shape[:2], "skip features must be (b,n,c1)")
        c1 = skip.shape[2]
    c = points.shape[2]
    if width is None:
        width = (c + c1 + 3) // 4 * 4
    require(width >= c + c1 and width % 4 == 0, "width must be >= c + c1 and a multiple of 4")
    offsets, entries = inverse
    require(offsets.shape == (points.shape[0], points.shape[1] + 1) and entries.shape == (idx.shape[0], 3 * idx.shape[1]),
            "ThreeInterpolate expects the inverse of this idx: offsets (b,m+1), entries (b,3n)")
    return _InterpolateConcat.apply(points, skip, idx, weight, dev_tensor(offsets, torch.int32, "offsets"),
                                    dev_tensor(entries, torch.int32, "entries"), width)


class _ThreeInterpolate(torch.autograd.Function):
    @staticmethod
    def forward(ctx, points, idx, weight):
        b, m, c = points.shape
        n = idx.shape[1]
        out = torch.empty((b, n, c), dtype=torch.float32, device=points.device)
        check(_lib.lib().hf_three_interpolate_cl(b, m, c, n, ptr(points), ptr(idx), ptr(weight), ptr(out),
                                                 stream_ptr()), "three_interpolate")
        ctx.save_for_backward(idx, weight)
        ctx.shape = (b, m, c)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        idx, weight = ctx.saved_tensors
        b, m, c = ctx.shape
        n = idx.shape[1]
        grad_out = grad_out.contiguous()
        g = torch.empty((b, m, c), dtype=torch.float32, device=grad_out.device)
        check(_lib.lib().hf_three_interpolate_cl_grad(b, n, c, m, ptr(grad_out), ptr(idx), ptr(weight), ptr(g),
                                                      stream_ptr()), "three_interpolate_grad")
        return g, None, None


def three_interpolate(points, idx, weight, inverse=None):
    """points (B,M,C), idx (B,N,3) int32, weight (B,N,3) -> (B,N,C).
    `inverse` = three_nn_inverse(idx, M) (optional): the gradient then gathers instead of using atomics.
    Reference: tf_interpolate.py:26-49 -- it transposes to (B,C,M), runs the channel-first op and
    transposes back; the channel-last kernel gives the same values without the two transposes.
    Gradient w.r.t. points only (idx, weight get None, :44-48)."""
    require(points.dim() == 3, "ThreeInterpolate expects (b,m,c) points shape")
    require(idx.dim() == 3 and idx.shape[2] == 3 and idx.shape[0] == points.shape[0],
            "ThreeInterpolate expects (b,n,3) idx shape")
    require(weight.shape == idx.shape, "ThreeInterpolate expects (b,n,3) weight shape")
    points = dev_tensor(points, torch.float32, "points")
    idx = dev_tensor(idx, torch.int32, "idx")
    weight = dev_tensor(weight.detach(), torch.float32, "weight")
    if inverse is not None:
        offsets, entries = inverse
        require(offsets.shape == (points.shape[0], points.shape[1] + 1) and entries.shape == (idx.shape[0], 3 * idx.shape[1]),
                "ThreeInterpolate expects the inverse of this idx: offsets (b,m+1), entries (b,3n)")
        return _ThreeInterpolateInv.apply(points, idx, weight, dev_tensor(offsets, torch.int32, "offsets"),
                                          dev_tensor(entries, torch.int32, "entries"))
    return _ThreeInterpolate.apply(points, idx, weight)


def three_interpolate_channel_first(points, idx, weight):
    """The raw op layout of tf_interpolate.cpp:25-37: points (B,C,M) -> (B,C,N); no autograd."""
    require(points.dim() == 3 and idx.dim() == 3 and idx.shape[2] == 3 and weight.shape == idx.shape,
            "ThreeInterpolate expects (b,c,m) points and (b,n,3) idx / weight")
    points = dev_tensor(points.detach(), torch.float32, "points")
    idx = dev_tensor(idx, torch.int32, "idx")
    weight = dev_tensor(weight.detach(), torch.float32, "weight")
    b, c, m = points.shape
    n = idx.shape[1]
    out = torch.empty((b, c, n), dtype=torch.float32, device=points.device)
    check(_lib.lib().hf_three_interpolate(b, c, m, n, ptr(points), ptr(idx), ptr(weight), ptr(out), stream_ptr()),
          "three_interpolate_channel_first")
    return out


def three_interpolate_channel_first_grad(points_shape, idx, weight, grad_out):
    """ThreeInterpolateGrad op (tf_interpolate.cpp:39-48): grad_out (B,C,N) -> grad_points (B,C,M)."""
    b, c, m = points_shape
    idx = dev_tensor(idx, torch.int32, "idx")
    weight = dev_tensor(weight.detach(), torch.float32, "weight")
    grad_out = dev_tensor(grad_out.detach(), torch.float32, "grad_out")
    n = idx.shape[1]
    g = torch.empty((b, c, m), dtype=torch.float32, device=grad_out.device)
    check(_lib.lib().hf_three_interpolate_grad(b, c, n, m, ptr(grad_out), ptr(idx), ptr(weight), ptr(g),
                                               stream_ptr()), "three_interpolate_channel_first_grad")
    return g
