"""sampling ops -- same surface as the reference's sampling/tf_sampling.py:18-69.

farthest_point_sample(npoint, inp), gather_point(inp, idx) [+ gradient], prob_sample.
"""
import torch

from . import _lib
from ._lib import check, dev_tensor, ptr, require, stream_ptr


FPS_KERNELS = {"auto": 0, "plain": 1, "bucket": 2, "wave": 3}   # HF_FPS_* of include/hfops.h


def farthest_point_sample(npoint, inp, kernel="auto", threads=0):
    """inp (B,N,3) float32 -> (B,npoint) int32.  Reference: tf_sampling.py:61-69 (note the
    Python argument order (npoint, xyz)); non-differentiable (ops.NoGradient, :72).
    kernel / threads force one of the on-chip kernels (FPS_KERNELS; the tests compare them): same output."""
    npoint = int(npoint)
    require(npoint > 0, "FarthestPointSample expects positive npoint")
    require(inp.dim() == 3 and inp.shape[2] == 3, "FarthestPointSample expects (batch_size,num_points,3) inp shape")
    inp = dev_tensor(inp.detach(), torch.float32, "inp")
    b, n, _ = inp.shape
    require(n > 0, "FarthestPointSample expects at least one point")
    out = torch.empty((b, npoint), dtype=torch.int32, device=inp.device)
    L = _lib.lib()
    ws = L.hf_fps_workspace(b, n)
    temp = torch.empty((ws // 4,), dtype=torch.float32, device=inp.device) if ws else None
    if kernel != "auto" or threads:
        check(L.hf_farthest_point_sample_variant(FPS_KERNELS[kernel], int(threads), b, n, npoint, ptr(inp), ptr(temp), ptr(out),
                                                 stream_ptr()), "farthest_point_sample")
        return out
    check(L.hf_farthest_point_sample(b, n, npoint, ptr(inp), ptr(temp), ptr(out), stream_ptr()), "farthest_point_sample")
    return out


class _GatherPoint(torch.autograd.Function):
    @staticmethod
    def forward(ctx, inp, idx):
        b, n, _ = inp.shape
        m = idx.shape[1]
        out = torch.empty((b, m, 3), dtype=torch.float32, device=inp.device)
        check(_lib.lib().hf_gather_point(b, n, m, ptr(inp), ptr(idx), ptr(out), stream_ptr()), "gather_point")
        ctx.save_for_backward(idx)
        ctx.n = n
        return out

    @staticmethod
    def backward(ctx, out_g):
        (idx,) = ctx.saved_tensors
        out_g = out_g.contiguous()
        b, m = idx.shape
        inp_g = torch.empty((b, ctx.n, 3), dtype=torch.float32, device=out_g.device)
        check(_lib.lib().hf_gather_point_grad(b, ctx.n, m, ptr(out_g), ptr(idx), ptr(inp_g), stream_ptr()),
              "gather_point_grad")
        return inp_g, None


def gather_point(inp, idx):
    """inp (B,N,3) float32, idx (B,M) int32 -> (B,M,3).  Reference: tf_sampling.py:38-58
    (gradient w.r.t. inp only, registered at :54-58)."""
    require(inp.dim() == 3 and inp.shape[2] == 3, "GatherPoint expects (batch_size,num_points,3) inp shape")
    require(idx.dim() == 2 and idx.shape[0] == inp.shape[0], "GatherPoint expects (batch_size,num_result) idx shape")
    inp = dev_tensor(inp, torch.float32, "inp")
    idx = dev_tensor(idx, torch.int32, "idx")
    return _GatherPoint.apply(inp, idx)


def prob_sample(inp, inpr):
    """Reference: tf_sampling.py:18-27.  Used by no model or config of the reference (only its
    __main__ demo, :90); deliberately not built -- see DESIGN.md "out of scope"."""
    raise NotImplementedError("prob_sample is not on the hot path and is not provided (DESIGN.md)")
