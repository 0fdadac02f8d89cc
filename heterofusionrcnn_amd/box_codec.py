"""Bin-based 3D box encoding: hf/core/bin_based_box3d_encoder.py:9-269 with the same argument
order.  Each call is ONE element-wise HIP kernel (csrc/glue.hip) instead of ~20 TensorFlow ops.

decode(...) == tf_decode(...): (B,p,K,7) for (B,p,3) reference points (RPN), (N,K,7) for (N,3) (RCNN).
encode(...) == tf_encode(...): the 8-tuple of bin assignments and normalised residuals.
ref_theta is a tensor (B,p) / (N,) or the constant 0 (the RPN's call, rpn_model.py:623-639, 749-759).
"""
import numpy as np
import torch

from . import _lib
from ._lib import check, dev_tensor, ptr, require, stream_ptr


_CONSTS = {}


def const_f32(device, values):
    """Small fp32 constant on the device, uploaded ONCE per (device, values): a pageable host-to-device copy is a
    synchronising call, and one of them per step makes the whole train step wait for the device."""
    a = np.ascontiguousarray(np.asarray(values, dtype=np.float64).astype(np.float32))
    key = (str(device), a.shape, a.tobytes())
    t = _CONSTS.get(key)
    if t is None:
        t = torch.as_tensor(a, device=device)
        _CONSTS[key] = t
    return t


_f32 = const_f32


def _theta(ref_theta, rows, device):
    if isinstance(ref_theta, torch.Tensor):
        t = dev_tensor(ref_theta.detach().reshape(-1), torch.float32, "ref_theta")
        require(t.numel() == rows, "ref_theta must hold one angle per reference point")
        return t
    require(ref_theta == 0, "ref_theta must be a tensor or the constant 0")
    return None


def decode(ref_pts, ref_theta, bin_x, res_x_norm, bin_z, res_z_norm, bin_theta, res_theta_norm, res_y, res_size_norm,
           mean_sizes, Ss, DELTAs, R, DELTA_THETA):
    require(ref_pts.dim() in (2, 3) and ref_pts.shape[-1] == 3, "decode expects (B,p,3) or (N,3) ref_pts")
    lead = tuple(ref_pts.shape[:-1])
    k = bin_x.shape[-1]
    rows = int(np.prod(lead)) if lead else 0
    dev = ref_pts.device
    f = lambda t, name: dev_tensor(t.detach().reshape(rows, -1), torch.float32, name)
    i = lambda t, name: dev_tensor(t.detach().reshape(rows, -1).to(torch.int32), torch.int32, name)
    require(tuple(bin_x.shape) == lead + (k,) and tuple(mean_sizes.shape) == lead + (k, 3) and
            tuple(res_size_norm.shape) == lead + (k, 3), "decode expects (..., K) bins / residuals and (..., K, 3) sizes")
    ss, deltas = _f32(dev, Ss), _f32(dev, DELTAs)
    require(ss.numel() == k and deltas.numel() == k, "Ss / DELTAs must have one entry per class")
    boxes = torch.empty(lead + (k, 7), dtype=torch.float32, device=dev)
    # converted / reshaped operands are held in `args` until the launch is enqueued (a temporary freed earlier could be
    # handed out again by the caching allocator and overwritten by the next conversion)
    args = [f(ref_pts, "ref_pts"), _theta(ref_theta, rows, dev), i(bin_x, "bin_x"), f(res_x_norm, "res_x_norm"),
            i(bin_z, "bin_z"), f(res_z_norm, "res_z_norm"), i(bin_theta, "bin_theta"), f(res_theta_norm, "res_theta_norm"),
            f(res_y, "res_y"), f(res_size_norm, "res_size_norm"), f(mean_sizes, "mean_sizes"), ss, deltas]
    check(_lib.lib().hf_bin_box_decode(rows, k, *[ptr(a) for a in args], float(np.float32(R)),
                                       float(np.float32(DELTA_THETA)), ptr(boxes), stream_ptr()), "bin_box_decode")
    return boxes


def encode(ref_pts, ref_theta, boxes_3d, mean_sizes, Ss, DELTAs, R, DELTA_THETA, K):
    require(ref_pts.dim() in (2, 3) and ref_pts.shape[-1] == 3, "encode expects (B,p,3) or (N,3) ref_pts")
    lead = tuple(ref_pts.shape[:-1])
    rows = int(np.prod(lead)) if lead else 0
    dev = ref_pts.device
    k = int(K)
    require(tuple(boxes_3d.shape) == lead + (7,) and tuple(mean_sizes.shape) == lead + (3,),
            "encode expects (..., 7) boxes_3d and (..., 3) mean_sizes")
    f = lambda t, name: dev_tensor(t.detach().reshape(rows, -1), torch.float32, name)
    ss64, d64 = np.asarray(Ss, dtype=np.float64), np.asarray(DELTAs, dtype=np.float64)
    require(ss64.size == k and d64.size == k, "Ss / DELTAs must have one entry per class")
    ss, deltas, hi_xz = _f32(dev, ss64), _f32(dev, d64), _f32(dev, 2.0 * ss64 - 1e-3)
    e = lambda shape, dt: torch.empty(lead + shape, dtype=dt, device=dev)
    bin_x, res_x, bin_z, res_z = e((k,), torch.int32), e((k,), torch.float32), e((k,), torch.int32), e((k,), torch.float32)
    bin_t, res_t, res_y, res_s = e((), torch.int32), e((), torch.float32), e((), torch.float32), e((3,), torch.float32)
    args = [f(ref_pts, "ref_pts"), _theta(ref_theta, rows, dev), f(boxes_3d, "boxes_3d"), f(mean_sizes, "mean_sizes"), ss,
            deltas, hi_xz]  # held until the launch is enqueued, see decode()
    check(_lib.lib().hf_bin_box_encode(rows, k, 1 if ref_pts.dim() == 2 else 0, *[ptr(a) for a in args],
                                       float(np.float32(R)), float(np.float32(2.0 * float(R) - 1e-3)),
                                       float(np.float32(DELTA_THETA)), float(np.float32(0.5 * float(DELTA_THETA))),
                                       ptr(bin_x), ptr(res_x), ptr(bin_z), ptr(res_z), ptr(bin_t), ptr(res_t), ptr(res_y),
                                       ptr(res_s), stream_ptr()), "bin_box_encode")
    return bin_x, res_x, bin_z, res_z, bin_t, res_t, res_y, res_s


def decode_head(head, ref_pts, ref_theta, mean_sizes_k, num_bin_x, num_bin_z, num_bin_theta, Ss, DELTAs, R, DELTA_THETA,
                cls=None):
    """The decoding block of the RPN / RCNN heads as one kernel (rpn_model.py:593-642): head (..., K*D) or (..., K, D)
    with D = 2*num_bin_x + 2*num_bin_z + 2*num_bin_theta + 4 -> boxes (..., K, 7), or (..., 7) when `cls` (...) picks
    each row's class.  mean_sizes_k (K, 3) are the per-class mean sizes (`cluster_sizes`)."""
    require(ref_pts.dim() in (2, 3) and ref_pts.shape[-1] == 3, "decode_head expects (B,p,3) or (N,3) ref_pts")
    lead = tuple(ref_pts.shape[:-1])
    rows = int(np.prod(lead)) if lead else 0
    dev = ref_pts.device
    ms = const_f32(dev, mean_sizes_k) if not isinstance(mean_sizes_k, torch.Tensor) \
        else dev_tensor(mean_sizes_k.detach(), torch.float32, "mean_sizes_k")
    require(ms.dim() == 2 and ms.shape[1] == 3, "mean_sizes_k must be (K, 3)")
    k = ms.shape[0]
    d = 2 * num_bin_x + 2 * num_bin_z + 2 * num_bin_theta + 4
    require(head.numel() == rows * k * d, "head must hold K * (2*nbx + 2*nbz + 2*nbt + 4) values per reference point")
    head = dev_tensor(head.detach().reshape(rows, k * d), torch.float32, "head")
    ss, deltas = _f32(dev, Ss), _f32(dev, DELTAs)
    require(ss.numel() == k and deltas.numel() == k, "Ss / DELTAs must have one entry per class")
    c = None
    if cls is not None:
        c = dev_tensor(cls.detach().reshape(rows).to(torch.int32), torch.int32, "cls")
        boxes = torch.zeros(lead + (7,), dtype=torch.float32, device=dev)
    else:
        boxes = torch.empty(lead + (k, 7), dtype=torch.float32, device=dev)
    ref = dev_tensor(ref_pts.detach().reshape(rows, 3), torch.float32, "ref_pts")
    theta = _theta(ref_theta, rows, dev)
    ms = ms.contiguous()  # every operand has a name here: nothing is freed before the launch is enqueued
    check(_lib.lib().hf_bin_head_decode(rows, k, num_bin_x, num_bin_z, num_bin_theta, ptr(head), ptr(ref), ptr(theta),
                                        ptr(ms), ptr(ss), ptr(deltas), float(np.float32(R)),
                                        float(np.float32(DELTA_THETA)), ptr(c), ptr(boxes), stream_ptr()),
          "bin_head_decode")
    return boxes
