"""ctypes binding of libhfops.so (the C ABI declared in include/hfops.h).

PyTorch is plumbing here: it owns device memory and streams; every op hands raw device
pointers and the current HIP stream to the C ABI.  There is no CPU fallback and no eager
PyTorch fallback: if the shared library is missing the import of any op module fails loudly.
"""
import ctypes
import os
import subprocess

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
# HFOPS_LIBRARY: load another build of the same ABI (scripts/probes/build_diag.sh makes one with -DHF_DIAG, whose kernels
# honour the diagnostic knobs the product library compiles out); read once at import, never on a launch path
LIB_PATH = os.environ.get("HFOPS_LIBRARY") or os.path.join(CSRC, "libhfops.so")

HF_OK, HF_EINVAL, HF_EHIP, HF_EWORKSPACE = 0, -1, -2, -3

_vp, _i, _f, _sz = ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_size_t

# name -> argtypes (restype int unless listed in _RESTYPES); mirrors include/hfops.h
_SIGNATURES = {
    "hf_farthest_point_sample": [_i, _i, _i, _vp, _vp, _vp, _vp],
    "hf_farthest_point_sample_variant": [_i, _i, _i, _i, _i, _vp, _vp, _vp, _vp],
    "hf_fps_workspace": [_i, _i],
    "hf_fps_onchip_limit": [],
    "hf_gather_point": [_i, _i, _i, _vp, _vp, _vp, _vp],
    "hf_gather_point_grad": [_i, _i, _i, _vp, _vp, _vp, _vp],
    "hf_query_ball_point": [_i, _i, _i, _f, _i, _vp, _vp, _vp, _vp, _vp],
    "hf_group_point": [_i, _i, _i, _i, _i, _vp, _vp, _vp, _vp],
    "hf_group_point_grad": [_i, _i, _i, _i, _i, _vp, _vp, _vp, _vp],
    "hf_query_ball_group_xyz": [_i, _i, _i, _f, _i, _vp, _vp, _i, _vp, _vp, _vp, _vp],
    "hf_ball_query_workspace": [_i, _i],
    "hf_query_ball_group_xyz_ws": [_i, _i, _i, _i, _f, _i, _vp, _vp, _i, _vp, _vp, _vp, _vp, _sz, _vp],
    "hf_index_inverse": [_i, ctypes.c_longlong, _i, _vp, _vp, _vp, _vp],
    "hf_group_point_grad_gather": [_i, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp],
    "hf_knn_point": [_i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp],
    "hf_knn_workspace": [_i, _i],
    "hf_knn_point_sorted": [_i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _sz, _vp],
    "hf_select_top_k": [_i, _i, _i, _i, _vp, _vp, _vp, _vp],
    "hf_three_nn": [_i, _i, _i, _vp, _vp, _vp, _vp, _vp],
    "hf_three_interpolate": [_i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp],
    "hf_three_interpolate_grad": [_i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp],
    "hf_three_interpolate_cl": [_i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp],
    "hf_three_interpolate_cl_grad": [_i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp],
    "hf_compute_bev_iou": [_i, _vp, _i, _vp, _vp, _vp, _vp],
    "hf_nms_mask": [_vp, _vp, _i, _f, _vp],
    "hf_oriented_nms_workspace": [_i],
    "hf_oriented_nms": [_vp, _i, _f, _vp, _vp, _vp, _sz, _vp],
    "hf_oriented_nms_batched": [_i, _vp, _i, _f, _vp, _vp, _vp, _sz, _vp],
    "hf_pc_crop_and_sample": [_vp] * 6 + [_i] * 6 + [_vp] * 6 + [_vp],
    "hf_pc_crop_and_sample_grad_fts": [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp],
    "hf_bn_workspace": [ctypes.c_longlong, _i],
    "hf_bn_relu_fwd_train": [ctypes.c_longlong, _i, _vp, _vp, _vp, _f, _f, _vp, _vp, _i, _vp, _vp, _vp, _vp, _sz, _vp],
    "hf_bn_relu_fwd_eval": [ctypes.c_longlong, _i, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp],
    "hf_bn_relu_bwd": [ctypes.c_longlong, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _sz, _vp],
    "hf_bn_dropout_fwd_train": [ctypes.c_longlong, _i, _vp, _vp, _vp, _f, _f, _vp, _vp, _i, _f, ctypes.c_ulonglong, _vp, _vp, _vp, _vp, _vp, _vp, _sz,
                                _vp],
    "hf_bn_dropout_bwd": [ctypes.c_longlong, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _f, _vp, _vp, _vp, _vp, _vp, _sz, _vp],
    "hf_narrow_linear_dx": [ctypes.c_longlong, _i, _i, _vp, _vp, _vp, _vp],
    "hf_bn_relu_fwd_train_ld": [ctypes.c_longlong, _i, _vp, _vp, _vp, _f, _f, _vp, _vp, _i, _vp, ctypes.c_longlong, _vp, _vp, _vp, _sz, _vp],
    "hf_bn_relu_bwd_ld": [ctypes.c_longlong, _i, _vp, _vp, ctypes.c_longlong, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _sz, _vp],
    "hf_group_concat": [_i, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp],
    "hf_group_concat_grad": [_i, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp],
    "hf_three_interpolate_concat": [_i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp],
    "hf_three_interpolate_concat_grad": [_i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp],
    "hf_bn_stats": [ctypes.c_longlong, _i, _vp, _f, _f, _vp, _vp, _vp, _vp, _vp, _sz, _vp],
    "hf_three_nn_workspace": [_i, _i],
    "hf_three_nn_sorted": [_i, _i, _i, _vp, _vp, _vp, _vp, _vp, _sz, _vp],
    "hf_three_nn_inverse": [_i, _i, _i, _vp, _vp, _vp, _vp],
    "hf_three_interpolate_cl_grad_gather": [_i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp],
    "hf_bn_relu_maxpool_fwd": [ctypes.c_longlong, _i, _i, _vp, _vp, _vp, _i, _f, _f, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                               _sz, _vp],
    "hf_bn_relu_maxpool_bwd": [ctypes.c_longlong, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                               _sz, _vp],
    "hf_linear_wgrad_workspace": [ctypes.c_longlong, _i, _i],
    "hf_linear_wgrad": [ctypes.c_longlong, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp],
    "hf_linear_bn_fwd_workspace": [_i],
    "hf_linear_bn_fwd": [ctypes.c_longlong, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _f, _f, _vp, _vp, _vp, _vp,
                         _vp, _sz, _vp],
    "hf_bn_relu_bwd_dx": [ctypes.c_longlong, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp],
    "hf_linear_bn_fwd_gather": [ctypes.c_longlong, _i, _i, _vp, _i, ctypes.c_longlong, _vp, _vp, _vp, _vp, _vp, _f, _f, _vp, _vp, _vp, _vp,
                                _vp, _sz, _vp],
    "hf_linear_wgrad_gather": [ctypes.c_longlong, _i, _i, _vp, _vp, _i, ctypes.c_longlong, _vp, _vp, _vp, _vp, _sz, _vp],
    "hf_adam_chunk": [],
    "hf_adam_multi": [_i, _vp, _vp, _vp, _f, _f, _f, _f, _f, _i, _vp],
    "hf_copy_multi_max": [],
    "hf_copy_multi": [_i, _vp, _vp, _vp, _vp],
    "hf_linear_bn_bwd_workspace": [_i],
    "hf_linear_bn_bwd": [ctypes.c_longlong, _i, _i] + [_vp] * 18 + [_vp, _sz, _vp],
    "hf_linear_elu_bn_fwd": [ctypes.c_longlong, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _f, _f, _vp, _vp, _vp, _vp, _vp, _sz, _vp],
    "hf_linear_elu_bn_bwd": [ctypes.c_longlong, _i, _i] + [_vp] * 10 + [_vp, _sz, _vp],
    "hf_lift_elu_bn_fwd_workspace": [_i, _i],
    "hf_lift_elu_bn_fwd": [ctypes.c_longlong, _i, _i, _vp, _vp, _vp, _vp, _f, _f, _vp, _vp, _vp, _vp, _vp, _vp, _f, _f, _vp, _vp, _vp, _vp,
                           _vp, _sz, _vp],
    "hf_lift_elu_fwd_eval": [ctypes.c_longlong, _i, _i] + [_vp] * 8 + [_vp, _sz, _vp],
    "hf_lift_elu_fwd_eval_bn": [ctypes.c_longlong, _i, _i] + [_vp] * 12 + [_vp, _sz, _vp],
    "hf_lift_elu_bn_bwd_workspace": [ctypes.c_longlong, _i, _i],
    "hf_lift_elu_bn_bwd": [ctypes.c_longlong, _i, _i] + [_vp] * 12 + [_vp, _sz, _vp],
    "hf_project_gather": [_i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp],
    "hf_project_gather_grad": [_i, _i, _i, _i, _i, _vp, _vp, _vp, _vp],
    "hf_fuse_concat": [ctypes.c_longlong, _i, _i, _vp, _vp, _vp, _vp, _vp],
    "hf_fuse_concat_grad": [ctypes.c_longlong, _i, _i, _vp, _vp, _vp, _vp, _vp],
    "hf_rpn_loss_workspace": [],
    "hf_rpn_loss_fwd": [ctypes.c_longlong, _i, _i, _i] + [_vp] * 11 + [_f, _f, _f, _vp, _vp, _sz, _vp],
    "hf_rpn_loss_bwd": [ctypes.c_longlong, _i, _i, _i] + [_vp] * 11 + [_f, _f, _f, _vp, _vp, _vp, _vp, _vp],
    "hf_bin_box_decode": [ctypes.c_longlong, _i] + [_vp] * 13 + [_f, _f, _vp, _vp],
    "hf_bin_box_encode": [ctypes.c_longlong, _i, _i] + [_vp] * 7 + [_f, _f, _f, _f] + [_vp] * 8 + [_vp],
    "hf_bin_head_decode": [ctypes.c_longlong, _i, _i, _i, _i] + [_vp] * 6 + [_f, _f, _vp, _vp, _vp],
    "hf_group_point_into": [_i, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp],
    "hf_group_point_grad_from": [_i, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp],
    "hf_xconv_apply": [ctypes.c_longlong, _i, _i, _vp, _vp, _vp, _vp],
    "hf_xconv_apply_grad": [ctypes.c_longlong, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp],
    "hf_depthwise_k": [ctypes.c_longlong, _i, _i, _i, _vp, _vp, _vp, _vp],
    "hf_depthwise_k_grad": [ctypes.c_longlong, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp],
    "hf_depthwise_k_grad_workspace": [ctypes.c_longlong, _i, _i, _i],
    "hf_depthwise_k_grad_ws": [ctypes.c_longlong, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp],
    "hf_xconv_depthwise": [ctypes.c_longlong, _i, _i, _i, _vp, _vp, _vp, _vp, _vp],
    "hf_xconv_depthwise_grad": [ctypes.c_longlong, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "hf_xconv_depthwise_gather": [_i, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "hf_xconv_depthwise_gather_grad": [_i, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                                       _vp, _sz, _vp],
    "hf_xconv_depthwise_gather_grad_workspace": [_i, _i, _i, _i, _i, _i],
    "hf_version": [],
    "hf_strerror": [_i],
    "hf_last_hip_error": [],
}
_RESTYPES = {
    "hf_fps_workspace": _sz,
    "hf_ball_query_workspace": _sz,
    "hf_rpn_loss_workspace": _sz,
    "hf_oriented_nms_workspace": _sz,
    "hf_bn_workspace": _sz,
    "hf_three_nn_workspace": _sz,
    "hf_knn_workspace": _sz,
    "hf_linear_wgrad_workspace": _sz,
    "hf_linear_bn_fwd_workspace": _sz,
    "hf_linear_bn_bwd_workspace": _sz,
    "hf_xconv_depthwise_gather_grad_workspace": _sz,
    "hf_lift_elu_bn_fwd_workspace": _sz,
    "hf_depthwise_k_grad_workspace": _sz,
    "hf_lift_elu_bn_bwd_workspace": _sz,
    "hf_version": ctypes.c_char_p,
    "hf_strerror": ctypes.c_char_p,
}

EXPORTED_SYMBOLS = tuple(sorted(_SIGNATURES))


def build(verbose=False):
    """Compile libhfops.so for gfx950 with hipcc (cross-compiles without a GPU)."""
    out = subprocess.run(["make", "-C", CSRC, "-j8", "all"], capture_output=True, text=True)
    if out.returncode != 0:
        raise RuntimeError("libhfops.so build failed:\n" + out.stdout + out.stderr)
    if verbose:
        print(out.stdout)


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                "heterofusionrcnn_amd: %s is missing. Build it with `python -c 'import __graft_entry__ as g; "
                "g.build()'` or `make -C heterofusionrcnn_amd/csrc`. There is no fallback path." % LIB_PATH)
        L = ctypes.CDLL(LIB_PATH)
        for name, argtypes in _SIGNATURES.items():
            fn = getattr(L, name)  # AttributeError here = header / library mismatch
            fn.argtypes = argtypes
            fn.restype = _RESTYPES.get(name, _i)
        _lib = L
    return _lib


def check(status, opname):
    if status == HF_OK:
        return
    L = lib()
    msg = L.hf_strerror(status).decode()
    if status == HF_EINVAL:
        raise ValueError("%s: %s" % (opname, msg))
    if status == HF_EHIP:
        raise RuntimeError("%s: %s (hipError_t %d)" % (opname, msg, L.hf_last_hip_error()))
    raise RuntimeError("%s: %s" % (opname, msg))


def stream_ptr():
    return torch.cuda.current_stream().cuda_stream


def require(cond, message):
    """Shape/attribute checks that the reference performs with OP_REQUIRES -> InvalidArgumentError."""
    if not cond:
        raise ValueError(message)


def dev_tensor(t, dtype, name):
    """Return a contiguous device tensor of `dtype`; refuse host tensors (no CPU path exists)."""
    if not isinstance(t, torch.Tensor):
        raise TypeError("%s must be a torch.Tensor" % name)
    if not t.is_cuda:
        raise RuntimeError("%s must live on the GPU: heterofusionrcnn_amd has no CPU implementation" % name)
    if t.dtype != dtype:
        raise TypeError("%s must be %s, got %s" % (name, dtype, t.dtype))
    return t.contiguous()


def ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None
