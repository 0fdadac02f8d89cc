"""PointCNN backbone (the `pc_pointcnn` extractor of hf/configs/rpn_multiclass.config:62-118) on the HIP ops.

  xconv        hf/core/feature_extractors/pointcnn.py:16-151   Algorithm 1 of the PointCNN paper
  PointCNN     pointcnn.py:186-388   encoder of xconv layers (FPS down-sampling), decoder of xdconv layers (an xconv from
               a coarser layer onto a finer one, concat with the finer layer's encoder features, dense), fc layers
  layers       hf/core/pointfly.py:371-497   dense / conv2d / depthwise_conv2d / separable_conv2d, every one
               linear (no bias) -> ELU -> BatchNorm(momentum 0.99, eps 1e-3)

The neighbour search is the HIP kNN kernel (hf_knn_point_sorted: grid binning + ring search, ties to the lower index)
instead of the reference's dense (B,P,N) distance matrix + tf.nn.top_k (pointfly.py:185-212: 1 GiB per frame at
P = N = 16384); the neighbourhoods are gathered by hf_group_point (+ its scatter gradient); the sampling is the HIP FPS.
The convolutions with a (1,K) window over the K neighbours are written as the matrix products they are:

  conv2d(.., K*K, (1,K)) on (B,P,K,3)            = Linear(3K -> K*K) on the flattened [k][c] window        (X_0)
  depthwise_conv2d(.., K, (1,K)) on (B,P,K,K)    = out[c*K + m] = sum_w in[w][c] * W[w][c][m]                (X_1, X_2)
  separable_conv2d(.., C, (1,K), dm) on (B,P,K,Cin) = depthwise out[c*dm + m] = sum_k in[k][c] * Wd[k][c][m],
                                                      then Linear(Cin*dm -> C)                              (fts_conv)

The weight layouts are TensorFlow's (HWIO / depthwise HW, in, multiplier) so that a checkpoint would map one to one;
tests/test_pointcnn.py checks the einsum forms against torch.nn.functional.conv2d with the same weights.
TensorFlow cannot be imported here: parity unpinned against reference outputs.
"""
import contextlib
import math
from dataclasses import dataclass
from typing import Optional, Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib
from ._lib import check, ptr, require, stream_ptr
from .grouping import INVERSE_MAX_TARGETS, concat_group, group_point, index_inverse, knn_point
from .sampling import farthest_point_sample, gather_point
from .mlp import BatchNormReLU, linear_nobias, running_stats_written


class EluBN(nn.Module):
    """activation (ELU or none) then tf.layers.batch_normalization(momentum 0.99, epsilon 1e-3) over the last dimension.
    The ELU is applied on load inside the HIP BatchNorm passes (statistics, normalisation, both backward passes): the
    activated tensor never exists in memory."""

    def __init__(self, channels, activation=True):
        super().__init__()
        self.bn = BatchNormReLU(channels, eps=1e-3, momentum=0.01, relu=False, elu_in=activation)
        self.activation = activation

    def forward(self, x, dropout=0.0):
        shape = x.shape
        return self.bn(x.reshape(-1, shape[-1]), dropout).reshape(shape)


class Dense(nn.Module):
    """pf.dense: Linear without bias -> ELU -> BN"""

    def __init__(self, cin, cout, activation=True):
        super().__init__()
        self.linear = nn.Linear(cin, cout, bias=False)
        nn.init.xavier_normal_(self.linear.weight)
        self.post = EluBN(cout, activation)

    def forward(self, x, dropout=0.0):
        """dropout: the rate of the tf.layers.dropout that follows this layer (training mode: fused into its BatchNorm)"""
        return self.post(linear_nobias(x, self.linear.weight), dropout)


class _DenseChainElu(torch.autograd.Function):
    """Two pf.dense layers (linear -> ELU -> BatchNorm, twice: the lifting of the local coordinates, pointcnn.py:96-99) as one
    node on the fp32 MFMA kernels.  Forward: each layer's batch statistics come out of its GEMM's accumulators (no statistics
    pass), the first layer's normalisation is applied while the second GEMM stages its operand (no normalisation pass; the
    normalised activation is stored once, for the weight gradient).  Backward: the BatchNorm-backward sums of the first layer
    come out of the input-gradient GEMM of the second (no reduction pass)."""

    @staticmethod
    def forward(ctx, x, w0, g0, b0, w1, g1, b1, bn0, bn1):
        L = _lib.lib()
        rows = x.shape[0]
        c0, c1 = w0.shape[0], w1.shape[0]
        dev = x.device

        def layer(inp, w, bn, in_bn, want_act):
            running_stats_written()
            cout, cin = w.shape
            nbytes = L.hf_linear_bn_fwd_workspace(cout)
            ws = torch.empty((nbytes // 4,), dtype=torch.float32, device=dev)
            z = torch.empty((rows, cout), dtype=torch.float32, device=dev)
            mean, invstd = torch.empty((cout,), dtype=torch.float32, device=dev), torch.empty((cout,), dtype=torch.float32, device=dev)
            act = torch.empty_like(inp) if want_act else None
            g, b, m, i = in_bn if in_bn is not None else (None, None, None, None)
            check(L.hf_linear_elu_bn_fwd(rows, cin, cout, ptr(inp), ptr(g), ptr(b), ptr(m), ptr(i), ptr(act), ptr(w), ptr(z), bn.eps,
                                         bn.momentum, ptr(bn.running_mean), ptr(bn.running_var), ptr(mean), ptr(invstd), ptr(ws), nbytes,
                                         stream_ptr()), "linear_elu_bn_fwd")
            return z, mean, invstd, act

        w0, w1 = w0.contiguous(), w1.contiguous()
        z0, m0, i0, _ = layer(x, w0, bn0, None, False)
        z1, m1, i1, y0 = layer(z0, w1, bn1, (g0, b0, m0, i0), True)
        out = torch.empty_like(z1)
        check(L.hf_bn_relu_fwd_eval(rows, c1, ptr(z1), ptr(g1), ptr(b1), ptr(m1), ptr(i1), 2, ptr(out), stream_ptr()), "bn_elu_apply")
        ctx.save_for_backward(x, z0, m0, i0, y0, z1, m1, i1, w0, g0, b0, w1, g1, b1)
        return out

    @staticmethod
    def backward(ctx, dout):
        from .mlp import _splitk_wgrad
        L = _lib.lib()
        x, z0, m0, i0, y0, z1, m1, i1, w0, g0, b0, w1, g1, b1 = ctx.saved_tensors
        rows, c0, c1 = x.shape[0], w0.shape[0], w1.shape[0]
        dout = dout.contiguous()
        dz1 = torch.empty_like(z1)
        dg1, db1 = torch.empty_like(g1), torch.empty_like(b1)
        ws, nbytes = _bn_ws(rows, c1, x.device)
        check(L.hf_bn_relu_bwd(rows, c1, ptr(z1), ptr(dout), ptr(g1), ptr(b1), ptr(m1), ptr(i1), 2, ptr(dz1), ptr(dg1), ptr(db1), None,
                               ptr(ws), nbytes, stream_ptr()), "bn_elu_bwd")
        dy0 = torch.empty_like(z0)
        dg0, db0 = torch.empty_like(g0), torch.empty_like(b0)
        nbytes = L.hf_linear_bn_bwd_workspace(c0)
        ws = torch.empty((nbytes // 4,), dtype=torch.float32, device=x.device)
        w1t = w1.t().contiguous()
        check(L.hf_linear_elu_bn_bwd(rows, c1, c0, ptr(dz1), ptr(w1t), ptr(dy0), ptr(z0), ptr(g0), ptr(b0), ptr(m0), ptr(i0), ptr(dg0),
                                     ptr(db0), ptr(ws), nbytes, stream_ptr()), "linear_elu_bn_bwd")
        dz0 = torch.empty_like(z0)
        check(L.hf_bn_relu_bwd_dx(rows, c0, ptr(z0), ptr(dy0), ptr(g0), ptr(b0), ptr(m0), ptr(i0), ptr(dg0), ptr(db0), 2, ptr(dz0),
                                  stream_ptr()), "bn_elu_bwd_dx")
        dx = dz0 @ w0 if ctx.needs_input_grad[0] else None
        return dx, _splitk_wgrad(dz0, x), dg0, db0, _splitk_wgrad(dz1, y0), dg1, db1, None, None


class _LiftChain(torch.autograd.Function):
    """The lifting chain on a THREE-channel input (the local coordinates, pointcnn.py:96-99) without its first layer's output in
    memory: the statistics of elu(W0 x) from one pass over x, y0 = BN0(elu(W0 x)) rebuilt from x inside the second GEMM's operand
    staging, inside the weight-gradient GEMM and inside the two backward passes (hf_lift_elu_bn_fwd / _bwd).  Of the chain's
    rows x C tensors only z1 (pre-activation of the second layer), the output and dz1 are ever written."""

    @staticmethod
    def forward(ctx, x, w0, g0, b0, w1, g1, b1, bn0, bn1):
        running_stats_written()
        L = _lib.lib()
        rows, c0, c1, dev = x.shape[0], w0.shape[0], w1.shape[0], x.device
        w0, w1 = w0.contiguous(), w1.contiguous()
        new = lambda *shape: torch.empty(shape, dtype=torch.float32, device=dev)
        m0, i0, m1, i1, z1 = new(c0), new(c0), new(c1), new(c1), new(rows, c1)
        nbytes = L.hf_lift_elu_bn_fwd_workspace(c0, c1)
        ws = new(nbytes // 4)
        check(L.hf_lift_elu_bn_fwd(rows, c0, c1, ptr(x), ptr(w0), ptr(g0), ptr(b0), bn0.eps, bn0.momentum, ptr(bn0.running_mean),
                                   ptr(bn0.running_var), ptr(m0), ptr(i0), ptr(w1), ptr(z1), bn1.eps, bn1.momentum,
                                   ptr(bn1.running_mean), ptr(bn1.running_var), ptr(m1), ptr(i1), ptr(ws), nbytes, stream_ptr()),
              "lift_elu_bn_fwd")
        out = torch.empty_like(z1)
        check(L.hf_bn_relu_fwd_eval(rows, c1, ptr(z1), ptr(g1), ptr(b1), ptr(m1), ptr(i1), 2, ptr(out), stream_ptr()), "bn_elu_apply")
        ctx.save_for_backward(x, m0, i0, z1, m1, i1, w0, g0, b0, w1, g1, b1)
        return out

    @staticmethod
    def backward(ctx, dout):
        L = _lib.lib()
        x, m0, i0, z1, m1, i1, w0, g0, b0, w1, g1, b1 = ctx.saved_tensors
        rows, c0, c1, dev = x.shape[0], w0.shape[0], w1.shape[0], x.device
        dout = dout.contiguous()
        dz1 = torch.empty_like(z1)
        dg1, db1 = torch.empty_like(g1), torch.empty_like(b1)
        ws, nbytes = _bn_ws(rows, c1, dev)
        check(L.hf_bn_relu_bwd(rows, c1, ptr(z1), ptr(dout), ptr(g1), ptr(b1), ptr(m1), ptr(i1), 2, ptr(dz1), ptr(dg1), ptr(db1), None,
                               ptr(ws), nbytes, stream_ptr()), "bn_elu_bwd")
        dw0_t = torch.empty((3, c0), dtype=torch.float32, device=dev)
        dw1 = torch.empty_like(w1)
        dg0, db0 = torch.empty_like(g0), torch.empty_like(b0)
        nbytes = L.hf_lift_elu_bn_bwd_workspace(rows, c0, c1)
        ws = torch.empty((nbytes // 4,), dtype=torch.float32, device=dev)
        w1t = w1.t().contiguous()
        check(L.hf_lift_elu_bn_bwd(rows, c0, c1, ptr(x), ptr(w0), ptr(g0), ptr(b0), ptr(m0), ptr(i0), ptr(dz1), ptr(w1t), ptr(dw0_t),
                                   ptr(dw1), ptr(dg0), ptr(db0), ptr(ws), nbytes, stream_ptr()), "lift_elu_bn_bwd")
        return None, dw0_t.t(), dg0, db0, dw1, dg1, db1, None, None


def _lift_chain_eval(d0, d1, x):
    """inference twin of _LiftChain on the 3-channel input: the first layer's output is rebuilt from x inside the second GEMM with
    its RUNNING statistics, the second layer's running statistics are applied by one pass; nothing else is written"""
    L = _lib.lib()
    rows = x.numel() // 3
    bn0, bn1 = d0.post.bn, d1.post.bn
    c0, c1 = d0.linear.out_features, d1.linear.out_features
    x2 = x.reshape(rows, 3).contiguous()
    z1 = torch.empty((rows, c1), dtype=torch.float32, device=x.device)
    nbytes = L.hf_lift_elu_bn_fwd_workspace(c0, c1)
    ws = torch.empty((nbytes // 4,), dtype=torch.float32, device=x.device)
    w0, w1 = d0.linear.weight.contiguous(), d1.linear.weight.contiguous()
    # the second layer's a (elu(z1) - running_mean) + beta leaves the GEMM's epilogue: no normalisation pass
    check(L.hf_lift_elu_fwd_eval_bn(rows, c0, c1, ptr(x2), ptr(w0), ptr(bn0.weight), ptr(bn0.bias), ptr(bn0.running_mean),
                                    ptr(bn0.eval_invstd()), ptr(w1), ptr(bn1.weight), ptr(bn1.bias), ptr(bn1.running_mean),
                                    ptr(bn1.eval_invstd()), ptr(z1), ptr(ws), nbytes, stream_ptr()), "lift_elu_fwd_eval_bn")
    return z1.reshape(*x.shape[:-1], c1)


def dense_chain(d0, d1, x):
    """d1(d0(x)) for two Dense modules; tall batches on the device take the one-node MFMA route"""
    from .mlp import FUSED_FWD_MIN_ROWS
    rows = x.numel() // x.shape[-1]
    bn0, bn1 = d0.post.bn, d1.post.bn
    c0, c1 = d0.linear.out_features, d1.linear.out_features
    if (x.is_cuda and x.dtype == torch.float32 and not bn0.training and not bn1.training and not torch.is_grad_enabled()
            and d0.post.activation and d1.post.activation and x.shape[-1] == 3 and rows >= FUSED_FWD_MIN_ROWS and c0 <= 256
            and c0 % 4 == 0 and c1 <= 256):
        return _lift_chain_eval(d0, d1, x)
    if (x.is_cuda and x.dtype == torch.float32 and bn0.training and bn1.training and d0.post.activation and d1.post.activation
            and rows >= FUSED_FWD_MIN_ROWS and c0 <= 160 and c1 <= 224 and x.shape[-1] <= 1024):
        if x.shape[-1] == 3 and c0 % 4 == 0 and not x.requires_grad:
            out = _LiftChain.apply(x.reshape(rows, 3).contiguous(), d0.linear.weight, bn0.weight, bn0.bias, d1.linear.weight, bn1.weight,
                                   bn1.bias, bn0, bn1)
            return out.reshape(*x.shape[:-1], c1)
        out = _DenseChainElu.apply(x.reshape(rows, x.shape[-1]).contiguous(), d0.linear.weight, bn0.weight, bn0.bias, d1.linear.weight,
                                   bn1.weight, bn1.bias, bn0, bn1)
        return out.reshape(*x.shape[:-1], c1)
    return d1(d0(x))


def _bn_mode(bn):
    return (1 if bn.relu else 0) | (2 if bn.elu_in else 0)


def _bn_ws(rows, c, device):
    nbytes = _lib.lib().hf_bn_workspace(rows, c)
    return torch.empty((nbytes // 4,), dtype=torch.float32, device=device), nbytes


class _BNConcatGroup(torch.autograd.Function):
    """[bn(elu(z)) | points[idx]] for z (B*M*K, Ch), points (B,N,C), idx (B,M,K) -> (B,M,K,Ch+C): the BatchNorm of the lifting
    layer writes its output straight into the concat buffer of pointcnn.py:104 (row stride Ch+C) and the gather fills the
    rest; backward reads both halves of the concat's gradient in place.  No tf.concat copy in either direction."""

    @staticmethod
    def forward(ctx, z, gamma, beta, running_mean, running_var, eps, momentum, mode, points, idx, offsets, entries):
        running_stats_written()          # the kernel below writes the running statistics through raw pointers
        L = _lib.lib()
        b, n, c = points.shape
        _, m, ns = idx.shape
        rows, ch = z.shape
        width = ch + c
        out = torch.empty((b, m, ns, width), dtype=torch.float32, device=z.device)
        mean = torch.empty((ch,), dtype=torch.float32, device=z.device)
        invstd = torch.empty((ch,), dtype=torch.float32, device=z.device)
        ws, nbytes = _bn_ws(rows, ch, z.device)
        check(L.hf_bn_relu_fwd_train_ld(rows, ch, ptr(z), ptr(gamma), ptr(beta), eps, momentum, ptr(running_mean), ptr(running_var),
                                        mode, ptr(out), width, ptr(mean), ptr(invstd), ptr(ws), nbytes, stream_ptr()), "bn_relu_fwd_train_ld")
        check(L.hf_group_point_into(b, n, c, m, ns, width, ch, ptr(points), ptr(idx), ptr(out), stream_ptr()), "group_point_into")
        ctx.save_for_backward(z, gamma, beta, mean, invstd, idx, *([offsets, entries] if offsets is not None else []))
        ctx.meta = (b, n, c, m, ns, ch, mode)
        return out

    @staticmethod
    def backward(ctx, go):
        L = _lib.lib()
        z, gamma, beta, mean, invstd, idx = ctx.saved_tensors[:6]
        b, n, c, m, ns, ch, mode = ctx.meta
        rows, width = z.shape[0], ch + c
        go = go.contiguous()
        dz = torch.empty_like(z)
        dgamma, dbeta = torch.empty_like(gamma), torch.empty_like(beta)
        ws, nbytes = _bn_ws(rows, ch, z.device)
        check(L.hf_bn_relu_bwd_ld(rows, ch, ptr(z), ptr(go), width, ptr(gamma), ptr(beta), ptr(mean), ptr(invstd), mode, ptr(dz),
                                  ptr(dgamma), ptr(dbeta), None, ptr(ws), nbytes, stream_ptr()), "bn_relu_bwd_ld")
        g = None
        if ctx.needs_input_grad[8]:
            g = torch.empty((b, n, c), dtype=torch.float32, device=z.device)
            if len(ctx.saved_tensors) == 8:
                offsets, entries = ctx.saved_tensors[6:]
                check(L.hf_group_point_grad_gather(b, n, c, m, ns, width, ch, ptr(go), ptr(offsets), ptr(entries), ptr(g), stream_ptr()),
                      "group_point_grad_gather")
            else:
                check(L.hf_group_point_grad_from(b, n, c, m, ns, width, ch, ptr(go), ptr(idx), ptr(g), stream_ptr()),
                      "group_point_grad_from")
        return dz, dgamma, dbeta, None, None, None, None, None, g, None, None, None


class _BNConcatSkip(torch.autograd.Function):
    """[bn(elu(z)) | skip] for z (R, C), skip (R, Cs) -> (R, C+Cs): the X-DeConv output normalised straight into the concat with
    the encoder's skip features (pointcnn.py:349); backward reads its half of the gradient in place, the skip's half is a view."""

    @staticmethod
    def forward(ctx, z, gamma, beta, running_mean, running_var, eps, momentum, mode, skip):
        running_stats_written()          # the kernel below writes the running statistics through raw pointers
        L = _lib.lib()
        rows, c = z.shape
        cs = skip.shape[1]
        out = torch.empty((rows, c + cs), dtype=torch.float32, device=z.device)
        mean = torch.empty((c,), dtype=torch.float32, device=z.device)
        invstd = torch.empty((c,), dtype=torch.float32, device=z.device)
        ws, nbytes = _bn_ws(rows, c, z.device)
        check(L.hf_bn_relu_fwd_train_ld(rows, c, ptr(z), ptr(gamma), ptr(beta), eps, momentum, ptr(running_mean), ptr(running_var),
                                        mode, ptr(out), c + cs, ptr(mean), ptr(invstd), ptr(ws), nbytes, stream_ptr()), "bn_relu_fwd_train_ld")
        out[:, c:].copy_(skip)
        ctx.save_for_backward(z, gamma, beta, mean, invstd)
        ctx.meta = (c, cs, mode)
        return out

    @staticmethod
    def backward(ctx, go):
        L = _lib.lib()
        z, gamma, beta, mean, invstd = ctx.saved_tensors
        c, cs, mode = ctx.meta
        rows = z.shape[0]
        go = go.contiguous()
        dz = torch.empty_like(z)
        dgamma, dbeta = torch.empty_like(gamma), torch.empty_like(beta)
        ws, nbytes = _bn_ws(rows, c, z.device)
        check(L.hf_bn_relu_bwd_ld(rows, c, ptr(z), ptr(go), c + cs, ptr(gamma), ptr(beta), ptr(mean), ptr(invstd), mode, ptr(dz),
                                  ptr(dgamma), ptr(dbeta), None, ptr(ws), nbytes, stream_ptr()), "bn_relu_bwd_ld")
        return dz, dgamma, dbeta, None, None, None, None, None, (go[:, c:] if ctx.needs_input_grad[8] else None)


def _fusable(bn, ref, ch, width):
    """the strided BatchNorm passes: training mode on the device; 16-byte rows on both sides when the channels are vectorised"""
    return bn.training and ref.is_cuda and ref.dtype == torch.float32 and (ch % 4 != 0 or width % 4 == 0)


_XAPPLY_K = (4, 8)                                             # K with a HIP kernel (csrc/xconv.hip); shipped configs: 8
_DEPTHWISE_KM = {(8, 1), (8, 2), (8, 3), (8, 4), (8, 8), (4, 1), (4, 4)}


def _hip_ok(*tensors):
    return all(t.is_cuda and t.dtype == torch.float32 for t in tensors)


class _DepthwiseK(torch.autograd.Function):
    """hf_depthwise_k / hf_depthwise_k_grad: one pass over x, no permuted temporaries (the einsum route copies x twice)"""

    @staticmethod
    def forward(ctx, x, weight):
        k, c, m = weight.shape
        x2 = x.reshape(-1, k, c).contiguous()
        w = weight.contiguous()
        y = torch.empty((x2.shape[0], c * m), dtype=torch.float32, device=x.device)
        check(_lib.lib().hf_depthwise_k(x2.shape[0], k, c, m, ptr(x2), ptr(w), ptr(y), stream_ptr()), "depthwise_k")
        ctx.save_for_backward(x2, w)
        ctx.lead = tuple(x.shape[:-2])
        return y.reshape(*ctx.lead, c * m)

    @staticmethod
    def backward(ctx, gy):
        x2, w = ctx.saved_tensors
        k, c, m = w.shape
        gy = gy.reshape(-1, c * m).contiguous()
        gx = torch.empty_like(x2) if ctx.needs_input_grad[0] else None
        gw = torch.empty_like(w) if ctx.needs_input_grad[1] else None
        L = _lib.lib()
        nbytes = L.hf_depthwise_k_grad_workspace(x2.shape[0], k, c, m) if gw is not None else 0
        ws = torch.empty((nbytes // 4,), dtype=torch.float32, device=x2.device) if nbytes else None
        check(L.hf_depthwise_k_grad_ws(x2.shape[0], k, c, m, ptr(x2), ptr(w), ptr(gy), ptr(gx), ptr(gw), ptr(ws), nbytes, stream_ptr()),
              "depthwise_k_grad")
        return (gx.reshape(*ctx.lead, k, c) if gx is not None else None), gw


def depthwise_k(x, weight):
    """x (.., K, C), weight (K, C, M) (TensorFlow's depthwise filter (1, K, C, M) without the unit height) -> (.., C*M):
    a VALID depthwise convolution with a (1, K) window over a width-K input leaves one position, channel c*M + m.
    On the device with a supported (K, M): the HIP kernel; otherwise the einsum form (same values to fp32 rounding)."""
    k, c, m = weight.shape
    if _hip_ok(x, weight) and (k, m) in _DEPTHWISE_KM and x.shape[-2] == k and x.shape[-1] == c:
        return _DepthwiseK.apply(x, weight)
    y = torch.einsum("...wc,wcm->...cm", x, weight)
    return y.reshape(*y.shape[:-2], -1)


class _XApply(torch.autograd.Function):
    """hf_xconv_apply / hf_xconv_apply_grad: F_X = X x F_* per representative point (pointcnn.py:133)"""

    @staticmethod
    def forward(ctx, x, f):
        k, c = f.shape[-2], f.shape[-1]
        x2 = x.reshape(-1, k, k).contiguous()
        f2 = f.reshape(-1, k, c).contiguous()
        out = torch.empty_like(f2)
        check(_lib.lib().hf_xconv_apply(f2.shape[0], k, c, ptr(x2), ptr(f2), ptr(out), stream_ptr()), "xconv_apply")
        ctx.save_for_backward(x2, f2)
        ctx.shapes = (tuple(x.shape), tuple(f.shape))
        return out.reshape(f.shape)

    @staticmethod
    def backward(ctx, go):
        x2, f2 = ctx.saved_tensors
        k, c = f2.shape[1], f2.shape[2]
        go = go.reshape(-1, k, c).contiguous()
        gx = torch.empty_like(x2) if ctx.needs_input_grad[0] else None
        gf = torch.empty_like(f2) if ctx.needs_input_grad[1] else None
        check(_lib.lib().hf_xconv_apply_grad(f2.shape[0], k, c, ptr(x2), ptr(f2), ptr(go), ptr(gx), ptr(gf), stream_ptr()),
              "xconv_apply_grad")
        sx, sf = ctx.shapes
        return (gx.reshape(sx) if gx is not None else None), (gf.reshape(sf) if gf is not None else None)


class _XConvDepthwise(torch.autograd.Function):
    """hf_xconv_depthwise (+ grad): F_X = X x F_* and the depthwise half of the separable convolution in one pass; the
    (B,P,K,C) product is neither stored in the forward pass nor read back in the backward pass (recomputed in registers)"""

    @staticmethod
    def forward(ctx, x, f, wd):
        k, c, m = wd.shape
        x2 = x.reshape(-1, k, k).contiguous()
        f2 = f.reshape(-1, k, c).contiguous()
        w = wd.contiguous()
        out = torch.empty((f2.shape[0], c * m), dtype=torch.float32, device=f.device)
        check(_lib.lib().hf_xconv_depthwise(f2.shape[0], k, c, m, ptr(x2), ptr(f2), ptr(w), ptr(out), stream_ptr()),
              "xconv_depthwise")
        ctx.save_for_backward(x2, f2, w)
        ctx.shapes = (tuple(x.shape), tuple(f.shape))
        return out.reshape(*f.shape[:-2], c * m)

    @staticmethod
    def backward(ctx, go):
        x2, f2, w = ctx.saved_tensors
        k, c, m = w.shape
        go = go.reshape(-1, c * m).contiguous()
        gx = torch.empty_like(x2) if ctx.needs_input_grad[0] else None
        gf = torch.empty_like(f2) if ctx.needs_input_grad[1] else None
        gw = torch.empty_like(w) if ctx.needs_input_grad[2] else None
        check(_lib.lib().hf_xconv_depthwise_grad(f2.shape[0], k, c, m, ptr(x2), ptr(f2), ptr(w), ptr(go), ptr(gx), ptr(gf),
                                                 ptr(gw), stream_ptr()), "xconv_depthwise_grad")
        sx, sf = ctx.shapes
        return (gx.reshape(sx) if gx is not None else None), (gf.reshape(sf) if gf is not None else None), gw


class _XConvDepthwiseGather(torch.autograd.Function):
    """hf_xconv_depthwise_gather (+ grad): the same pass with F_* = [F_delta | fts[idx]] never materialised -- the gathered block
    (K x the feature table: 1.1 GB at the last decoder layers) is read in place through the neighbour table, and its
    gradient reaches the table in gather form through the table's CSR inverse"""

    @staticmethod
    def forward(ctx, x, f_delta, fts, idx, offsets, entries, wd, use_workspace=True):
        ctx.use_workspace = use_workspace
        k, c, m = wd.shape
        b, p = f_delta.shape[0], f_delta.shape[1]
        c0, c1, n = f_delta.shape[-1], fts.shape[-1], fts.shape[1]
        x2, f2, t2, w = x.reshape(-1, k, k).contiguous(), f_delta.reshape(-1, k, c0).contiguous(), fts.contiguous(), wd.contiguous()
        out = torch.empty((b * p, c * m), dtype=torch.float32, device=fts.device)
        idx = idx.contiguous()
        check(_lib.lib().hf_xconv_depthwise_gather(b, n, p, k, c0, c1, m, ptr(x2), ptr(f2), ptr(t2), ptr(idx), ptr(w), ptr(out),
                                                   stream_ptr()), "xconv_depthwise_gather")
        ctx.save_for_backward(x2, f2, t2, idx, w, *([offsets, entries] if offsets is not None else []))
        ctx.dims = (b, p, n, tuple(x.shape), tuple(f_delta.shape))
        return out.reshape(b, p, c * m)

    @staticmethod
    def backward(ctx, go):
        x2, f2, t2, idx, w, *inv = ctx.saved_tensors
        b, p, n, sx, sf = ctx.dims
        k, c, m = w.shape
        c0, c1 = f2.shape[-1], t2.shape[-1]
        go = go.reshape(-1, c * m).contiguous()
        gx = torch.empty_like(x2) if ctx.needs_input_grad[0] else None
        gf = torch.empty_like(f2) if ctx.needs_input_grad[1] else None
        gt = torch.empty_like(t2) if ctx.needs_input_grad[2] else None
        gw = torch.empty_like(w) if ctx.needs_input_grad[6] else None
        require(gt is None or inv, "xconv_depthwise_gather: the gradient of the feature table needs the inverse neighbour table")
        off, ent = inv if inv else (None, None)
        L = _lib.lib()
        # the gathered block's gradient is staged once in a workspace (freed right after) and summed per table row; the row
        # chunks' partial weight gradients go through it as well (fixed-order sum instead of atomics)
        nbytes = L.hf_xconv_depthwise_gather_grad_workspace(b, p, k, c0, c1, m) if ctx.use_workspace else 0
        ws = torch.empty((nbytes // 4,), dtype=torch.float32, device=go.device) if nbytes else None
        check(L.hf_xconv_depthwise_gather_grad(b, n, p, k, c0, c1, m, ptr(x2), ptr(f2), ptr(t2), ptr(idx), ptr(w), ptr(go),
                                               ptr(off), ptr(ent), ptr(gx), ptr(gf), ptr(gt), ptr(gw), ptr(ws), nbytes, stream_ptr()),
              "xconv_depthwise_gather_grad")
        return (gx.reshape(sx) if gx is not None else None), (gf.reshape(sf) if gf is not None else None), gt, None, None, None, gw, None


_XDW_KM = {(8, 1), (8, 2), (8, 3), (8, 4), (4, 1), (4, 4), (12, 1), (12, 2)}


def xconv_depthwise_gather(x, f_delta, fts, idx, wd, inverse=None, use_workspace=True):
    """xconv_depthwise(x, [f_delta | fts[idx]], wd) without the concatenation in memory.  x (B,P,K,K), f_delta (B,P,K,C0),
    fts (B,N,C1), idx (B,P,K) int32, wd (K, C0+C1, M); inverse = index_inverse(idx, N) when fts needs a gradient"""
    off, ent = inverse if inverse is not None else (None, None)
    return _XConvDepthwiseGather.apply(x, f_delta, fts, idx, off, ent, wd, use_workspace)


def _gather_fusable(x, c0, fts, wd, inverse):
    """the fused kernel's conditions: 64-aligned lifted block, supported (K, M), and -- when the feature table needs a
    gradient -- the inverse neighbour table to collect it with"""
    k, c, m = wd.shape
    return (_hip_ok(x, fts, wd) and (k, m) in _XDW_KM and c0 % 64 == 0 and c0 + fts.shape[-1] == c
            and (inverse is not None or not (torch.is_grad_enabled() and fts.requires_grad)))


def xconv_depthwise(x, f, wd):
    """depthwise_k(x @ f, wd): x (.., K, K), f (.., K, C), wd (K, C, M) -> (.., C*M)"""
    k, c, m = wd.shape
    if _hip_ok(x, f, wd) and (k, m) in _XDW_KM and f.shape[-2] == k and f.shape[-1] == c and x.shape[-1] == x.shape[-2] == k:
        return _XConvDepthwise.apply(x, f, wd)
    return depthwise_k(x_apply(x, f), wd)


def x_apply(x, f):
    """x (.., K, K), f (.., K, C) -> (.., K, C) = x @ f"""
    if _hip_ok(x, f) and f.shape[-2] in _XAPPLY_K and x.shape[-1] == x.shape[-2] == f.shape[-2]:
        return _XApply.apply(x, f)
    return torch.matmul(x, f)


class DepthwiseK(nn.Module):
    """pf.depthwise_conv2d(x, K, (1,K)) on x (.., K, C): weight (K_w, C, M), out (.., C*M) with out[c*M + m]"""

    def __init__(self, k, channels, multiplier, activation=True):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(k, channels, multiplier))
        nn.init.xavier_normal_(self.weight.view(k, channels * multiplier))
        self.post = EluBN(channels * multiplier, activation)

    def forward(self, x):
        return self.post(depthwise_k(x, self.weight))


class SeparableK(nn.Module):
    """pf.separable_conv2d(x, C, (1,K), depth_multiplier): depthwise over the K neighbours, pointwise to C, ELU, BN
    (tf.layers.separable_conv2d applies the activation after the pointwise convolution only)"""

    def __init__(self, k, cin, cout, multiplier):
        super().__init__()
        self.depthwise = nn.Parameter(torch.empty(k, cin, multiplier))
        nn.init.xavier_normal_(self.depthwise.view(k, cin * multiplier))
        self.pointwise = nn.Linear(cin * multiplier, cout, bias=False)
        nn.init.xavier_normal_(self.pointwise.weight)
        self.post = EluBN(cout, True)

    def forward(self, x):                                    # (B,P,K,Cin)
        return self.post(linear_nobias(depthwise_k(x, self.depthwise), self.pointwise.weight))


# The X-transformation branch of an X-Conv (x0 -> x1 -> x2 on (B,P,K*3): rows = P) and its lifting branch (lift0 -> lift1 on
# (B,P,K,3): rows = P*K) read COORDINATES only -- the neighbours' offsets P' = P - p -- and meet the feature chain only in the
# X x F_* product (pointcnn.py:96-133).  With CONCURRENT_X_BRANCH each of the two is enqueued on its own side HIP stream: their
# small launches (forward + backward: autograd runs a node's backward on the stream its forward ran on) overlap the feature chain
# instead of queueing inside it; inside a captured step the forks and joins become edges of the graph.  When the caller has forked
# the side streams once, after the geometry is in place (PointCnnBackbone.forward: `ahead`), the branches of LATER layers do not
# wait for the feature chain of earlier ones at all: they run ahead, and the chain finds X and F_delta ready.  Off by default;
# graph_step.TrainStep / bench.py switch it on where the kernels of one branch do not fill the chip.
CONCURRENT_X_BRANCH = False
CONCURRENT_X_BRANCH_MIN_POINTS = 0      # only layers with at least this many output points (B*P) fork
CONCURRENT_LIFT_BRANCH_MAX_ROWS = 65536  # (with CONCURRENT_X_BRANCH) the lifting branch of layers with at most this many rows (B*P*K) on a second side stream
CONCURRENT_RUN_AHEAD = False            # (with CONCURRENT_X_BRANCH) fork once per forward pass, not once per layer
_side_streams = {}


def _side_stream(device, which=0):
    key = (device.index if device.index is not None else torch.cuda.current_device(), which)
    if key not in _side_streams:
        _side_streams[key] = torch.cuda.Stream(device=device)
    return _side_streams[key]


def fork_side_streams(device):
    """the side streams of the coordinate-only branches pick up from the current stream (call once the geometry is in place)"""
    cur = torch.cuda.current_stream(device)
    for which in (0, 1):
        _side_stream(device, which).wait_stream(cur)


class XConv(nn.Module):
    """pointcnn.py:16-151.  pts (B,N,3), fts (B,N,Cprev) or None, qrs (B,P,3) -> (B,P,C) (+ C//4 with_global)"""

    def __init__(self, k, dilation, c_prev, c, c_pts_fts, depth_multiplier, with_x=True, with_global=False):
        super().__init__()
        self.k, self.d = k, dilation
        self.lift0 = Dense(3, c_pts_fts)
        self.lift1 = Dense(c_pts_fts, c_pts_fts)
        cin = c_pts_fts + c_prev
        self.with_x = with_x
        if with_x:
            self.x0 = Dense(3 * k, k * k)                    # conv2d (1,K) over the [k][c] window
            self.x1 = DepthwiseK(k, k, k)
            self.x2 = DepthwiseK(k, k, k, activation=False)
        self.conv = SeparableK(k, cin, c, depth_multiplier)
        self.with_global = with_global
        if with_global:
            self.g0 = Dense(3, c // 4)
            self.g1 = Dense(c // 4, c // 4)
        self.out_channel = c + (c // 4 if with_global else 0)

    def _x_transform(self, local, b, p, k):
        """pointcnn.py:107-131: conv2d (1,K) over the [k][c] window, two depthwise (1,K) convolutions -> (B,P,K,K)"""
        x = self.x0(local.reshape(b, p, 1, k * 3)).reshape(b, p, k, k)
        x = self.x1(x).reshape(b, p, k, k)
        return self.x2(x).reshape(b, p, k, k)

    def neighbours(self, pts, qrs):
        """knn_indices_general(qrs, pts, K*D)[:, :, ::D] (pointcnn.py:72-73): coordinates only"""
        with torch.no_grad():
            _, idx = knn_point(self.k * self.d, pts, qrs)
            return idx[:, :, ::self.d].contiguous() if self.d > 1 else idx

    def forward(self, pts, fts, qrs, idx=None, inverse=None, skip=None, ahead=False, local=None):
        """inverse = index_inverse(idx, N) prepared with the geometry: the gradient of the feature gather then gathers too.
        skip (B,P,Cs): returns [x-conv output | skip] (the concat of an X-DeConv with the encoder features, pointcnn.py:349).
        ahead: the caller forked the side streams after pts / qrs / idx were in place (fork_side_streams).
        local (B,P,K,3) = the neighbours' offsets P' = P - p (pointcnn.py:96), coordinates only: prepared with the geometry
        (PointCnnBackbone.geometry), else gathered here -- by each of the two branches that read it"""
        idx = idx if idx is not None else self.neighbours(pts, qrs)
        b, p, k = idx.shape
        offsets = (lambda: local) if local is not None else (lambda: group_point(pts, idx) - qrs.unsqueeze(2))
        bn1 = self.lift1.post.bn
        c_delta = self.lift1.linear.out_features
        x = side = lside = None
        concurrent = self.with_x and CONCURRENT_X_BRANCH and pts.is_cuda and b * p >= CONCURRENT_X_BRANCH_MIN_POINTS
        if concurrent:
            # fork: the X-transformation on a side stream, from the offsets (made there too) to the (B,P,K,K) matrices
            cur = torch.cuda.current_stream(pts.device)
            side = _side_stream(pts.device)
            if not ahead:
                side.wait_stream(cur)
            with torch.cuda.stream(side):
                x = self._x_transform(offsets(), b, p, k)
            if b * p * k <= CONCURRENT_LIFT_BRANCH_MAX_ROWS:
                lside = _side_stream(pts.device, 1)
                if not ahead:
                    lside.wait_stream(cur)
        gather = self.with_x and fts is not None and _gather_fusable(pts, c_delta, fts, self.conv.depthwise, inverse)
        fuse_concat = (not gather) and fts is not None and _fusable(bn1, fts, c_delta, c_delta + fts.shape[-1])
        # the coordinate-only part of the lifting branch (on its side stream when there is one) ...
        with torch.cuda.stream(lside) if lside is not None else contextlib.nullcontext():
            local = offsets()                                    # (B,P,K,3)  P' <- P - p
            if gather:
                f = dense_chain(self.lift0, self.lift1, local)    # F_delta alone: the neighbours' features are read in place below
            elif fuse_concat:
                f = linear_nobias(self.lift0(local), self.lift1.linear.weight)      # z of the second lifting layer
            else:
                f = self.lift1(self.lift0(local))                 # F_delta
        if lside is not None:                                  # join: this stream's next kernel reads f
            cur.wait_stream(lside)
            f.record_stream(cur)
        # ... and the part that reads the previous layer's features
        if fuse_concat:
            # F_* <- [F_delta, F]: the second lifting layer's BatchNorm writes into the concat, the gather fills the rest
            off, ent = inverse if inverse is not None else (None, None)
            f = _BNConcatGroup.apply(f.reshape(-1, f.shape[-1]), bn1.weight, bn1.bias, bn1.running_mean, bn1.running_var, bn1.eps,
                                     bn1.momentum, _bn_mode(bn1), fts.contiguous(), idx, off, ent)
        elif not gather and fts is not None:
            f = concat_group(f, fts, idx, inverse)             # F_* <- [F_delta, F]: gathered straight into the concat
        if self.with_x:
            if side is not None:                               # join: this stream's next kernel reads x
                cur.wait_stream(side)
                x.record_stream(cur)
            else:
                x = self._x_transform(local, b, p, k)
            # F_X <- X x F_*, then the depthwise half of the separable convolution, in one pass
            fx = (xconv_depthwise_gather(x, f, fts, idx, self.conv.depthwise, inverse) if gather
                  else xconv_depthwise(x, f, self.conv.depthwise))
            zc = linear_nobias(fx, self.conv.pointwise.weight)
            bnc = self.conv.post.bn
            if skip is not None and not self.with_global and _fusable(bnc, zc, zc.shape[-1], zc.shape[-1] + skip.shape[-1]):
                out = _BNConcatSkip.apply(zc.reshape(-1, zc.shape[-1]), bnc.weight, bnc.bias, bnc.running_mean, bnc.running_var, bnc.eps,
                                          bnc.momentum, _bn_mode(bnc), skip.reshape(-1, skip.shape[-1]).contiguous())
                return out.reshape(b, p, -1)
            out = self.conv.post(zc)
        else:
            out = self.conv(f)                                # (B,P,C)
        if self.with_global:
            out = torch.cat([self.g1(self.g0(qrs)), out], dim=-1)
        return out if skip is None else torch.cat([out, skip], dim=-1)


@dataclass
class PointCnnConfig:
    """xconv_param = (K, D, P, C); xdconv_param = (K, D, pts_layer_idx, qrs_layer_idx) -- rpn_multiclass.config:67-112"""
    xconv: Tuple[Tuple[int, int, int, int], ...] = ((8, 1, -1, 256), (8, 1, 4096, 256), (8, 1, 1024, 512), (8, 1, 256, 1024),
                                                    (8, 1, 64, 1024))
    xdconv: Tuple[Tuple[int, int, int, int], ...] = ((8, 1, 4, 4), (8, 1, 4, 3), (8, 1, 3, 2), (8, 1, 2, 1), (8, 1, 1, 0),
                                                     (8, 1, 0, 0))
    fc: Tuple[Tuple[int, float], ...] = ((256, 0.5), (256, 0.5))
    in_channel: int = 1
    with_x: bool = True
    with_global: bool = True


class PointCnnBackbone(nn.Module):
    """pointcnn.py:186-388 (sampling 'fps', multi_scale_grouping off)"""

    def __init__(self, cfg: PointCnnConfig = PointCnnConfig()):
        super().__init__()
        self.cfg = cfg
        self.enc = nn.ModuleList()
        chans = [cfg.in_channel]
        for li, (k, d, p, c) in enumerate(cfg.xconv):
            if li == 0:
                c_pts, dm, c_prev = (c // 2 if cfg.in_channel == 0 else c // 4), 4, cfg.in_channel      # :259-261
            else:
                c_before = cfg.xconv[li - 1][3]
                c_pts, dm, c_prev = c_before // 4, math.ceil(c / c_before), chans[-1]                     # :262-265
            m = XConv(k, d, c_prev, c, c_pts, dm, cfg.with_x, cfg.with_global and li == len(cfg.xconv) - 1)
            self.enc.append(m)
            chans.append(m.out_channel)
        self.dec = nn.ModuleList()
        self.fuse = nn.ModuleList()
        c_last = chans[-1]
        for li, (k, d, pi, qi) in enumerate(cfg.xdconv):
            c = cfg.xconv[qi][3]
            c_prev = cfg.xconv[pi][3]
            fts_c = chans[pi + 1] if li == 0 else c_last                                                 # :318-322
            self.dec.append(XConv(k, d, fts_c, c, c_prev // 4, 1, cfg.with_x, False))                     # :326-348
            self.fuse.append(Dense(c + chans[qi + 1], c))                                                # :349-352
            c_last = c
        if not cfg.xdconv:
            c_last = chans[-1]
        self.fc = nn.ModuleList()
        self.fc_drop = []
        for (w, rate) in cfg.fc:
            self.fc.append(Dense(c_last, w))
            self.fc_drop.append(rate)
            c_last = w
        self.out_channel = c_last

    def geometry(self, xyz):
        """layer points (FPS) and every neighbour table: functions of the coordinates alone"""
        with torch.no_grad():
            pts = [xyz]
            enc_idx = []
            for li, (k, d, p, c) in enumerate(self.cfg.xconv):
                cur = pts[-1]
                same = p == -1 or (li > 0 and p == self.cfg.xconv[li - 1][2])                             # :215-218
                qrs = cur if same else gather_point(cur, farthest_point_sample(p, cur))
                enc_idx.append(self.enc[li].neighbours(cur, qrs))
                pts.append(qrs)
            dec_idx = [self.dec[li].neighbours(pts[pi + 1], pts[qi + 1]) for li, (k, d, pi, qi) in enumerate(self.cfg.xdconv)]
            # the inverse of every neighbour table (for each data point the (query, slot) pairs that name it): the gradient of
            # the feature gather then writes every row once instead of scattering atomics
            inv = lambda idx, n: index_inverse(idx, n) if (self.training and n <= INVERSE_MAX_TARGETS) else None   # a backward pass only
            enc_inv = [inv(ix, pts[li].shape[1]) for li, ix in enumerate(enc_idx)]
            dec_inv = [inv(ix, pts[pi + 1].shape[1]) for ix, (k, d, pi, qi) in zip(dec_idx, self.cfg.xdconv)]
            # the neighbours' offsets P' = P - p of every layer (pointcnn.py:96): read by the lifting branch AND by the
            # X-transformation branch of the layer -- two gathers and two subtractions per layer inside the step otherwise
            enc_local = [group_point(pts[li], ix) - pts[li + 1].unsqueeze(2) for li, ix in enumerate(enc_idx)]
            dec_local = [group_point(pts[pi + 1], ix) - pts[qi + 1].unsqueeze(2) for ix, (k, d, pi, qi) in zip(dec_idx, self.cfg.xdconv)]
        return {"pts": pts, "enc": enc_idx, "dec": dec_idx, "enc_inv": enc_inv, "dec_inv": dec_inv, "enc_local": enc_local,
                "dec_local": dec_local}

    def forward(self, xyz, features, geometry=None, taps=None):
        """taps (a list): filled with (output, detached copy) of every encoder layer -- each tensor that flows from the encoder
        into the decoder; the decoder then reads the copies (see below)"""
        g = geometry if geometry is not None else self.geometry(xyz)
        pts = g["pts"]
        fts = [features]
        ahead = CONCURRENT_X_BRANCH and CONCURRENT_RUN_AHEAD and xyz.is_cuda
        if ahead:
            fork_side_streams(xyz.device)      # the geometry is in place: the coordinate-only branches of every layer may start
        for li, m in enumerate(self.enc):
            fts.append(m(pts[li], fts[-1], pts[li + 1], g["enc"][li], g.get("enc_inv", [None] * len(self.enc))[li], ahead=ahead,
                         local=g.get("enc_local", [None] * len(self.enc))[li]))
        if taps is not None:
            # a clean cut: the decoder (and everything after it) reads detached copies that are leaves of their own graph, so
            # d loss / d copy holds the downstream paths only; the encoder's own chain (layer i+1 reads layer i) stays on the
            # originals, through which graph_step.TrainStep later pushes those gradients
            cuts = [t.detach().requires_grad_(True) for t in fts[1:]]
            taps.extend(zip(fts[1:], cuts))
            fts = [fts[0]] + cuts
        cur = None
        for li, (k, d, pi, qi) in enumerate(self.cfg.xdconv):
            src = fts[pi + 1] if li == 0 else cur
            x = self.dec[li](pts[pi + 1], src, pts[qi + 1], g["dec"][li], g.get("dec_inv", [None] * len(self.dec))[li], skip=fts[qi + 1],
                             ahead=ahead, local=g.get("dec_local", [None] * len(self.dec))[li])
            cur = self.fuse[li](x)                                   # x = [x-deconv | encoder features of the query layer]
        out = cur if cur is not None else fts[-1]        # no decoder (the RCNN's extractor): the last encoder layer
        for layer, rate in zip(self.fc, self.fc_drop):
            out = layer(out, dropout=rate if self.training else 0.0)                                      # dense -> dropout, :371-384
        return out
