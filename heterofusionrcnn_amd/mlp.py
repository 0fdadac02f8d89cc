"""Fused training-mode BatchNorm(+ReLU) over channel-last rows -- the MLP half of a set-abstraction
level (tf_util.conv2d's batch_norm + relu, hf/core/feature_extractors/tf_util.py:190-203,554-581).

HIP kernels in csrc/mlp.hip behind hf_bn_relu_* (include/hfops.h); no framework fallback on the GPU
path: CUDA tensors always take the HIP kernels."""
import torch
import torch.nn as nn

from . import _lib
from ._lib import check, ptr, stream_ptr


def _workspace(rows, c, device):
    nbytes = _lib.lib().hf_bn_workspace(rows, c)
    return torch.empty((nbytes // 4,), dtype=torch.float32, device=device), nbytes


class _BNReLUTrain(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, running_mean, running_var, eps, momentum, relu):
        rows, c = x.shape
        y = torch.empty_like(x)
        mean = torch.empty((c,), dtype=torch.float32, device=x.device)
        invstd = torch.empty((c,), dtype=torch.float32, device=x.device)
        ws, nbytes = _workspace(rows, c, x.device)
        check(_lib.lib().hf_bn_relu_fwd_train(rows, c, ptr(x), ptr(gamma), ptr(beta), eps, momentum, ptr(running_mean),
                                              ptr(running_var), 1 if relu else 0, ptr(y), ptr(mean), ptr(invstd),
                                              ptr(ws), nbytes, stream_ptr()), "bn_relu_fwd_train")
        ctx.save_for_backward(x, gamma, beta, mean, invstd)
        ctx.relu = relu
        return y

    @staticmethod
    def backward(ctx, dy):
        x, gamma, beta, mean, invstd = ctx.saved_tensors
        rows, c = x.shape
        dy = dy.contiguous()
        dx = torch.empty_like(x)
        dgamma = torch.empty_like(gamma)
        dbeta = torch.empty_like(beta)
        ws, nbytes = _workspace(rows, c, x.device)
        check(_lib.lib().hf_bn_relu_bwd(rows, c, ptr(x), ptr(dy), ptr(gamma), ptr(beta), ptr(mean), ptr(invstd),
                                        1 if ctx.relu else 0, ptr(dx), ptr(dgamma), ptr(dbeta), None, ptr(ws), nbytes,
                                        stream_ptr()), "bn_relu_bwd")
        return dx, dgamma, dbeta, None, None, None, None, None


def _splitk_wgrad(g, x, chunk=None):
    """dW = g^T x for tall-skinny g (R,Cout), x (R,Cin): the reduction over R split into chunks (a batched GEMM
    that fills the chip) instead of one workgroup walking all R rows (see modules._TallSkinnyLinear).
    Chunk sizes from scripts/gemm_pad_probe.py on MI355X: a plain GEMM with R = 65536, 128x196 outputs takes
    230 us, 64 chunks of 1024 rows 46 us."""
    r = x.shape[0]
    if chunk is None:
        chunk = 4096 if r >= 131072 else 1024
    if r < 32 * chunk or g.shape[1] * x.shape[1] > 512 * 512:
        return g.t() @ x
    main = (r // chunk) * chunk
    gw = torch.bmm(g[:main].view(-1, chunk, g.shape[1]).transpose(1, 2), x[:main].view(-1, chunk, x.shape[1])).sum(dim=0)
    if main < r:
        gw = gw + g[main:].t() @ x[main:]
    return gw


def linear_wgrad(grad_z, x, in_bn=None):
    """dW (Cout, Cin) = grad_z^T x on the fp32 MFMA kernel (csrc/gemm.hip): rows cut into chunks, partial tiles
    summed in a fixed order.  in_bn = (gamma, beta, mean, invstd) of the previous layer: x is then that layer's
    pre-BN output and relu(bn(x)) is applied while it is staged."""
    rows, cout = grad_z.shape
    cin = x.shape[1]
    L = _lib.lib()
    nbytes = L.hf_linear_wgrad_workspace(rows, cout, cin)
    ws = torch.empty((nbytes // 4,), dtype=torch.float32, device=x.device)
    dw = torch.empty((cout, cin), dtype=torch.float32, device=x.device)
    g, b, m, i = in_bn if in_bn is not None else (None, None, None, None)
    check(L.hf_linear_wgrad(rows, cout, cin, ptr(grad_z), ptr(x), ptr(g), ptr(b), ptr(m), ptr(i), ptr(dw), ptr(ws), nbytes,
                            stream_ptr()), "linear_wgrad")
    return dw


class _LinearBNReLU(torch.autograd.Function):
    """One autograd node for tf_util.conv2d([1,1], bn=True): z = x W^T + b; y = relu(bn(z)).
    Saves x and z only (y is never needed again); the BN backward pass also returns the column sums of dz,
    which ARE the bias gradient -- no separate reduction over the (R, Cout) gradient tensor."""

    @staticmethod
    def forward(ctx, x, weight, bias, gamma, beta, running_mean, running_var, eps, momentum, relu):
        rows, cout = x.shape[0], weight.shape[0]
        z = torch.addmm(bias, x, weight.t())
        y = torch.empty_like(z)
        mean = torch.empty((cout,), dtype=torch.float32, device=x.device)
        invstd = torch.empty((cout,), dtype=torch.float32, device=x.device)
        ws, nbytes = _workspace(rows, cout, x.device)
        check(_lib.lib().hf_bn_relu_fwd_train(rows, cout, ptr(z), ptr(gamma), ptr(beta), eps, momentum,
                                              ptr(running_mean), ptr(running_var), 1 if relu else 0, ptr(y),
                                              ptr(mean), ptr(invstd), ptr(ws), nbytes, stream_ptr()),
              "bn_relu_fwd_train")
        ctx.save_for_backward(x, weight, z, gamma, beta, mean, invstd)
        ctx.relu = relu
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight, z, gamma, beta, mean, invstd = ctx.saved_tensors
        rows, cout = z.shape
        dy = dy.contiguous()
        dz = torch.empty_like(z)
        dgamma = torch.empty_like(gamma)
        dbeta = torch.empty_like(beta)
        dbias = torch.empty_like(beta)
        ws, nbytes = _workspace(rows, cout, z.device)
        check(_lib.lib().hf_bn_relu_bwd(rows, cout, ptr(z), ptr(dy), ptr(gamma), ptr(beta), ptr(mean), ptr(invstd),
                                        1 if ctx.relu else 0, ptr(dz), ptr(dgamma), ptr(dbeta), ptr(dbias), ptr(ws),
                                        nbytes, stream_ptr()), "bn_relu_bwd")
        dx = dz @ weight if ctx.needs_input_grad[0] else None
        dw = _splitk_wgrad(dz, x)
        return dx, dw, dbias, dgamma, dbeta, None, None, None, None, None


def linear_bn_relu(x, weight, bias, bn):
    """x (R, Cin) -> relu(bn(x W^T + b)) with `bn` a BatchNormReLU module (training mode: fused node)"""
    if bn.training:
        return _LinearBNReLU.apply(x.contiguous(), weight, bias, bn.weight, bn.bias, bn.running_mean, bn.running_var,
                                   bn.eps, bn.momentum, bn.relu)
    return bn(torch.addmm(bias, x, weight.t()))


class BatchNormReLU(nn.Module):
    """BatchNorm over the last dimension of (rows, C) followed by ReLU, one fused op.
    Parameter / buffer names follow nn.BatchNorm1d (weight, bias, running_mean, running_var)."""

    def __init__(self, num_features, eps=1e-3, momentum=0.1, relu=True):
        super().__init__()
        self.num_features, self.eps, self.momentum, self.relu = num_features, eps, momentum, relu
        self.weight = nn.Parameter(torch.ones(num_features))   # gamma: tf.constant_initializer(1.0)
        self.bias = nn.Parameter(torch.zeros(num_features))    # beta: 0
        self.register_buffer("running_mean", torch.zeros(num_features))
        self.register_buffer("running_var", torch.ones(num_features))

    def forward(self, x):
        assert x.dim() == 2 and x.shape[1] == self.num_features
        if not x.is_cuda:
            raise RuntimeError("BatchNormReLU: heterofusionrcnn_amd has no CPU implementation")
        x = x.contiguous()
        if self.training:
            return _BNReLUTrain.apply(x, self.weight, self.bias, self.running_mean, self.running_var, self.eps,
                                      self.momentum, self.relu)
        y = torch.empty_like(x)
        invstd = torch.rsqrt(self.running_var + self.eps)
        check(_lib.lib().hf_bn_relu_fwd_eval(x.shape[0], x.shape[1], ptr(x), ptr(self.weight), ptr(self.bias),
                                             ptr(self.running_mean), ptr(invstd), 1 if self.relu else 0, ptr(y),
                                             stream_ptr()), "bn_relu_fwd_eval")
        return y


class _LinearBNReLUMaxPool(torch.autograd.Function):
    """x (G*K, Cin) -> max over the K rows of each group of relu(bn(x W^T + b)): (G, Cout).
    The tail of a set-abstraction MLP as one node; the (G*K, Cout) normalised activation and its gradient are
    never materialised (the backward pass rebuilds the one-hot dy from the stored arg-max rows)."""

    @staticmethod
    def forward(ctx, x, weight, bias, gamma, beta, running_mean, running_var, eps, momentum, k):
        rows, cout = x.shape[0], weight.shape[0]
        groups = rows // k
        z = torch.addmm(bias, x, weight.t())
        pooled = torch.empty((groups, cout), dtype=torch.float32, device=x.device)
        argmax = torch.empty((groups, cout), dtype=torch.uint8, device=x.device)
        mean = torch.empty((cout,), dtype=torch.float32, device=x.device)
        invstd = torch.empty((cout,), dtype=torch.float32, device=x.device)
        ws, nbytes = _workspace(rows, cout, x.device)
        check(_lib.lib().hf_bn_relu_maxpool_fwd(groups, k, cout, ptr(z), ptr(gamma), ptr(beta), 1, eps, momentum,
                                                ptr(running_mean), ptr(running_var), ptr(mean), ptr(invstd),
                                                ptr(pooled), ptr(argmax), ptr(ws), nbytes, stream_ptr()),
              "bn_relu_maxpool_fwd")
        ctx.save_for_backward(x, weight, z, gamma, beta, mean, invstd, argmax)
        ctx.k = k
        return pooled

    @staticmethod
    def backward(ctx, dpooled):
        x, weight, z, gamma, beta, mean, invstd, argmax = ctx.saved_tensors
        rows, cout = z.shape
        groups = rows // ctx.k
        dpooled = dpooled.contiguous()
        dz = torch.empty_like(z)
        dgamma = torch.empty_like(gamma)
        dbeta = torch.empty_like(beta)
        dbias = torch.empty_like(beta)
        ws, nbytes = _workspace(rows, cout, z.device)
        check(_lib.lib().hf_bn_relu_maxpool_bwd(groups, ctx.k, cout, ptr(z), ptr(dpooled), ptr(argmax), ptr(gamma),
                                                ptr(beta), ptr(mean), ptr(invstd), ptr(dz), ptr(dgamma), ptr(dbeta),
                                                ptr(dbias), ptr(ws), nbytes, stream_ptr()), "bn_relu_maxpool_bwd")
        dx = dz @ weight if ctx.needs_input_grad[0] else None
        dw = _splitk_wgrad(dz, x)
        return dx, dw, dbias, dgamma, dbeta, None, None, None, None, None


def linear_bn_relu_maxpool(x, weight, bias, bn, k):
    """x (G*K, Cin), K rows per group -> (G, Cout); `bn` a BatchNormReLU module with relu=True"""
    assert bn.relu and x.shape[0] % k == 0 and k <= 255
    x = x.contiguous()
    if bn.training:
        return _LinearBNReLUMaxPool.apply(x, weight, bias, bn.weight, bn.bias, bn.running_mean, bn.running_var,
                                          bn.eps, bn.momentum, k)
    z = torch.addmm(bias, x, weight.t())
    groups, cout = x.shape[0] // k, weight.shape[0]
    pooled = torch.empty((groups, cout), dtype=torch.float32, device=x.device)
    invstd = torch.rsqrt(bn.running_var + bn.eps)
    check(_lib.lib().hf_bn_relu_maxpool_fwd(groups, k, cout, ptr(z), ptr(bn.weight), ptr(bn.bias), 0, bn.eps, bn.momentum,
                                            None, None, ptr(bn.running_mean), ptr(invstd), ptr(pooled), None, None, 0,
                                            stream_ptr()), "bn_relu_maxpool_fwd")
    return pooled
