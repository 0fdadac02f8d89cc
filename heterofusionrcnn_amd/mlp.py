"""Fused training-mode BatchNorm(+ReLU) over channel-last rows -- the MLP half of a set-abstraction
level (tf_util.conv2d's batch_norm + relu, hf/core/feature_extractors/tf_util.py:190-203,554-581).

HIP kernels in csrc/mlp.hip behind hf_bn_relu_* (include/hfops.h); no framework fallback on the GPU
path: CUDA tensors always take the HIP kernels."""
import torch
import torch.nn as nn

from . import _lib
from ._lib import check, ptr, stream_ptr


def _on_gpu(*tensors):
    """every entry point below hands raw device pointers to HIP kernels: a host tensor must raise, not fault"""
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError("heterofusionrcnn_amd.mlp: tensors must live on the GPU (there is no CPU implementation)")


def _workspace(rows, c, device):
    nbytes = _lib.lib().hf_bn_workspace(rows, c)
    return torch.empty((nbytes // 4,), dtype=torch.float32, device=device), nbytes


# The training kernels update running_mean / running_var through raw pointers, which bumps no tensor version: every launch
# site that hands those pointers to a kernel calls running_stats_written(), and BatchNormReLU.eval_invstd keys its cache on it.
_STATS_GENERATION = [0]


_DROP_LAYERS = [0]      # BatchNormReLU modules built so far in this process (distinct dropout seeds per layer)


DROPOUT_SALT = None     # tests: a fixed salt instead of the rank (to reproduce another rank's masks in one process)


def _dropout_salt():
    """mixed into the dropout seeds: ranks share every buffer after the parameter broadcast, their masks must still differ"""
    if DROPOUT_SALT is not None:
        return int(DROPOUT_SALT)
    import torch.distributed as dist
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def running_stats_written():
    _STATS_GENERATION[0] += 1


class _BNReLUTrain(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, running_mean, running_var, eps, momentum, relu):
        running_stats_written()          # the kernel below writes the running statistics through raw pointers
        rows, c = x.shape
        y = torch.empty_like(x)
        mean = torch.empty((c,), dtype=torch.float32, device=x.device)
        invstd = torch.empty((c,), dtype=torch.float32, device=x.device)
        ws, nbytes = _workspace(rows, c, x.device)
        check(_lib.lib().hf_bn_relu_fwd_train(rows, c, ptr(x), ptr(gamma), ptr(beta), eps, momentum, ptr(running_mean),
                                              ptr(running_var), int(relu), ptr(y), ptr(mean), ptr(invstd),
                                              ptr(ws), nbytes, stream_ptr()), "bn_relu_fwd_train")
        ctx.save_for_backward(x, gamma, beta, mean, invstd)
        ctx.relu = int(relu)   # mode bits: 1 = ReLU after, 2 = ELU before
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
                                        ctx.relu, ptr(dx), ptr(dgamma), ptr(dbeta), None, ptr(ws), nbytes,
                                        stream_ptr()), "bn_relu_bwd")
        return dx, dgamma, dbeta, None, None, None, None, None


class _BNDropoutTrain(torch.autograd.Function):
    """dropout(bn(act(x))) as ONE node: the keep decisions are recomputed from a per-call seed in the backward passes
    (hf_bn_dropout_fwd_train / _bwd): no mask tensor, no dropout pass in either direction"""

    @staticmethod
    def forward(ctx, x, gamma, beta, running_mean, running_var, eps, momentum, relu, rate, drop_state, salt):
        running_stats_written()
        rows, c = x.shape
        y = torch.empty_like(x)
        mean = torch.empty((c,), dtype=torch.float32, device=x.device)
        invstd = torch.empty((c,), dtype=torch.float32, device=x.device)
        seed = torch.empty((1,), dtype=torch.int64, device=x.device)
        ws, nbytes = _workspace(rows, c, x.device)
        check(_lib.lib().hf_bn_dropout_fwd_train(rows, c, ptr(x), ptr(gamma), ptr(beta), eps, momentum, ptr(running_mean), ptr(running_var),
                                                 int(relu), float(rate), int(salt), ptr(drop_state), ptr(seed), ptr(y), ptr(mean), ptr(invstd),
                                                 ptr(ws), nbytes, stream_ptr()), "bn_dropout_fwd_train")
        ctx.save_for_backward(x, gamma, beta, mean, invstd, seed)
        ctx.relu, ctx.rate = int(relu), float(rate)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, gamma, beta, mean, invstd, seed = ctx.saved_tensors
        rows, c = x.shape
        dy = dy.contiguous()
        dx = torch.empty_like(x)
        dgamma = torch.empty_like(gamma)
        dbeta = torch.empty_like(beta)
        ws, nbytes = _workspace(rows, c, x.device)
        check(_lib.lib().hf_bn_dropout_bwd(rows, c, ptr(x), ptr(dy), ptr(gamma), ptr(beta), ptr(mean), ptr(invstd), ctx.relu, ctx.rate,
                                           ptr(seed), ptr(dx), ptr(dgamma), ptr(dbeta), ptr(ws), nbytes, stream_ptr()), "bn_dropout_bwd")
        return dx, dgamma, dbeta, None, None, None, None, None, None, None, None


def _splitk_wgrad(g, x, chunk=None):
    """dW = g^T x for tall-skinny g (R,Cout), x (R,Cin): the reduction over R split into chunks (a batched GEMM
    that fills the chip) instead of one workgroup walking all R rows (see modules._TallSkinnyLinear).
    Routes by shape from scripts/probes/wgrad_shapes_probe.py on MI355X (us; plain library GEMM / best batched split / MFMA kernel):
      (131072, 64x3) 285 / 26 / 25     (131072, 64x64) 364 / 26 / 22    (131072, 256x320) 372 / 178 (4096-row chunks) / 234
      (16384, 64x24) 115 / 25 / 15      (16384, 256x256) 102 / 27 / 32    (32768, 256x320) 141 / 50 (2048) / 90
      (1048576, 64x3) 1881 / 75 (4096) / 132                              (1048576, 64x64) 1541 / 113 / 110"""
    r = x.shape[0]
    out = g.shape[1] * x.shape[1]
    if chunk is None:
        if 512 <= r <= (1 << 18) and out <= 128 * 128:
            return linear_wgrad(g.contiguous(), x.contiguous())   # small results: the MFMA kernel cuts the rows into chunks itself
        chunk = 4096 if r >= (1 << 17) else (2048 if r >= (1 << 15) else 1024)
    if r < 8 * chunk or out > 512 * 512:
        return linear_wgrad(g.contiguous(), x.contiguous()) if 512 <= r and out <= 512 * 512 else g.t() @ x
    main = (r // chunk) * chunk
    gw = torch.bmm(g[:main].view(-1, chunk, g.shape[1]).transpose(1, 2), x[:main].view(-1, chunk, x.shape[1])).sum(dim=0)
    if main < r:
        gw = gw + g[main:].t() @ x[main:]
    return gw


class _LinearSplitK(torch.autograd.Function):
    """y = x W^T (no bias) with the weight gradient on the split-K path: a plain nn.Linear hands dW = g^T x with a
    million rows and a 64 x 3 ... 256 x 256 result to ONE library workgroup (1.45 ms for the 3 -> 64 lifting layer of an
    X-Conv); _splitk_wgrad cuts the rows into chunks that fill the chip."""

    @staticmethod
    def forward(ctx, x, weight):
        ctx.save_for_backward(x, weight)
        rows, cin = x.shape
        cout = weight.shape[0]
        if _mfma_forward_pays(rows, cin, cout) and cout < 128:
            # narrow outputs (the 76-value box head: 131 072 x 512 -> 76) leave the library at 15 TFLOP/s (650 us);
            # the MFMA forward kernel streams x once (its statistics epilogue is simply not used)
            return linear_bn_fwd(x, weight.contiguous(), _zeros(cout, x.device), _NO_BN, update_running=False)[0]
        return x @ weight.t()

    @staticmethod
    def backward(ctx, g):
        x, weight = ctx.saved_tensors
        g = g.contiguous()
        dx = g @ weight if ctx.needs_input_grad[0] else None
        dw = _splitk_wgrad(g, x) if ctx.needs_input_grad[1] else None
        return dx, dw


class _NoBN:
    eps, momentum, running_mean, running_var = 1e-5, 0.0, None, None


_NO_BN = _NoBN()
_ZEROS = {}


def _zeros(n, device):
    key = (n, str(device))
    if key not in _ZEROS:
        _ZEROS[key] = torch.zeros(n, dtype=torch.float32, device=device)
    return _ZEROS[key]


class _NarrowLinear(torch.autograd.Function):
    """y = x W^T + b for a handful of outputs (the segmentation head: 256 -> 2).  The input gradient g W is a GEMM with
    an inner dimension of 2, which the library runs at 490 us for 131 072 rows; written as Cout scaled additions it is
    one streaming pass (hf_narrow_linear_dx).  Weight gradient on the split-K path."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        ctx.save_for_backward(x, weight)
        return torch.nn.functional.linear(x, weight, bias)

    @staticmethod
    def backward(ctx, g):
        x, weight = ctx.saved_tensors
        g = g.contiguous()
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            check(_lib.lib().hf_narrow_linear_dx(x.shape[0], x.shape[1], weight.shape[0], ptr(g), ptr(weight.contiguous()), ptr(dx),
                                                 stream_ptr()), "narrow_linear_dx")
        dw = _splitk_wgrad(g, x) if ctx.needs_input_grad[1] else None
        db = g.sum(0) if ctx.needs_input_grad[2] else None
        return dx, dw, db


def linear_narrow(x, weight, bias):
    """x (.., Cin) -> (.., Cout) with Cout <= 4"""
    if x.is_cuda and x.dtype == torch.float32 and weight.shape[0] <= 4 and x.numel() // x.shape[-1] >= 2048:
        y = _NarrowLinear.apply(x.reshape(-1, x.shape[-1]).contiguous(), weight, bias)
        return y.reshape(*x.shape[:-1], weight.shape[0])
    return torch.nn.functional.linear(x, weight, bias)


def linear_nobias(x, weight):
    """x (.., Cin) @ weight (Cout, Cin)^T; tall inputs on the device take the split-K weight gradient (from 2048 rows: the
    library's plain weight-gradient GEMM is one workgroup per output tile walking every row -- 115 us for 16384 rows x 64 x 24,
    the per-layer cost of the one-frame-per-GPU step)"""
    if x.is_cuda and x.dtype == torch.float32 and x.numel() // x.shape[-1] >= 2048:
        y = _LinearSplitK.apply(x.reshape(-1, x.shape[-1]).contiguous(), weight)
        return y.reshape(*x.shape[:-1], weight.shape[0])
    return torch.nn.functional.linear(x, weight)


def linear_wgrad(grad_z, x, in_bn=None):
    """dW (Cout, Cin) = grad_z^T x on the fp32 MFMA kernel (csrc/gemm.hip): rows cut into chunks, partial tiles
    summed in a fixed order.  in_bn = (gamma, beta, mean, invstd) of the previous layer: x is then that layer's
    pre-BN output and relu(bn(x)) is applied while it is staged."""
    _on_gpu(grad_z, x)
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


def linear_bn_fwd(x, weight, bias, bn, in_bn=None, keep_act=False, update_running=True):
    """z = act(x) W^T + b and the batch statistics of z in one MFMA pass (csrc/gemm.hip).
    Returns (z, mean, invstd, x_act); `bn` supplies eps / momentum / running buffers (updated in place).
    in_bn = (gamma, beta, mean, invstd) of the previous layer: x is that layer's pre-BN output and relu(bn(x)) is
    applied on load; keep_act also stores that activated input (x_act, else None); update_running=False leaves the
    running estimates alone (inference: the returned batch statistics are then simply not used)."""
    _on_gpu(x, weight, bias)
    if update_running:
        running_stats_written()
    rows, cin = x.shape
    cout = weight.shape[0]
    L = _lib.lib()
    nbytes = L.hf_linear_bn_fwd_workspace(cout)
    ws = torch.empty((nbytes // 4,), dtype=torch.float32, device=x.device)
    z = torch.empty((rows, cout), dtype=torch.float32, device=x.device)
    mean = torch.empty((cout,), dtype=torch.float32, device=x.device)
    invstd = torch.empty((cout,), dtype=torch.float32, device=x.device)
    g, b, m, i = in_bn if in_bn is not None else (None, None, None, None)
    x_act = torch.empty_like(x) if (keep_act and in_bn is not None) else None
    check(L.hf_linear_bn_fwd(rows, cin, cout, ptr(x), ptr(g), ptr(b), ptr(m), ptr(i), ptr(x_act), ptr(weight), ptr(bias),
                             ptr(z), bn.eps, bn.momentum, ptr(bn.running_mean) if update_running else None,
                             ptr(bn.running_var) if update_running else None, ptr(mean),
                             ptr(invstd), ptr(ws), nbytes, stream_ptr()), "linear_bn_fwd")
    return z, mean, invstd, x_act


# The MFMA forward pays off where the layer is bound by memory traffic: enough 128-row tiles to fill the chip and at
# most 7 accumulator tiles per wave (Cout <= 224); wider / shorter layers keep the library GEMM plus a statistics pass
# (scripts/gemm_fwd_probe.py).
FUSED_FWD_MIN_ROWS = 32768
FUSED_FWD_MAX_COUT = 224


def _mfma_forward_pays(rows, cin, cout):
    return rows >= FUSED_FWD_MIN_ROWS and cout <= FUSED_FWD_MAX_COUT and cin <= 1024


def _mfma_backward_pays(rows, cin, cout):
    # the input-gradient kernel holds Cin / 32 accumulator tiles per wave; beyond 5 it spills
    return rows >= FUSED_FWD_MIN_ROWS and cin <= 160 and cout <= 256


def _bn_apply(z, gamma, beta, mean, invstd):
    y = torch.empty_like(z)
    check(_lib.lib().hf_bn_relu_fwd_eval(z.shape[0], z.shape[1], ptr(z), ptr(gamma), ptr(beta), ptr(mean), ptr(invstd), 1,
                                         ptr(y), stream_ptr()), "bn_relu_apply")
    return y


def _bn_stats(z, bn):
    running_stats_written()
    rows, c = z.shape
    mean = torch.empty((c,), dtype=torch.float32, device=z.device)
    invstd = torch.empty((c,), dtype=torch.float32, device=z.device)
    ws, nbytes = _workspace(rows, c, z.device)
    check(_lib.lib().hf_bn_stats(rows, c, ptr(z), bn.eps, bn.momentum, ptr(bn.running_mean), ptr(bn.running_var), ptr(mean),
                                 ptr(invstd), ptr(ws), nbytes, stream_ptr()), "bn_stats")
    return mean, invstd


class _SharedMLPChain(torch.autograd.Function):
    """A whole shared MLP (tf_util.conv2d([1,1], bn=True) x L, pointnet_util.py:156-160) as ONE autograd node,
    optionally ending in the max over the K grouped rows (:163-176).

    Forward: layer l runs hf_linear_bn_fwd on the PREVIOUS layer's pre-BN output -- BN affine + ReLU applied while
    the operand is staged, batch statistics of its own output taken from the accumulators -- so between two layers
    there is no statistics pass and no separate normalisation pass; the activated input is stored once (the weight
    gradient needs it).  The tail is either the fused BN+ReLU+max-pool or one BN+ReLU apply.
    Backward: per layer the fused BN backward (dz, dgamma, dbeta), split-K dW, dX = dz W."""

    @staticmethod
    def forward(ctx, x, pool_k, layers, first, *params):
        """first = (mean0, invstd0): x IS the first layer's pre-BN output (its linear ran in _GatherLinear, on neighbourhoods
        read in place); the node then starts at that layer's BatchNorm and hands dz0 back as the gradient of x"""
        L = _lib.lib()
        n = len(layers)
        saved, cur, in_bn = [], x, None
        for li, layer in enumerate(layers):
            w, b, gamma, beta = params[4 * li:4 * li + 4]
            if li == 0 and first is not None:
                xin, z, mean, invstd = x.new_empty(0), x, first[0], first[1]
            elif _mfma_forward_pays(cur.shape[0], w.shape[1], w.shape[0]):
                z, mean, invstd, x_act = linear_bn_fwd(cur, w, b, layer.bn, in_bn, keep_act=True)
                xin = x_act if x_act is not None else cur
            else:  # library GEMM on the materialised input, statistics in their own pass
                xin = _bn_apply(cur, *in_bn) if in_bn is not None else cur
                z = torch.addmm(b, xin, w.t())
                mean, invstd = _bn_stats(z, layer.bn)
            saved.append((xin, z, mean, invstd))
            cur, in_bn = z, (gamma, beta, mean, invstd)
        gamma, beta, mean, invstd = in_bn
        rows, cout = cur.shape
        if pool_k:
            groups = rows // pool_k
            out = torch.empty((groups, cout), dtype=torch.float32, device=x.device)
            argmax = torch.empty((groups, cout), dtype=torch.uint8, device=x.device)
            check(L.hf_bn_relu_maxpool_fwd(groups, pool_k, cout, ptr(cur), ptr(gamma), ptr(beta), 0, 0.0, 0.0, None, None,
                                           ptr(mean), ptr(invstd), ptr(out), ptr(argmax), None, 0, stream_ptr()),
                  "bn_relu_maxpool_fwd")
        else:
            argmax = None
            out = _bn_apply(cur, gamma, beta, mean, invstd)
        flat = []
        for t in saved:
            flat.extend(t)
        ctx.save_for_backward(*flat, *params, *([argmax] if argmax is not None else []))
        ctx.n, ctx.pool_k, ctx.first = n, pool_k, first is not None
        return out

    @staticmethod
    def backward(ctx, dout):
        L = _lib.lib()
        n, pool_k = ctx.n, ctx.pool_k
        tensors = ctx.saved_tensors
        saved = [tensors[4 * i:4 * i + 4] for i in range(n)]
        params = tensors[4 * n:8 * n]
        argmax = tensors[8 * n] if pool_k else None
        grads = [None] * (4 * n)
        dy = dout.contiguous()
        known = None  # (dgamma, dbeta) of the layer about to be processed, when the layer above already reduced them
        # A bias that feeds a batch norm has an exactly zero gradient: sum_r dz[r, c] = a_c (sum dh - R c1 - c2 sum xhat)
        # with c1 = sum dh / R and sum xhat = 0.  The column sums the kernels can also produce are rounding noise
        # around that zero (what autodiff frameworks return); the exact value is returned and the pass skipped.
        # One zero buffer for the whole node, one slice per layer.
        couts = [params[4 * i].shape[0] for i in range(n)]
        zero_bias = torch.zeros((sum(couts),), dtype=torch.float32, device=dy.device)
        for li in range(n - 1, -1, -1):
            xin, z, mean, invstd = saved[li]
            w, b, gamma, beta = params[4 * li:4 * li + 4]
            rows, cout = z.shape
            cin = w.shape[1]
            gathered = li == 0 and ctx.first                       # this layer's linear lives in _GatherLinear: stop at dz
            need_dx = (li > 0 or ctx.needs_input_grad[0]) and not gathered
            mfma = need_dx and _mfma_backward_pays(rows, cin, cout)
            dbias = zero_bias[sum(couts[:li]):sum(couts[:li + 1])]
            dz, from_dy = None, False
            if known is None:  # reduce + dx passes of the BN backward (the pooled tail has its own pair)
                dz = torch.empty_like(z)
                dgamma, dbeta = torch.empty_like(gamma), torch.empty_like(beta)
                ws, nbytes = _workspace(rows, cout, z.device)
                if li == n - 1 and pool_k:
                    check(L.hf_bn_relu_maxpool_bwd(rows // pool_k, pool_k, cout, ptr(z), ptr(dy), ptr(argmax), ptr(gamma),
                                                   ptr(beta), ptr(mean), ptr(invstd), ptr(dz), ptr(dgamma), ptr(dbeta),
                                                   None, ptr(ws), nbytes, stream_ptr()), "bn_relu_maxpool_bwd")
                else:
                    check(L.hf_bn_relu_bwd(rows, cout, ptr(z), ptr(dy), ptr(gamma), ptr(beta), ptr(mean), ptr(invstd), 1,
                                           ptr(dz), ptr(dgamma), ptr(dbeta), None, ptr(ws), nbytes, stream_ptr()),
                          "bn_relu_bwd")
            else:
                dgamma, dbeta = known
                if mfma:
                    from_dy = True  # dz is rebuilt inside the input-gradient kernel while its operand is staged
                else:
                    dz = torch.empty_like(z)
                    check(L.hf_bn_relu_bwd_dx(rows, cout, ptr(z), ptr(dy), ptr(gamma), ptr(beta), ptr(mean), ptr(invstd),
                                              ptr(dgamma), ptr(dbeta), 1, ptr(dz), stream_ptr()), "bn_relu_bwd_dx")
            known = None
            if mfma:
                wt = w.t().contiguous()
                dx = torch.empty((rows, cin), dtype=torch.float32, device=z.device)
                if from_dy:
                    dz = torch.empty_like(z)
                if li > 0:  # the BN-backward sums of the layer below come out of this kernel's accumulators
                    pz, pmean, pinv = saved[li - 1][1:4]
                    pgamma, pbeta = params[4 * (li - 1) + 2], params[4 * (li - 1) + 3]
                    pdg, pdb = torch.empty_like(pgamma), torch.empty_like(pbeta)
                    nbytes = L.hf_linear_bn_bwd_workspace(cin)
                    ws = torch.empty((nbytes // 4,), dtype=torch.float32, device=z.device)
                    known = (pdg, pdb)
                else:
                    pz = pmean = pinv = pgamma = pbeta = pdg = pdb = ws = None
                    nbytes = 0
                check(L.hf_linear_bn_bwd(rows, cout, cin, ptr(dy if from_dy else dz), ptr(z) if from_dy else None,
                                         ptr(gamma), ptr(beta), ptr(mean), ptr(invstd), ptr(dgamma), ptr(dbeta),
                                         ptr(dz) if from_dy else None, ptr(wt), ptr(dx), ptr(pz), ptr(pgamma), ptr(pbeta),
                                         ptr(pmean), ptr(pinv), ptr(pdg), ptr(pdb), ptr(ws), nbytes, stream_ptr()),
                      "linear_bn_bwd")
                dy = dx
            elif need_dx:
                dy = dz @ w
            if gathered:
                dy = dz
                grads[0:4] = [None, None, dgamma, dbeta]
            else:
                grads[4 * li:4 * li + 4] = [_splitk_wgrad(dz, xin), dbias, dgamma, dbeta]
        return (dy if ctx.needs_input_grad[0] else None, None, None, None, *grads)


def _shared_mlp_eval(layers, x, pool_k, extra, first_done=False):
    """inference twin of _SharedMLPChain.forward: running statistics instead of batch statistics, nothing saved.
    first_done: x is already the first layer's pre-BN output (shared_mlp_grouped)"""
    L = _lib.lib()
    cur, in_bn = x, None
    for i, layer in enumerate(layers):
        w = layer.fc.weight if (i or not extra) else torch.nn.functional.pad(layer.fc.weight, (0, extra))
        bn = layer.bn
        if i == 0 and first_done:
            z = x
        elif _mfma_forward_pays(cur.shape[0], w.shape[1], w.shape[0]):
            z = linear_bn_fwd(cur, w, layer.fc.bias, bn, in_bn, update_running=False)[0]
        else:
            xin = _bn_apply(cur, *in_bn) if in_bn is not None else cur
            z = torch.addmm(layer.fc.bias, xin, w.t())
        cur, in_bn = z, (bn.weight, bn.bias, bn.running_mean, bn.eval_invstd())
    gamma, beta, mean, invstd = in_bn
    if not pool_k:
        return _bn_apply(cur, gamma, beta, mean, invstd)
    rows, cout = cur.shape
    out = torch.empty((rows // pool_k, cout), dtype=torch.float32, device=x.device)
    check(L.hf_bn_relu_maxpool_fwd(rows // pool_k, pool_k, cout, ptr(cur), ptr(gamma), ptr(beta), 0, 0.0, 0.0, None, None,
                                   ptr(mean), ptr(invstd), ptr(out), None, None, 0, stream_ptr()), "bn_relu_maxpool_fwd")
    return out


def _gather_weight(w, c, xyz_first):
    """the first layer's weight (cout, 3 + c) in the reference's column order ([xyz | features], pointnet_util.py:58-60, or
    [features | xyz] in the multi-scale module, :264) -> the gathering kernels' order [features, 0.., x, y, z, 0]"""
    cfp = (c + 3) // 4 * 4
    wx, wf = (w[:, :3], w[:, 3:3 + c]) if xyz_first else (w[:, c:c + 3], w[:, :c])
    out = w.new_zeros((w.shape[0], cfp + 4))
    out[:, :c] = wf
    out[:, cfp:cfp + 3] = wx
    return out


def _ungather_weight(dw_int, c, xyz_first):
    cfp = (c + 3) // 4 * 4
    parts = [dw_int[:, cfp:cfp + 3], dw_int[:, :c]]
    return torch.cat(parts if xyz_first else parts[::-1], dim=1)


class _GatherLinear(torch.autograd.Function):
    """z0 = [points[idx] | grouped_xyz] W^T + b with the batch statistics of z0, the neighbourhoods read IN PLACE
    (hf_linear_bn_fwd_gather / hf_linear_wgrad_gather): the (B,M,K,C+3) tensor of sample_and_group is never written,
    neither for the forward GEMM nor for the weight gradient.  Returns (z0, mean, invstd); the BatchNorm that consumes them
    is the head of _SharedMLPChain (first=...)."""

    @staticmethod
    def forward(ctx, points, idx, gxyz, weight, bias, bn, xyz_first, update_running):
        L = _lib.lib()
        b, m, k = idx.shape
        rows = b * m * k
        c = points.shape[2] if points is not None else 0
        n_src = points.shape[1] if points is not None else 1
        cout = weight.shape[0]
        dev = idx.device
        wi = _gather_weight(weight, c, xyz_first)
        z = torch.empty((rows, cout), dtype=torch.float32, device=dev)
        mean = torch.empty((cout,), dtype=torch.float32, device=dev)
        invstd = torch.empty((cout,), dtype=torch.float32, device=dev)
        nbytes = L.hf_linear_bn_fwd_workspace(cout)
        ws = torch.empty((nbytes // 4,), dtype=torch.float32, device=dev)
        if update_running:
            running_stats_written()
        check(L.hf_linear_bn_fwd_gather(rows, c, cout, ptr(points), n_src, m * k, ptr(idx), ptr(gxyz), ptr(wi), ptr(bias), ptr(z),
                                        bn.eps, bn.momentum, ptr(bn.running_mean) if update_running else None,
                                        ptr(bn.running_var) if update_running else None, ptr(mean), ptr(invstd), ptr(ws), nbytes,
                                        stream_ptr()), "linear_bn_fwd_gather")
        ctx.save_for_backward(idx, gxyz, weight, *([points] if points is not None else []))
        ctx.dims = (b, m, k, c, n_src, cout, xyz_first)
        ctx.mark_non_differentiable(mean, invstd)
        return z, mean, invstd

    @staticmethod
    def backward(ctx, dz, _dmean, _dinvstd):
        L = _lib.lib()
        idx, gxyz, weight, *rest = ctx.saved_tensors
        points = rest[0] if rest else None
        b, m, k, c, n_src, cout, xyz_first = ctx.dims
        rows = b * m * k
        dz = dz.contiguous()
        cin = (c + 3) // 4 * 4 + 4
        nbytes = L.hf_linear_wgrad_workspace(rows, cout, cin)
        ws = torch.empty((nbytes // 4,), dtype=torch.float32, device=dz.device)
        dwi = torch.empty((cout, cin), dtype=torch.float32, device=dz.device)
        check(L.hf_linear_wgrad_gather(rows, cout, c, ptr(dz), ptr(points), n_src, m * k, ptr(idx), ptr(gxyz), ptr(dwi), ptr(ws),
                                       nbytes, stream_ptr()), "linear_wgrad_gather")
        dw = _ungather_weight(dwi, c, xyz_first)
        dpoints = None
        if points is not None and ctx.needs_input_grad[0]:
            wf = (weight[:, 3:3 + c] if xyz_first else weight[:, :c]).contiguous()
            dg = dz @ wf                                               # (rows, c): the gradient of the gathered features
            dpoints = torch.empty((b, n_src, c), dtype=torch.float32, device=dz.device)
            check(L.hf_group_point_grad(b, n_src, c, m, k, ptr(dg), ptr(idx), ptr(dpoints), stream_ptr()), "group_point_grad")
        # the bias feeds a BatchNorm: its gradient is exactly zero (see _SharedMLPChain.backward)
        return dpoints, None, None, dw, torch.zeros((cout,), dtype=torch.float32, device=dz.device), None, None, None


GATHER_MAX_COUT = 256           # accumulator tiles of the MFMA forward kernel
GATHER_MAX_CFEAT = 1020
# below this many feature channels the (B,M,K,C+3) tensor is a few columns wide and cheaper to write once than to gather one
# 4-byte feature per neighbour behind its index (level 1 of config 2, C = 1: +210 us in place; level 2, C = 64: equal time, -170 MB
# of traffic -- profiles/r04_sa_gather_traffic.txt)
GATHER_MIN_CFEAT = 16


def grouped_mlp_fusable(layers, points, idx, min_cfeat=0):
    """shared_mlp_grouped's conditions: BatchNorm + ReLU on every layer, on the device, first layer within the MFMA kernel's
    range (Cout <= 256), pool_k <= 255; min_cfeat: the callers that CHOOSE between the routes pass GATHER_MIN_CFEAT"""
    first = layers[0]
    c = points.shape[2] if points is not None else 0
    return (idx.is_cuda and all(l.bn is not None and l.bn.relu for l in layers) and first.fc.out_features <= GATHER_MAX_COUT
            and min_cfeat <= c <= GATHER_MAX_CFEAT and idx.shape[2] <= 255 and first.fc.in_features == c + 3)


def shared_mlp_grouped(layers, points, idx, grouped_xyz, xyz_first=True, pool=True):
    """The MLP of a set-abstraction level on neighbourhoods read in place (pointnet_util.py:42-64 + 156-176 as one chain):
    points (B,N,C) or None, idx (B,M,K) int32, grouped_xyz (B,M,K,3) centred -> (B*M, Cout) with the max over the K
    neighbours (pool) or (B*M*K, Cout).  The first layer gathers while it stages its operand; the rest is shared_mlp's chain.
    xyz_first: the reference's column order of the first weight, [xyz | features] (single-scale module) or [features | xyz]."""
    layers = list(layers)
    b, m, k = idx.shape
    training = all(l.bn.training for l in layers)
    first = layers[0]
    if training and torch.is_grad_enabled():
        z0, mean0, invstd0 = _GatherLinear.apply(points, idx, grouped_xyz.detach().contiguous(), first.fc.weight, first.fc.bias, first.bn,
                                                 xyz_first, True)
        params = []
        for l in layers:
            params += [l.fc.weight, l.fc.bias, l.bn.weight, l.bn.bias]
        return _SharedMLPChain.apply(z0, k if pool else 0, layers, (mean0, invstd0), *params)
    require_eval = not any(l.bn.training for l in layers)
    assert require_eval, "shared_mlp_grouped: training-mode BatchNorm needs autograd enabled (or call .eval())"
    # inference: running statistics; the first layer still gathers in place (its batch statistics are simply not used)
    with torch.no_grad():
        z0, _, _ = _GatherLinear.apply(points, idx, grouped_xyz.contiguous(), first.fc.weight, first.fc.bias, first.bn, xyz_first, False)
    return _shared_mlp_eval(layers, z0, k if pool else 0, 0, first_done=True)


def shared_mlp(layers, x, pool_k=0):
    """x (R, Cin) through `layers` (SharedMLPLayer-like: .fc, .bn with relu) -> (R, Cout), or (R / pool_k, Cout) with
    the max over each run of pool_k rows.  One fused node in training when every layer has BN+ReLU (each layer on
    the MFMA forward kernel where that pays, else library GEMM + statistics pass); otherwise layer by layer."""
    layers = list(layers)
    _on_gpu(x)
    x = x.contiguous()
    bn_relu = all(l.bn is not None and l.bn.relu for l in layers)
    pool_ok = pool_k == 0 or (pool_k <= 255 and x.shape[0] % pool_k == 0)
    fused = bn_relu and pool_ok and all(l.bn.training for l in layers)
    extra = x.shape[1] - layers[0].fc.in_features  # zero columns appended by the producer (grouping.group_concat)
    assert extra >= 0
    if fused:
        params = []
        for i, l in enumerate(layers):
            w = l.fc.weight if (i or not extra) else torch.nn.functional.pad(l.fc.weight, (0, extra))
            params += [w, l.fc.bias, l.bn.weight, l.bn.bias]
        return _SharedMLPChain.apply(x, pool_k, layers, None, *params)
    if bn_relu and pool_ok and not torch.is_grad_enabled() and not any(l.bn.training for l in layers):
        return _shared_mlp_eval(layers, x, pool_k, extra)
    if extra:
        x = x[:, :layers[0].fc.in_features]
    for l in layers[:-1] if pool_k else layers:
        x = l(x)
    if pool_k:
        last = layers[-1]
        if last.bn is not None and last.bn.relu and pool_k <= 255:
            return linear_bn_relu_maxpool(x, last.fc.weight, last.fc.bias, last.bn, pool_k)
        x = last(x)
        return x.view(-1, pool_k, x.shape[-1]).max(dim=1).values
    return x


class _LinearBNReLU(torch.autograd.Function):
    """One autograd node for tf_util.conv2d([1,1], bn=True): z = x W^T + b; y = relu(bn(z)).
    Saves x and z only (y is never needed again); the BN backward pass also returns the column sums of dz,
    which ARE the bias gradient -- no separate reduction over the (R, Cout) gradient tensor."""

    @staticmethod
    def forward(ctx, x, weight, bias, gamma, beta, running_mean, running_var, eps, momentum, relu):
        running_stats_written()          # the kernel below writes the running statistics through raw pointers
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
    _on_gpu(x, weight, bias)
    if bn.training:
        return _LinearBNReLU.apply(x.contiguous(), weight, bias, bn.weight, bn.bias, bn.running_mean, bn.running_var,
                                   bn.eps, bn.momentum, bn.relu)
    return bn(torch.addmm(bias, x, weight.t()))


class BatchNormReLU(nn.Module):
    """BatchNorm over the last dimension of (rows, C) followed by ReLU, one fused op.
    Parameter / buffer names follow nn.BatchNorm1d (weight, bias, running_mean, running_var)."""

    def __init__(self, num_features, eps=1e-3, momentum=0.1, relu=True, elu_in=False):
        """elu_in: ELU applied to the input on load (pointfly's linear -> ELU -> BatchNorm), fused into every pass"""
        super().__init__()
        self.num_features, self.eps, self.momentum, self.relu = num_features, eps, momentum, relu
        self.elu_in = elu_in
        self.weight = nn.Parameter(torch.ones(num_features))   # gamma: tf.constant_initializer(1.0)
        self.bias = nn.Parameter(torch.zeros(num_features))    # beta: 0
        self.register_buffer("running_mean", torch.zeros(num_features))
        self.register_buffer("running_var", torch.ones(num_features))
        self._eval_invstd = None      # (key, 1/sqrt(running_var + eps)): inference reuses it
        # fused dropout (forward(x, dropout=rate)): [base seed, forward calls so far], advanced on the device by the kernels.  The
        # base seed follows torch.manual_seed and the order in which layers are built (no draw from the generator: the initial
        # weights stay what they were); not part of the state dict.
        _DROP_LAYERS[0] += 1
        base = (torch.initial_seed() * 0x9E3779B97F4A7C15 + _DROP_LAYERS[0] * 0xBF58476D1CE4E5B9) % (1 << 62)
        self.register_buffer("drop_state", torch.tensor([base, 0], dtype=torch.int64), persistent=False)

    def train(self, mode=True):
        self._eval_invstd = None
        return super().train(mode)

    def eval_invstd(self):
        """1/sqrt(running_var + eps), kept between inference calls (two tiny kernels per layer and batch otherwise: 1800 launches
        per batch of the two-stage inference).  The key holds the tensor version (framework writes), the package-wide
        generation counter (raw-pointer writes of the training kernels: running_stats_written) and the storage; nothing is cached while a stream
        capture is running (the graph's pool memory is reused afterwards)."""
        rv = self.running_var
        if torch.cuda.is_current_stream_capturing():
            return torch.rsqrt(rv + self.eps)
        key = (rv._version, _STATS_GENERATION[0], rv.device, rv.data_ptr())
        if self._eval_invstd is None or self._eval_invstd[0] != key:
            self._eval_invstd = (key, torch.rsqrt(rv + self.eps))
        return self._eval_invstd[1]

    def forward(self, x, dropout=0.0):
        """dropout (a rate; callers pass 0 at inference): tf.layers.dropout(rate) applied to the output -- inside the normalisation
        passes when the layer normalises with batch statistics (no pass of its own, no mask tensor), by the framework's dropout
        behind the running-statistics form (a frozen BatchNorm inside a model that still trains)"""
        assert x.dim() == 2 and x.shape[1] == self.num_features
        if not x.is_cuda:
            raise RuntimeError("BatchNormReLU: heterofusionrcnn_amd has no CPU implementation")
        x = x.contiguous()
        if self.training and dropout > 0.0:
            return _BNDropoutTrain.apply(x, self.weight, self.bias, self.running_mean, self.running_var, self.eps, self.momentum,
                                         (1 if self.relu else 0) | (2 if self.elu_in else 0), dropout, self.drop_state, _dropout_salt())
        if self.training:
            return _BNReLUTrain.apply(x, self.weight, self.bias, self.running_mean, self.running_var, self.eps,
                                      self.momentum, (1 if self.relu else 0) | (2 if self.elu_in else 0))
        y = torch.empty_like(x)
        invstd = self.eval_invstd()
        check(_lib.lib().hf_bn_relu_fwd_eval(x.shape[0], x.shape[1], ptr(x), ptr(self.weight), ptr(self.bias),
                                             ptr(self.running_mean), ptr(invstd),
                                             (1 if self.relu else 0) | (2 if self.elu_in else 0), ptr(y),
                                             stream_ptr()), "bn_relu_fwd_eval")
        return torch.nn.functional.dropout(y, p=dropout, training=True) if dropout > 0.0 else y


class _LinearBNReLUMaxPool(torch.autograd.Function):
    """x (G*K, Cin) -> max over the K rows of each group of relu(bn(x W^T + b)): (G, Cout).
    The tail of a set-abstraction MLP as one node; the (G*K, Cout) normalised activation and its gradient are
    never materialised (the backward pass rebuilds the one-hot dy from the stored arg-max rows)."""

    @staticmethod
    def forward(ctx, x, weight, bias, gamma, beta, running_mean, running_var, eps, momentum, k):
        running_stats_written()          # the kernel below writes the running statistics through raw pointers
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
    _on_gpu(x, weight, bias)
    x = x.contiguous()
    if bn.training:
        return _LinearBNReLUMaxPool.apply(x, weight, bias, bn.weight, bn.bias, bn.running_mean, bn.running_var,
                                          bn.eps, bn.momentum, k)
    z = torch.addmm(bias, x, weight.t())
    groups, cout = x.shape[0] // k, weight.shape[0]
    pooled = torch.empty((groups, cout), dtype=torch.float32, device=x.device)
    invstd = bn.eval_invstd()
    check(_lib.lib().hf_bn_relu_maxpool_fwd(groups, k, cout, ptr(z), ptr(bn.weight), ptr(bn.bias), 0, bn.eps, bn.momentum,
                                            None, None, ptr(bn.running_mean), ptr(invstd), ptr(pooled), None, None, 0,
                                            stream_ptr()), "bn_relu_maxpool_fwd")
    return pooled
