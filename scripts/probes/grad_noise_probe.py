"""Diagnostic: config-2 stack at full size, parameter gradients of (a) the fused HIP path, (b) the op-by-op torch fp32
form, each against (c) the op-by-op form in fp64 -- how much of the fused-vs-torch difference is fp32 noise of either."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import heterofusionrcnn_amd as hf
from heterofusionrcnn_amd import modules
from heterofusionrcnn_amd.modules import three_nn_weights
from bench import kitti_uniform
F = torch.nn.functional
torch.manual_seed(5)
rng = np.random.default_rng(5)
xyz = torch.from_numpy(kitti_uniform(rng, 8, 16384)).cuda()
inten = torch.from_numpy(rng.uniform(-0.5, 0.5, (8, 16384, 1)).astype(np.float32)).cuda()
model = modules.PointnetSAFPStack(in_channel=1).cuda().train()
g_out = torch.from_numpy(np.random.default_rng(6).standard_normal((8, 16384, model.out_channel)).astype(np.float32)).cuda() / 16384.0


def stack_ref(dt):
    cast = lambda t: t.to(dt)
    def layer_ref(layer, x):
        z = F.linear(x, cast(layer.fc.weight), cast(layer.fc.bias))
        mu, var = z.mean(0), z.var(0, unbiased=False)
        return torch.relu((z - mu) / torch.sqrt(var + 1e-3) * cast(layer.bn.weight) + cast(layer.bn.bias))
    xyzs, feats = [xyz], [cast(inten)]
    for m in model.sa:
        new_xyz = hf.gather_point(xyzs[-1], hf.farthest_point_sample(m.npoint, xyzs[-1]))
        idx, _ = hf.query_ball_point(m.radius, m.nsample, xyzs[-1], new_xyz)
        li = idx.long()
        b_, n_, k_ = idx.shape
        gat = lambda t: torch.gather(t, 1, li.reshape(b_, -1, 1).expand(-1, -1, t.shape[-1])).reshape(b_, n_, k_, -1)
        g = torch.cat([cast(gat(xyzs[-1]) - new_xyz.unsqueeze(2)), gat(feats[-1])], -1)
        x = g.reshape(-1, g.shape[-1])
        for layer in m.mlp:
            x = layer_ref(layer, x)
        xyzs.append(new_xyz)
        feats.append(x.reshape(b_, n_, k_, -1).max(2).values)
    up = feats[-1]
    for i, m in enumerate(model.fp):
        d = len(model.sa) - 1 - i
        dist, idx = hf.three_nn(xyzs[d], xyzs[d + 1])
        w = cast(three_nn_weights(dist))
        li = idx.long()
        b_, n_, _ = idx.shape
        gath = torch.gather(up, 1, li.reshape(b_, -1, 1).expand(-1, -1, up.shape[-1])).reshape(b_, n_, 3, -1)
        interp = (gath * w.unsqueeze(-1)).sum(2)
        x = torch.cat([interp, feats[d]], 2)
        x = x.reshape(-1, x.shape[-1])
        for layer in m.mlp:
            x = layer_ref(layer, x)
        up = x.reshape(b_, n_, -1)
    return up


params = list(model.parameters())
from heterofusionrcnn_amd import mlp as _mlp
if os.environ.get("NO_MFMA_BWD"):
    _mlp._mfma_backward_pays = lambda *a: False
if os.environ.get("NO_MFMA_FWD"):
    _mlp._mfma_forward_pays = lambda *a: False
if os.environ.get("NO_SPLITK"):
    _mlp._splitk_wgrad = lambda g, x, chunk=None: g.t() @ x
if os.environ.get("NO_CHAIN"):
    # layer-by-layer nodes (library GEMM + hf_bn_relu_* passes per layer) instead of the one-node shared MLP
    def nochain(x, pool_k, layers, *params):
        n = len(layers)
        for i, l in enumerate(layers):
            w, b = params[4 * i], params[4 * i + 1]
            args = (x, w, b, l.bn.weight, l.bn.bias, l.bn.running_mean, l.bn.running_var, l.bn.eps, l.bn.momentum)
            x = _mlp._LinearBNReLUMaxPool.apply(*args, pool_k) if (pool_k and i == n - 1) else _mlp._LinearBNReLU.apply(*args, True)
        return x
    _mlp._SharedMLPChain.apply = staticmethod(nochain)
if os.environ.get("TORCH_BN"):
    # ... and the BatchNorm of every layer by torch ops (two-pass variance, autograd's backward)
    def torchbn(x, pool_k, layers, *params):
        n = len(layers)
        for i, l in enumerate(layers):
            w, b = params[4 * i], params[4 * i + 1]
            z = F.linear(x, w, b)
            mu, var = z.mean(0), z.var(0, unbiased=False)
            x = torch.relu((z - mu) / torch.sqrt(var + l.bn.eps) * l.bn.weight + l.bn.bias)
        return x.view(-1, pool_k, x.shape[-1]).max(dim=1).values if pool_k else x
    _mlp._SharedMLPChain.apply = staticmethod(torchbn)
out = model(xyz, inten, geometry=model.geometry(xyz))
r32 = stack_ref(torch.float32)
r64 = stack_ref(torch.float64)
if os.environ.get("LOSS") == "smooth":
    # a loss whose gradient is a smooth function of the output (mean squared distance to a fixed per-channel target) instead of
    # a random cotangent: the column sums of the backward pass then do not cancel, and one flipped ReLU mask is one part in 10^6
    tgt = torch.linspace(-1.0, 1.0, model.out_channel, device="cuda")
    lossf = lambda o: ((o - tgt.to(o.dtype)) ** 2).mean()
    ga = torch.autograd.grad(lossf(out), params, allow_unused=True)
    gb = torch.autograd.grad(lossf(r32), params, allow_unused=True)
    gc = torch.autograd.grad(lossf(r64), params, allow_unused=True)
else:
    ga = torch.autograd.grad(out, params, g_out, allow_unused=True)
    gb = torch.autograd.grad(r32, params, g_out, allow_unused=True)
    gc = torch.autograd.grad(r64, params, g_out.double(), allow_unused=True)
print("output: fused vs fp64 %.3e, torch32 vs fp64 %.3e" % (float((out - r64).abs().max()), float((r32 - r64).abs().max())))
print("%-26s %10s %12s %12s" % ("parameter", "scale", "fused-fp64", "torch32-fp64"))
for (n, _), a, b, c in zip(model.named_parameters(), ga, gb, gc):
    if n.endswith("fc.bias"):
        continue
    c32 = c.float()
    z = torch.zeros_like(c32)
    a = z if a is None else a
    b = z if b is None else b
    sc = float(c32.abs().max())
    print("%-26s %10.3e %12.3e %12.3e   rel %6.3f%% %6.3f%%" % (n, sc, float((a - c32).abs().max()), float((b - c32).abs().max()),
                                                               100 * float((a - c32).abs().max()) / sc, 100 * float((b - c32).abs().max()) / sc))
