"""the pooled tail of a set-abstraction MLP in isolation: linear -> BN -> ReLU -> max over the K grouped rows.  (x, d pooled)
captured from the op-by-op fp32 graph of the config-2 stack; the HIP node (_LinearBNReLUMaxPool) and torch fp32 against torch fp64"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from heterofusionrcnn_amd import modules, mlp as _mlp
from bench import kitti_uniform
F = torch.nn.functional
torch.manual_seed(5)
rng = np.random.default_rng(5)
xyz = torch.from_numpy(kitti_uniform(rng, 8, 16384)).cuda()
inten = torch.from_numpy(rng.uniform(-0.5, 0.5, (8, 16384, 1)).astype(np.float32)).cuda()
model = modules.PointnetSAFPStack(in_channel=1).cuda().train()
g_out = torch.from_numpy(np.random.default_rng(6).standard_normal((8, 16384, model.out_channel)).astype(np.float32)).cuda() / 16384.0
cap = []
def torchbn(x, pool_k, layers, *params):
    n = len(layers)
    for i, l in enumerate(layers):
        w, b = params[4 * i], params[4 * i + 1]
        xin = x
        z = F.linear(x, w, b)
        mu, var = z.mean(0), z.var(0, unbiased=False)
        x = torch.relu((z - mu) / torch.sqrt(var + l.bn.eps) * l.bn.weight + l.bn.bias)
        if pool_k and i == n - 1:
            x = x.view(-1, pool_k, x.shape[-1]).max(dim=1).values
            x.retain_grad()
            cap.append((l, w.detach(), b.detach(), xin.detach(), x, pool_k))
    return x
_mlp._SharedMLPChain.apply = staticmethod(torchbn)
out = model(xyz, inten, geometry=model.geometry(xyz))
out.backward(g_out)
print("%-10s %9s | %-44s | %-44s" % ("node", "rows", "HIP: pooled dx dW dgamma dbeta (rel to max)", "torch32: pooled dx dW dgamma dbeta"))
for si, (l, w, b, xin, pooled, k) in enumerate(cap):
    dp = pooled.grad
    def ref(dt):
        xx = xin.to(dt).requires_grad_(True); ww = w.to(dt).requires_grad_(True)
        gg = l.bn.weight.detach().to(dt).requires_grad_(True); bb = l.bn.bias.detach().to(dt).requires_grad_(True)
        z = F.linear(xx, ww, b.to(dt))
        mu, var = z.mean(0), z.var(0, unbiased=False)
        y = torch.relu((z - mu) / torch.sqrt(var + l.bn.eps) * gg + bb).view(-1, k, z.shape[-1]).max(dim=1).values
        return (y.detach(),) + torch.autograd.grad(y, (xx, ww, gg, bb), dp.to(dt))
    r64, r32 = ref(torch.float64), ref(torch.float32)
    bn = _mlp.BatchNormReLU(w.shape[0], eps=l.bn.eps, momentum=0.1, relu=True).cuda()
    with torch.no_grad():
        bn.weight.copy_(l.bn.weight); bn.bias.copy_(l.bn.bias)
    xx = xin.clone().requires_grad_(True); ww = w.clone().requires_grad_(True)
    y = _mlp.linear_bn_relu_maxpool(xx, ww, b, bn, k)
    gh = (y.detach(),) + torch.autograd.grad(y, (xx, ww, bn.weight, bn.bias), dp)
    rel = lambda a, r: float((a.double() - r).abs().max() / r.abs().max())
    print("sa.%d.mlp.2 %9d | %s | %s" % (si, xin.shape[0], " ".join("%.1e" % rel(a, r) for a, r in zip(gh, r64)),
                                         " ".join("%.1e" % rel(a, r) for a, r in zip(r32, r64))))
    # ties: how many groups hold the maximum in more than one row
    with torch.no_grad():
        z = F.linear(xin, w, b); mu, var = z.mean(0), z.var(0, unbiased=False)
        yy = torch.relu((z - mu) / torch.sqrt(var + l.bn.eps) * l.bn.weight + l.bn.bias).view(-1, k, z.shape[-1])
        ties = ((yy == yy.max(dim=1, keepdim=True).values).sum(dim=1) > 1) & (yy.max(dim=1).values > 0)
        print("           (group, channel) pairs whose positive maximum is shared by several rows: %.1f%%" % (100 * float(ties.float().mean())))
