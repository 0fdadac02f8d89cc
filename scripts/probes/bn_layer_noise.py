"""single-layer BatchNorm+ReLU noise: (z, dy) of every MLP layer captured from the op-by-op fp32 graph of the config-2 stack;
forward y and backward (dx, dgamma, dbeta) of that one layer by (a) the HIP kernels, (b) torch fp32, against (c) torch fp64"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import heterofusionrcnn_amd as hf
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
    for i, l in enumerate(layers):
        w, b = params[4 * i], params[4 * i + 1]
        z = F.linear(x, w, b)
        mu, var = z.mean(0), z.var(0, unbiased=False)
        x = torch.relu((z - mu) / torch.sqrt(var + l.bn.eps) * l.bn.weight + l.bn.bias)
        x.retain_grad()
        cap.append((l, z.detach(), x))
    return x.view(-1, pool_k, x.shape[-1]).max(dim=1).values if pool_k else x
_mlp._SharedMLPChain.apply = staticmethod(torchbn)
out = model(xyz, inten, geometry=model.geometry(xyz))
out.backward(g_out)
names = [n[:-len(".bn.weight")] for n, _ in model.named_parameters() if n.endswith("bn.weight")]
print("%-14s %9s %6s %8s | %-32s | %-32s" % ("layer", "rows", "C", "max|mu/sd|", "HIP: y dx dgamma dbeta (rel to max)", "torch32: y dx dgamma dbeta"))
for name, (l, z, y) in zip(names, cap):
    dy = y.grad
    g, b = l.bn.weight.detach(), l.bn.bias.detach()
    def ref(dt):
        zz = z.to(dt).requires_grad_(True); gg = g.to(dt).requires_grad_(True); bb = b.to(dt).requires_grad_(True)
        mu, var = zz.mean(0), zz.var(0, unbiased=False)
        yy = torch.relu((zz - mu) / torch.sqrt(var + l.bn.eps) * gg + bb)
        dx, dg, db = torch.autograd.grad(yy, (zz, gg, bb), dy.to(dt))
        return yy.detach(), dx, dg, db, mu.detach(), var.detach()
    y64, dx64, dg64, db64, mu64, var64 = ref(torch.float64)
    y32, dx32, dg32, db32, _, _ = ref(torch.float32)
    bn = _mlp.BatchNormReLU(z.shape[1], eps=l.bn.eps, momentum=0.1, relu=True).cuda()
    with torch.no_grad():
        bn.weight.copy_(g); bn.bias.copy_(b)
    zz = z.clone().requires_grad_(True)
    yh = bn(zz)
    dxh, dgh, dbh = torch.autograd.grad(yh, (zz, bn.weight, bn.bias), dy)
    rel = lambda a, r: float((a.double() - r).abs().max() / r.abs().max())
    print("%-14s %9d %6d %8.1f | %.1e %.1e %.1e %.1e | %.1e %.1e %.1e %.1e" % (
        name, z.shape[0], z.shape[1], float((mu64.abs() / torch.sqrt(var64 + 1e-30)).max()),
        rel(yh, y64), rel(dxh, dx64), rel(dgh, dg64), rel(dbh, db64), rel(y32, y64), rel(dx32, dx64), rel(dg32, dg64), rel(db32, db64)))
