"""where does the column-sum error of the backward pass enter?  gradients w.r.t. every MLP layer OUTPUT of the config-2 stack
(layer-by-layer HIP nodes vs torch fp32, both against torch fp64): max error and error of the column sums (= what dbeta of the
layer below integrates), relative to the fp64 values"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import heterofusionrcnn_amd as hf
from heterofusionrcnn_amd import modules, mlp as _mlp
from heterofusionrcnn_amd.modules import three_nn_weights
from bench import kitti_uniform
F = torch.nn.functional
torch.manual_seed(5)
rng = np.random.default_rng(5)
xyz = torch.from_numpy(kitti_uniform(rng, 8, 16384)).cuda()
inten = torch.from_numpy(rng.uniform(-0.5, 0.5, (8, 16384, 1)).astype(np.float32)).cuda()
model = modules.PointnetSAFPStack(in_channel=1).cuda().train()
tgt = torch.linspace(-1.0, 1.0, model.out_channel, device="cuda")
lossf = lambda o: ((o - tgt.to(o.dtype)) ** 2).mean()
store = {}
def keep(tag, name, t):
    t.register_hook(lambda g, tag=tag, name=name: store.setdefault(tag, {}).__setitem__(name, g.detach().double()))
    return t
counter = [0]
def make(tag, mode, dt):
    def fn(x, pool_k, layers, *params):
        n = len(layers)
        node = counter[0]; counter[0] += 1
        for i, l in enumerate(layers):
            w, b = params[4 * i].to(dt), params[4 * i + 1].to(dt)
            if mode == "hip":
                args = (x, w, b, l.bn.weight, l.bn.bias, l.bn.running_mean, l.bn.running_var, l.bn.eps, l.bn.momentum)
                x = _mlp._LinearBNReLUMaxPool.apply(*args, pool_k) if (pool_k and i == n - 1) else _mlp._LinearBNReLU.apply(*args, True)
            else:
                z = F.linear(x, w, b)
                mu, var = z.mean(0), z.var(0, unbiased=False)
                x = torch.relu((z - mu) / torch.sqrt(var + l.bn.eps) * l.bn.weight.to(dt) + l.bn.bias.to(dt))
                if pool_k and i == n - 1:
                    x = x.view(-1, pool_k, x.shape[-1]).max(dim=1).values
            x = keep(tag, "node%d.layer%d" % (node, i), x)
        return x
    return fn
geo = model.geometry(xyz)
for tag, mode, dt in (("hip", "hip", torch.float32), ("t32", "torch", torch.float32)):
    counter[0] = 0
    _mlp._SharedMLPChain.apply = staticmethod(make(tag, mode, dt))
    model.zero_grad()
    lossf(model(xyz, inten, geometry=geo)).backward()
# fp64: the same graph with the features in double (the HIP gather / interpolate ops are fp32-only: torch gathers here)
def ref64():
    counter[0] = 0
    fn = make("t64", "torch", torch.float64)
    xyzs, feats = [xyz], [inten.double()]
    for m in model.sa:
        new_xyz = hf.gather_point(xyzs[-1], hf.farthest_point_sample(m.npoint, xyzs[-1]))
        idx, _ = hf.query_ball_point(m.radius, m.nsample, xyzs[-1], new_xyz)
        li = idx.long(); b_, n_, k_ = idx.shape
        gat = lambda t: torch.gather(t, 1, li.reshape(b_, -1, 1).expand(-1, -1, t.shape[-1])).reshape(b_, n_, k_, -1)
        g = torch.cat([(gat(xyzs[-1]) - new_xyz.unsqueeze(2)).double(), gat(feats[-1])], -1)
        params = []
        for l in m.mlp:
            params += [l.fc.weight, l.fc.bias, l.bn.weight, l.bn.bias]
        x = fn(g.reshape(-1, g.shape[-1]), k_, list(m.mlp), *params)
        xyzs.append(new_xyz); feats.append(x.reshape(b_, n_, -1))
    up = feats[-1]
    for i, m in enumerate(model.fp):
        d = len(model.sa) - 1 - i
        dist, idx = hf.three_nn(xyzs[d], xyzs[d + 1])
        w = three_nn_weights(dist).double(); li = idx.long(); b_, n_, _ = idx.shape
        gath = torch.gather(up, 1, li.reshape(b_, -1, 1).expand(-1, -1, up.shape[-1])).reshape(b_, n_, 3, -1)
        x = torch.cat([(gath * w.unsqueeze(-1)).sum(2), feats[d]], 2)
        params = []
        for l in m.mlp:
            params += [l.fc.weight, l.fc.bias, l.bn.weight, l.bn.bias]
        up = fn(x.reshape(-1, x.shape[-1]), 0, list(m.mlp), *params).reshape(b_, n_, -1)
    return up
lossf(ref64()).backward()
print("%-16s %10s | %-26s | %-26s" % ("grad w.r.t.", "rows", "HIP: max err, colsum err", "torch32: max err, colsum err"))
for name in sorted(store["t64"], key=lambda s: (int(s.split(".")[0][4:]), int(s.split("layer")[1]))):
    r = store["t64"][name]
    row = []
    for tag in ("hip", "t32"):
        g = store[tag][name]
        row.append("%.1e  %.1e" % (float((g - r).abs().max() / r.abs().max()), float((g.sum(0) - r.sum(0)).abs().max() / r.sum(0).abs().max())))
    print("%-16s %10d | %-26s | %-26s" % (name, r.shape[0], row[0], row[1]))
