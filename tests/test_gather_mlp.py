"""the first layer of a set-abstraction MLP on neighbourhoods read in place (mlp.shared_mlp_grouped:
hf_linear_bn_fwd_gather / hf_linear_wgrad_gather) against the materialised route (group_concat -> shared_mlp) and against an
fp64 evaluation of pointnet_util.py:42-64 + 156-176 (group, centre, concat, conv2d [1,1] + batch norm + ReLU, max over K)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _setup(b, n, m, k, c, widths, seed, xyz_first):
    import heterofusionrcnn_amd as hf
    from heterofusionrcnn_amd.modules import SharedMLPLayer
    rng = np.random.default_rng(seed)
    xyz = torch.from_numpy(rng.random((b, n, 3), dtype=np.float32)).cuda()
    new_xyz = hf.gather_point(xyz, hf.farthest_point_sample(m, xyz))
    idx, _, gxyz = hf.query_ball_group(0.25, k, xyz, new_xyz, center=True)
    pts = torch.from_numpy(rng.standard_normal((b, n, c)).astype(np.float32)).cuda().requires_grad_(True) if c else None
    torch.manual_seed(seed)
    layers, cin = [], c + 3
    for w in widths:
        layers.append(SharedMLPLayer(cin, w).cuda().train())
        cin = w
    for l in layers:                                              # non-trivial BatchNorm parameters
        with torch.no_grad():
            l.bn.weight.uniform_(0.5, 1.5)
            l.bn.bias.uniform_(-0.3, 0.3)
            l.fc.bias.uniform_(-0.1, 0.1)
    return xyz, idx, gxyz, pts, layers


def _fp64_reference(idx, gxyz, pts, layers, xyz_first, pool, dout):
    """the reference graph in fp64 torch ops with autograd"""
    b, m, k = idx.shape
    p64 = pts.detach().double().requires_grad_(True) if pts is not None else None
    parts = [gxyz.double()]
    if p64 is not None:
        gathered = torch.gather(p64.unsqueeze(1).expand(b, m, -1, -1), 2, idx.long().unsqueeze(-1).expand(b, m, k, p64.shape[-1]))
        parts = [gxyz.double(), gathered] if xyz_first else [gathered, gxyz.double()]
    x = torch.cat(parts, -1).reshape(b * m * k, -1)
    ws = []
    for l in layers:
        w, bb = l.fc.weight.detach().double().requires_grad_(True), l.fc.bias.detach().double().requires_grad_(True)
        g, be = l.bn.weight.detach().double().requires_grad_(True), l.bn.bias.detach().double().requires_grad_(True)
        ws += [w, bb, g, be]
        z = x @ w.t() + bb
        mu, var = z.mean(0), z.var(0, unbiased=False)
        x = torch.relu(g * (z - mu) / torch.sqrt(var + l.bn.eps) + be)
    out = x.reshape(b * m, k, -1).max(1).values if pool else x
    out.backward(dout.double())
    return out.detach(), (p64.grad if p64 is not None else None), [w.grad for w in ws]


@pytest.mark.parametrize("c,widths,k,xyz_first", [(0, (32, 64), 32, True), (1, (32, 32, 64), 32, True), (64, (64, 128), 16, True),
                                                   (67, (96,), 32, False), (128, (128, 256), 8, False), (5, (16, 224), 128, True)])
def test_grouped_mlp_in_place_matches_materialised_route_and_fp64(c, widths, k, xyz_first):
    from heterofusionrcnn_amd.grouping import group_concat
    from heterofusionrcnn_amd.mlp import shared_mlp, shared_mlp_grouped, grouped_mlp_fusable
    b, n, m = 2, 2048, 300
    if k == 128:
        m = 37
    xyz, idx, gxyz, pts, layers = _setup(b, n, m, k, c, widths, 100 + c, xyz_first)
    assert grouped_mlp_fusable(layers, pts, idx)
    torch.manual_seed(0)
    dout = torch.randn(b * m, widths[-1], device="cuda")
    params = [p for l in layers for p in (l.fc.weight, l.fc.bias, l.bn.weight, l.bn.bias)]

    def run(fn):
        for p in params + ([pts] if pts is not None else []):
            p.grad = None
        for l in layers:
            l.bn.running_mean.zero_(); l.bn.running_var.fill_(1.0)
        out = fn()
        out.backward(dout)
        return (out.detach().clone(), pts.grad.clone() if pts is not None else None, [p.grad.clone() for p in params],
                [(l.bn.running_mean.clone(), l.bn.running_var.clone()) for l in layers])

    got = run(lambda: shared_mlp_grouped(layers, pts, idx, gxyz, xyz_first=xyz_first))

    def materialised():
        if pts is not None:
            g = group_concat(pts, idx, gxyz, xyz_last=not xyz_first)
        else:
            g = torch.nn.functional.pad(gxyz, (0, 1))
        return shared_mlp(layers, g.reshape(-1, g.shape[-1]), pool_k=k)
    want = run(materialised)

    ref_out, ref_dp, ref_dw = _fp64_reference(idx, gxyz, pts, layers, xyz_first, True, dout)
    # against fp64: forward 1e-4 of the output scale (fp32 GEMM + batch statistics), gradients 2e-3 of their scale
    scale = float(ref_out.abs().max())
    assert float((got[0].double() - ref_out).abs().max()) <= 1e-4 * max(scale, 1.0)
    assert float((got[0] - want[0]).abs().max()) <= 2e-5 * max(scale, 1.0)            # the two fp32 routes against each other
    if pts is not None:
        gs = float(ref_dp.abs().max())
        assert float((got[1].double() - ref_dp).abs().max()) <= 2e-3 * gs
    for gi, (a, r) in enumerate(zip(got[2], ref_dw)):
        if gi % 4 == 1:
            assert float(a.abs().max()) == 0.0                                        # a bias under a BatchNorm: exactly zero
            continue
        rs = float(r.abs().max())
        assert float((a.double() - r).abs().max()) <= 3e-3 * rs + 1e-6, (gi, float((a.double() - r).abs().max()), rs)
    for (m1, v1), (m2, v2) in zip(got[3], want[3]):                                   # running statistics updated the same way
        assert torch.allclose(m1, m2, rtol=1e-4, atol=1e-6) and torch.allclose(v1, v2, rtol=1e-4, atol=1e-6)


def test_grouped_mlp_eval_mode_and_module_switch():
    """inference (running statistics) through the in-place route == the materialised route; modules.GATHER_ON_LOAD switches a
    PointnetSAModule between them"""
    from heterofusionrcnn_amd import modules
    from heterofusionrcnn_amd.grouping import group_concat
    from heterofusionrcnn_amd.mlp import shared_mlp, shared_mlp_grouped
    xyz, idx, gxyz, pts, layers = _setup(2, 1024, 128, 32, 16, (32, 64), 7, True)
    with torch.no_grad():                                         # give the running statistics some content
        for _ in range(3):
            pass
    for l in layers:
        l.bn.running_mean.normal_(0, 0.1); l.bn.running_var.uniform_(0.5, 1.5)
        l.eval()
    with torch.no_grad():
        a = shared_mlp_grouped(layers, pts.detach(), idx, gxyz, xyz_first=True)
        g = group_concat(pts.detach(), idx, gxyz)
        bb = shared_mlp(layers, g.reshape(-1, g.shape[-1]), pool_k=32)
    assert torch.allclose(a, bb, rtol=1e-5, atol=1e-5)
    torch.manual_seed(3)
    sa = modules.PointnetSAModule(128, 0.25, 32, 16, [32, 64]).cuda().train()
    outs = []
    for flag in (True, False):
        modules.GATHER_ON_LOAD = flag
        try:
            for l in sa.mlp:
                l.bn.running_mean.zero_(); l.bn.running_var.fill_(1.0)
            p = pts.detach().clone().requires_grad_(True)
            _, f, _ = sa(xyz, p)
            f.square().sum().backward()
            outs.append((f.detach(), p.grad.clone(), [q.grad.clone() for q in sa.parameters()]))
            sa.zero_grad()
        finally:
            modules.GATHER_ON_LOAD = True
    assert torch.allclose(outs[0][0], outs[1][0], rtol=1e-4, atol=1e-5)
    assert torch.allclose(outs[0][1], outs[1][1], rtol=1e-3, atol=1e-4 * float(outs[1][1].abs().max()))
    for a_, b_ in zip(outs[0][2], outs[1][2]):
        assert torch.allclose(a_, b_, rtol=2e-3, atol=2e-3 * float(b_.abs().max()) + 1e-7)


def test_gather_entry_points_reject_bad_arguments():
    from heterofusionrcnn_amd import _lib
    L = _lib.lib()
    z = torch.zeros(8, device="cuda")
    p = lambda t: t.data_ptr()
    # rows not a multiple of rows_per_cloud, missing idx, c_feat without points
    assert L.hf_linear_bn_fwd_gather(100, 0, 32, None, 1, 64, p(z), p(z), p(z), None, p(z), 1e-3, 0.1, None, None, p(z), p(z), p(z), 1 << 20, None) == _lib.HF_EINVAL
    assert L.hf_linear_bn_fwd_gather(128, 0, 32, None, 1, 64, None, p(z), p(z), None, p(z), 1e-3, 0.1, None, None, p(z), p(z), p(z), 1 << 20, None) == _lib.HF_EINVAL
    assert L.hf_linear_wgrad_gather(128, 32, 4, p(z), None, 16, 64, p(z), p(z), p(z), p(z), 1 << 20, None) == _lib.HF_EINVAL
