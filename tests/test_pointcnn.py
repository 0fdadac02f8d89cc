"""PointCNN backbone (heterofusionrcnn_amd/pointcnn.py) against restatements of hf/core/feature_extractors/pointcnn.py.

CPU: the (1,K)-window convolutions written as einsums against torch.nn.functional.conv2d with TensorFlow's weight layouts.
GPU: one XConv against the op-by-op form (kNN by brute force, gathers by indexing, conv2d calls), the rpn_multiclass
wiring (channel counts of every layer as the config implies them), one train step of the full RPN on it."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from heterofusionrcnn_amd import pointcnn as P_


def test_window_convolutions_equal_conv2d_with_tf_weight_layouts():
    g = torch.Generator().manual_seed(0)
    b, p, k, c, m = 2, 5, 8, 6, 3
    x = torch.randn(b, p, k, c, generator=g, dtype=torch.float64)
    w = torch.randn(k, c, m, generator=g, dtype=torch.float64)              # TF depthwise filter (1, K, C, M)
    got = P_.depthwise_k(x, w)                                              # (b, p, c*m)
    # torch: NCHW input (b, C, p, K), depthwise weight (C*M, 1, 1, K) with output channel c*M + m
    wt = w.permute(1, 2, 0).reshape(c * m, 1, 1, k)
    ref = F.conv2d(x.permute(0, 3, 1, 2), wt, groups=c).squeeze(-1).permute(0, 2, 1)
    assert torch.allclose(got, ref, rtol=1e-12, atol=1e-12)
    # X_0: tf.layers.conv2d(nn_pts_local (b,p,K,3), K*K, (1,K)) = a dense layer over the window flattened [k][c]
    local = torch.randn(b, p, k, 3, generator=g, dtype=torch.float64)
    w0 = torch.randn(1, k, 3, k * k, generator=g, dtype=torch.float64)      # HWIO
    lin = w0.reshape(k * 3, k * k).t()                                      # Linear weight (out, in) on the [k][c] flattening
    got0 = F.linear(local.reshape(b, p, k * 3), lin)
    ref0 = F.conv2d(local.permute(0, 3, 1, 2), w0.permute(3, 2, 0, 1)).squeeze(-1).permute(0, 2, 1)
    assert torch.allclose(got0, ref0, rtol=1e-12, atol=1e-12)


def test_rpn_multiclass_channel_wiring():
    cfg = P_.PointCnnConfig()
    assert [x[3] for x in cfg.xconv] == [256, 256, 512, 1024, 1024] and [x[2] for x in cfg.xconv] == [-1, 4096, 1024, 256, 64]
    with torch.device("meta"):
        bb = P_.PointCnnBackbone(cfg)
    # C_pts_fts and depth multipliers of pointcnn.py:257-266
    assert [m.lift0.linear.out_features for m in bb.enc] == [64, 64, 64, 128, 256]
    assert [m.conv.depthwise.shape[2] for m in bb.enc] == [4, 1, 2, 2, 1]
    assert [m.out_channel for m in bb.enc] == [256, 256, 512, 1024, 1280]          # with_global on the last layer
    # decoder: xconv onto the finer layer, concat with its encoder features, dense to its width (:303-356)
    assert [m.out_channel for m in bb.dec] == [1024, 1024, 512, 256, 256, 256]
    assert [f.linear.in_features for f in bb.fuse] == [1024 + 1280, 1024 + 1024, 512 + 512, 256 + 256, 256 + 256, 256 + 256]
    assert bb.out_channel == 256


@pytest.mark.gpu
def test_xconv_against_op_by_op_form():
    torch.manual_seed(2)
    b, n, p, k, cprev, c = 2, 300, 64, 8, 5, 32
    pts = torch.rand(b, n, 3, device="cuda")
    fts = torch.randn(b, n, cprev, device="cuda")
    qrs = pts[:, :p].contiguous()
    m = P_.XConv(k, 1, cprev, c, 8, 2, with_x=True, with_global=True).cuda().train()
    out = m(pts, fts, qrs)
    assert out.shape == (b, p, c + c // 4)

    def bn(x, mod):        # batch statistics, eps 1e-3
        x2 = x.reshape(-1, x.shape[-1])
        mu, var = x2.mean(0), x2.var(0, unbiased=False)
        return ((x - mu) / torch.sqrt(var + 1e-3)) * mod.bn.weight + mod.bn.bias

    def dense(x, mod):
        y = F.linear(x, mod.linear.weight)
        return bn(F.elu(y) if mod.post.activation else y, mod.post)

    d = ((qrs[:, :, None] - pts[:, None]) ** 2).sum(-1)
    idx = torch.topk(-d, k, dim=2).indices                                   # brute-force kNN, ascending distance
    bi = torch.arange(b, device="cuda")[:, None, None]
    local = pts[bi, idx] - qrs[:, :, None]
    f = torch.cat([dense(dense(local, m.lift0), m.lift1), fts[bi, idx]], -1)
    x0 = dense(local.reshape(b, p, 1, k * 3), m.x0).reshape(b, p, k, k)

    def dw(x, mod):        # depthwise (1,K) conv via conv2d, then ELU?, BN
        kk, cc, mm = mod.weight.shape
        wt = mod.weight.permute(1, 2, 0).reshape(cc * mm, 1, 1, kk)
        y = F.conv2d(x.permute(0, 3, 1, 2), wt, groups=cc).squeeze(-1).permute(0, 2, 1)
        return bn(F.elu(y) if mod.post.activation else y, mod.post)

    x1 = dw(x0, m.x1).reshape(b, p, k, k)
    x2 = dw(x1, m.x2).reshape(b, p, k, k)
    fx = torch.matmul(x2, f)
    kk, cc, mm = m.conv.depthwise.shape
    wt = m.conv.depthwise.permute(1, 2, 0).reshape(cc * mm, 1, 1, kk)
    y = F.conv2d(fx.permute(0, 3, 1, 2), wt, groups=cc).squeeze(-1).permute(0, 2, 1)
    y = bn(F.elu(F.linear(y, m.conv.pointwise.weight)), m.conv.post)
    ref = torch.cat([dense(dense(qrs, m.g0), m.g1), y], -1)
    torch.testing.assert_close(out, ref, rtol=2e-3, atol=2e-3)


@pytest.mark.gpu
def test_rpn_multiclass_train_step_small():
    """the rpn_multiclass wiring (5 xconv + 6 xdconv layers, fc, three-class heads, losses) at reduced sizes: one pass is
    finite, prefetched geometry equals inline, a few Adam steps reduce the loss"""
    import dataclasses
    from heterofusionrcnn_amd import rpn as R_
    small = P_.PointCnnConfig(xconv=((8, 1, -1, 32), (8, 1, 512, 32), (8, 1, 128, 64), (8, 1, 32, 128), (8, 1, 16, 128)),
                              fc=((32, 0.5), (32, 0.5)))
    cfg = dataclasses.replace(R_.rpn_multiclass(), pointcnn=small, rpn_fc=((64, 0.0), (64, 0.0)))
    torch.manual_seed(4)
    model = R_.RpnModel(cfg).cuda().train()
    model.backbone.fc_drop = [0.0, 0.0]
    rng = np.random.default_rng(0)
    b, p = 2, 2048
    xyz = np.stack([rng.uniform(-8, 8, (b, p)), rng.uniform(-1.0, 1.7, (b, p)), rng.uniform(2, 18, (b, p))], -1).astype(np.float32)
    boxes, cls = R_.synthetic_ground_truth(rng, b, 10, cfg, extent=((-7.0, 7.0), (3.0, 17.0)))
    xyz = torch.from_numpy(xyz).cuda()
    inten = torch.from_numpy(rng.uniform(-0.5, 0.5, (b, p, 1)).astype(np.float32)).cuda()
    label_cls, label_reg = R_.point_labels(xyz, torch.from_numpy(boxes).cuda(), torch.from_numpy(cls).cuda())
    seg1, head1 = model(xyz, inten)
    geo = model.geometry(xyz)
    seg2, head2 = model(xyz, inten, geometry=geo)
    assert seg1.shape == (b, p, 4) and head1.shape == (b, p, 3, cfg.head_width)
    assert torch.equal(seg1, seg2) and torch.equal(head1, head2)
    opt = torch.optim.Adam(model.parameters(), lr=2e-3)
    losses = []
    for _ in range(10):
        opt.zero_grad(set_to_none=True)
        seg, head = model(xyz, inten, geometry=geo)
        loss, _ = model.loss(xyz, seg, head, label_cls, label_reg)
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
    assert all(np.isfinite(losses)) and losses[-1] < 0.85 * losses[0], losses


@pytest.mark.gpu
@pytest.mark.parametrize("rows,k,c", [(1, 8, 1), (37, 8, 65), (300, 8, 320), (5, 4, 7), (2048, 8, 1280)])
def test_x_apply_kernel_against_matmul(rows, k, c):
    """hf_xconv_apply (+ grad) = torch.matmul(X, F) and its autograd, fp32 (pointcnn.py:133)"""
    from heterofusionrcnn_amd import pointcnn
    g = torch.Generator().manual_seed(rows + c)
    x = torch.randn(rows, k, k, generator=g).cuda().requires_grad_(True)
    f = torch.randn(rows, k, c, generator=g).cuda().requires_grad_(True)
    go = torch.randn(rows, k, c, generator=g).cuda()
    out = pointcnn.x_apply(x, f)
    out.backward(go)
    xr, fr = x.detach().double().requires_grad_(True), f.detach().double().requires_grad_(True)
    ref = torch.matmul(xr, fr)
    ref.backward(go.double())
    assert torch.allclose(out.double(), ref, rtol=1e-5, atol=1e-5)
    assert torch.allclose(f.grad.double(), fr.grad, rtol=1e-5, atol=1e-5)
    assert torch.allclose(x.grad.double(), xr.grad, rtol=1e-5, atol=1e-4 * max(1.0, c ** 0.5))


@pytest.mark.gpu
@pytest.mark.parametrize("rows,k,c,m", [(3, 8, 8, 8), (1000, 8, 8, 8), (77, 8, 65, 4), (500, 8, 320, 1), (129, 8, 640, 2), (9, 4, 5, 4)])
def test_depthwise_k_kernel_against_conv2d(rows, k, c, m):
    """hf_depthwise_k (+ grad) = tf.nn.depthwise_conv2d with a (1,K) window on a width-K input, channel c*M + m,
    checked against torch's grouped conv2d with the same weights and its autograd in fp64"""
    from heterofusionrcnn_amd import pointcnn
    g = torch.Generator().manual_seed(rows * 7 + c + m)
    x = torch.randn(rows, k, c, generator=g).cuda().requires_grad_(True)
    w = torch.randn(k, c, m, generator=g).cuda().requires_grad_(True)
    gy = torch.randn(rows, c * m, generator=g).cuda()
    y = pointcnn.depthwise_k(x, w)
    y.backward(gy)
    xr, wr = x.detach().double().requires_grad_(True), w.detach().double().requires_grad_(True)
    # NCHW: (rows, C, 1, K); grouped conv weight (C*M, 1, 1, K) with out channel c*M + m
    ref = torch.nn.functional.conv2d(xr.permute(0, 2, 1).unsqueeze(2), wr.permute(1, 2, 0).reshape(c * m, 1, 1, k), groups=c)
    ref = ref.reshape(rows, c * m)
    ref.backward(gy.double())
    assert torch.allclose(y.double(), ref, rtol=1e-5, atol=1e-5)
    assert torch.allclose(x.grad.double(), xr.grad, rtol=1e-5, atol=1e-5)
    assert torch.allclose(w.grad.double(), wr.grad, rtol=1e-4, atol=1e-4 * max(1.0, rows ** 0.5))


def test_oracle_xconv_products_against_conv2d():
    """the oracle's restatement of the two X-Conv products, pinned on CPU against torch.matmul / a grouped conv2d with
    TensorFlow's depthwise filter layout (the reference itself needs TensorFlow: parity unpinned beyond this)"""
    import oracle
    rng = np.random.default_rng(5)
    x = rng.standard_normal((40, 8, 8)).astype(np.float32)
    f = rng.standard_normal((40, 8, 37)).astype(np.float32)
    np.testing.assert_allclose(oracle.xconv_apply(x, f), torch.matmul(torch.from_numpy(x), torch.from_numpy(f)).numpy(), rtol=1e-5, atol=1e-5)
    xd = rng.standard_normal((30, 8, 13)).astype(np.float32)
    w = rng.standard_normal((8, 13, 4)).astype(np.float32)
    ref = F.conv2d(torch.from_numpy(xd).permute(0, 2, 1).unsqueeze(2), torch.from_numpy(w).permute(1, 2, 0).reshape(13 * 4, 1, 1, 8), groups=13)
    np.testing.assert_allclose(oracle.depthwise_k(xd, w), ref.reshape(30, 52).numpy(), rtol=1e-5, atol=1e-5)


@pytest.mark.gpu
def test_xconv_kernels_against_oracle():
    """hf_xconv_apply / hf_depthwise_k against the oracle: same multiply-then-add order, so bit for bit"""
    import oracle
    from heterofusionrcnn_amd import pointcnn
    rng = np.random.default_rng(6)
    x = rng.standard_normal((500, 8, 8)).astype(np.float32)
    f = rng.standard_normal((500, 8, 320)).astype(np.float32)
    got = pointcnn.x_apply(torch.from_numpy(x).cuda(), torch.from_numpy(f).cuda()).cpu().numpy()
    assert np.array_equal(got, oracle.xconv_apply(x, f))
    xd = rng.standard_normal((700, 8, 65)).astype(np.float32)
    w = rng.standard_normal((8, 65, 4)).astype(np.float32)
    got = pointcnn.depthwise_k(torch.from_numpy(xd).cuda(), torch.from_numpy(w).cuda()).cpu().numpy()
    assert np.array_equal(got, oracle.depthwise_k(xd, w))


@pytest.mark.gpu
@pytest.mark.parametrize("rows,c,m", [(1, 1, 1), (37, 65, 4), (300, 320, 1), (129, 640, 2), (2048, 1280, 1), (50, 70, 3)])
def test_fused_xconv_depthwise_equals_two_kernels(rows, c, m):
    """hf_xconv_depthwise (+ grad) against hf_xconv_apply followed by hf_depthwise_k: forward bit for bit (same multiply /
    add order), gradients to rounding (the weight gradient is accumulated with atomics in both routes)"""
    from heterofusionrcnn_amd import pointcnn
    g = torch.Generator().manual_seed(rows + c + m)
    k = 8
    mk = lambda *s: torch.randn(*s, generator=g).cuda().requires_grad_(True)
    x, f, w = mk(rows, k, k), mk(rows, k, c), mk(k, c, m)
    go = torch.randn(rows, c * m, generator=g).cuda()
    out = pointcnn.xconv_depthwise(x, f, w)
    out.backward(go)
    x2, f2, w2 = (t.detach().clone().requires_grad_(True) for t in (x, f, w))
    ref = pointcnn.depthwise_k(pointcnn.x_apply(x2, f2), w2)
    ref.backward(go)
    assert torch.equal(out, ref)
    assert torch.allclose(f.grad, f2.grad, rtol=1e-6, atol=1e-6)
    assert torch.allclose(x.grad, x2.grad, rtol=1e-5, atol=1e-5 * max(1.0, c ** 0.5))
    assert torch.allclose(w.grad, w2.grad, rtol=1e-4, atol=1e-4 * max(1.0, rows ** 0.5))
