"""RPN train step (heterofusionrcnn_amd/rpn.py) against restatements of hf/core/models/rpn_model.py.

CPU (-m "not gpu"): the loss against a LITERAL op-by-op restatement of the reference graph (one-hot targets, per-class
gather_nd, boolean_mask, hf/core/losses.py formulas) -- values and gradients; the per-point labels against a loop.
GPU (-m gpu): targets against the oracle's bin encoder, the whole step (finite, trains, prefetched geometry == inline,
fused nodes == the op-by-op torch form of the same network with shared weights).
TensorFlow is not importable here: parity unpinned against reference OUTPUTS; what is pinned is the text of the graph.
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from heterofusionrcnn_amd import rpn as R_


def _cfg(k=3):
    base = R_.rpn_stack_config2()
    return R_.rpn_multiclass_heads(base) if k == 3 else base


def _literal_reference_loss(cfg, seg_logits, head, label_cls, t):
    """rpn_model.py:708-796 + 1040-1128 + losses.py:131-226 written the way the reference writes them"""
    b, p, k1 = seg_logits.shape
    k = cfg.num_classes
    nbx, nbt = cfg.num_bin_xz, cfg.theta_bin_num
    seg_softmax = torch.softmax(seg_logits, -1)
    seg_gt = F.one_hot(label_cls.long(), k + 1).float()                               # :713-719
    pred = seg_softmax.clamp(1e-7, 1 - 1e-7)                                          # losses.py:213-225
    seg = (0.25 * seg_gt * (1 - pred) ** 2 * (-seg_gt * torch.log(pred))).sum() * cfg.seg_loss_weight
    seg = seg / (b * p)
    parts = R_.parse_rpn_output(head, nbx, nbx, nbt)                                   # (B,P,K,*)
    bI, pI = torch.meshgrid(torch.arange(b), torch.arange(p), indexing="ij")
    c = t["cls0"]
    gather_cls = lambda x: x[bI, pI, c]                                               # _gather_cls_preds
    bx, bz, bt = gather_cls(parts[0]), gather_cls(parts[2]), gather_cls(parts[4])
    ry, rs = gather_cls(parts[6]), gather_cls(parts[7])
    rx = parts[1][bI, pI, c, t["bin_x"]]                                              # _gather_cls_residuals
    rz = parts[3][bI, pI, c, t["bin_z"]]
    rt = parts[5][bI, pI, c, t["bin_theta"]]
    fg = label_cls > 0                                                                # :504-505
    nfg = fg.float().sum()
    cls_loss = 0.0
    for logits, gt, depth in ((bx, t["bin_x"], nbx), (bz, t["bin_z"], nbx), (bt, t["bin_theta"], nbt)):
        one_hot = F.one_hot(gt, depth).float()
        lg, oh = logits[fg], one_hot[fg]                                              # tf.boolean_mask
        cls_loss = cls_loss + (-(oh * torch.log_softmax(lg, -1)).sum(-1)).sum() * cfg.cls_loss_weight
    reg_loss = 0.0
    for pr, gt in ((rx, t["res_x"]), (rz, t["res_z"]), (rt, t["res_theta"]), (ry, t["res_y"]), (rs, t["res_size"])):
        d = (pr[fg] - gt[fg]).abs()
        reg_loss = reg_loss + torch.where(d < 1, 0.5 * d * d, d - 0.5).sum() * cfg.reg_loss_weight
    if nfg > 0:
        cls_loss, reg_loss = cls_loss / nfg, reg_loss / nfg
    else:
        cls_loss, reg_loss = cls_loss * 0.0, reg_loss * 0.0
    return seg + cls_loss + reg_loss


def _random_case(cfg, b, p, seed, fg_frac=0.2):
    g = torch.Generator().manual_seed(seed)
    k = cfg.num_classes
    label_cls = (torch.rand(b, p, generator=g) < fg_frac).long() * torch.randint(1, k + 1, (b, p), generator=g)
    t = {"cls0": torch.clamp(label_cls - 1, min=0),
         "bin_x": torch.randint(0, cfg.num_bin_xz, (b, p), generator=g), "res_x": torch.randn(b, p, generator=g),
         "bin_z": torch.randint(0, cfg.num_bin_xz, (b, p), generator=g), "res_z": torch.randn(b, p, generator=g),
         "bin_theta": torch.randint(0, cfg.theta_bin_num, (b, p), generator=g), "res_theta": torch.randn(b, p, generator=g),
         "res_y": torch.randn(b, p, generator=g) * 2, "res_size": torch.randn(b, p, 3, generator=g)}
    seg_logits = torch.randn(b, p, k + 1, generator=g, dtype=torch.float64).requires_grad_(True)
    head = (2 * torch.randn(b, p, k, cfg.head_width, generator=g, dtype=torch.float64)).requires_grad_(True)
    t64 = {n: (v.double() if v.is_floating_point() else v) for n, v in t.items()}
    return seg_logits, head, label_cls, t64


@pytest.mark.parametrize("k,fg_frac", [(1, 0.2), (3, 0.3), (3, 0.0)])
def test_rpn_loss_equals_literal_reference_graph(k, fg_frac):
    cfg = _cfg(k)
    seg_logits, head, label_cls, t = _random_case(cfg, 2, 257, seed=k * 10 + int(fg_frac * 10), fg_frac=fg_frac)
    loss, parts = R_.rpn_loss(cfg, seg_logits, head, label_cls, t)
    ref = _literal_reference_loss(cfg, seg_logits, head, label_cls, t)
    assert torch.allclose(loss, ref, rtol=1e-12, atol=1e-12)
    g1 = torch.autograd.grad(loss, (seg_logits, head), allow_unused=True)
    g2 = torch.autograd.grad(ref, (seg_logits, head), allow_unused=True)
    for a, b in zip(g1, g2):
        a = torch.zeros(1, dtype=torch.float64) if a is None else a
        b = torch.zeros(1, dtype=torch.float64) if b is None else b
        assert torch.allclose(a, b, rtol=1e-10, atol=1e-12)
    if fg_frac == 0.0:
        assert parts["bin_classification"] == 0 and parts["regression"] == 0 and parts["num_foreground"] == 0


def test_point_labels_against_a_loop():
    rng = np.random.default_rng(5)
    cfg = _cfg(3)
    boxes, cls = R_.synthetic_ground_truth(rng, 2, 6, cfg, extent=((-6.0, 6.0), (2.0, 14.0)))
    cls[1, 4:] = 0                                                    # padding entries are never matched
    boxes[0, 1, :3] = boxes[0, 0, :3]                                  # overlapping boxes: the first one wins
    xyz = np.stack([rng.uniform(-7, 7, (2, 4000)), rng.uniform(-1.0, 2.0, (2, 4000)), rng.uniform(1, 15, (2, 4000))], -1).astype(np.float32)
    lc, lr = R_.point_labels(torch.from_numpy(xyz), torch.from_numpy(boxes), torch.from_numpy(cls))
    exp_c = np.zeros((2, 4000), np.int64)
    exp_r = np.zeros((2, 4000, 7), np.float32)
    for b in range(2):
        for j in range(6):
            if cls[b, j] == 0:
                continue
            x, y, z, l, w, h, ry = boxes[b, j]
            dx, dy, dz = xyz[b, :, 0] - x, xyz[b, :, 1] - y, xyz[b, :, 2] - z
            lx = np.cos(ry) * dx - np.sin(ry) * dz
            lz = np.sin(ry) * dx + np.cos(ry) * dz
            inside = (np.abs(lx) <= l / 2) & (np.abs(lz) <= w / 2) & (dy <= 0) & (dy >= -h) & (exp_c[b] == 0)
            exp_c[b, inside] = cls[b, j]
            exp_r[b, inside] = boxes[b, j]
    assert exp_c.sum() > 0
    assert np.array_equal(lc.numpy(), exp_c) and np.array_equal(lr.numpy(), exp_r)


def test_configs_match_the_reference_files():
    c = R_.rpn_cars_pointnet_paper()
    assert [l.npoint for l in c.sa] == [4096, 1024, 512, 64] and c.sa[0].scales[1].radius == 0.5
    assert c.num_bin_xz == 12 and c.head_width == 76 and abs(c.delta_theta - 2 * np.pi / 12) < 1e-12
    m = R_.rpn_multiclass_heads(c)
    assert m.num_classes == 3 and m.num_bin_xz == 12 and len(m.cluster_sizes) == 3


# ------------------------------------------------------------------------------------------------ GPU
def _small_cfg():
    s = R_.SAScale
    return R_.rpn_multiclass_heads(R_.RpnConfig(
        name="small", sa=(R_.SALevel(256, (s(0.8, 16, (16, 16, 32)), s(1.6, 32, (16, 16, 32)))),
                          R_.SALevel(64, (s(3.2, 16, (32, 32, 64)),))),
        fp=((64, 64), (32, 32)), backbone_fc=((32, 0.5), (32, 0.5)), rpn_fc=((64, 0.5), (64, 0.5))))


def _small_batch(cfg, device, b=2, p=2048, seed=0):
    rng = np.random.default_rng(seed)
    xyz = np.stack([rng.uniform(-8, 8, (b, p)), rng.uniform(-1.0, 1.7, (b, p)), rng.uniform(2, 18, (b, p))], -1).astype(np.float32)
    boxes, cls = R_.synthetic_ground_truth(rng, b, 10, cfg, extent=((-7.0, 7.0), (3.0, 17.0)))
    xyz_t = torch.from_numpy(xyz).to(device)
    inten = torch.from_numpy(rng.uniform(-0.5, 0.5, (b, p, 1)).astype(np.float32)).to(device)
    label_cls, label_reg = R_.point_labels(xyz_t, torch.from_numpy(boxes).to(device), torch.from_numpy(cls).to(device))
    return xyz_t, inten, label_cls, label_reg


@pytest.mark.gpu
def test_rpn_targets_against_oracle_encoder():
    import oracle
    cfg = _small_cfg()
    xyz, _, label_cls, label_reg = _small_batch(cfg, "cuda")
    assert int((label_cls > 0).sum()) > 50
    t = R_.rpn_targets(cfg, xyz, label_cls, label_reg)
    c0 = np.clip(label_cls.cpu().numpy() - 1, 0, None).reshape(-1)
    ms = np.asarray(cfg.cluster_sizes, np.float32)[c0]
    o = oracle.bin_box_encode(xyz.cpu().numpy().reshape(-1, 3), None, label_reg.cpu().numpy().reshape(-1, 7), ms,
                              list(cfg.xz_search_range), list(cfg.xz_bin_len), cfg.r_theta, cfg.delta_theta, cfg.num_classes, False)
    pick = lambda a: np.take_along_axis(a, c0[:, None], 1)[:, 0]
    fg = (label_cls > 0).cpu().numpy().reshape(-1)
    for name, exp in (("bin_x", pick(o[0])), ("res_x", pick(o[1])), ("bin_z", pick(o[2])), ("res_z", pick(o[3])),
                      ("bin_theta", o[4]), ("res_theta", o[5]), ("res_y", o[6]), ("res_size", o[7])):
        got = t[name].cpu().numpy()
        got = got.reshape(-1, 3) if name == "res_size" else got.reshape(-1)
        if got.dtype.kind == "i":
            assert np.array_equal(got[fg], exp[fg]), name
        else:
            np.testing.assert_allclose(got[fg], exp[fg], rtol=0, atol=1e-5, err_msg=name)


@pytest.mark.gpu
def test_rpn_train_step_trains_and_prefetched_geometry_equals_inline():
    cfg = _small_cfg()
    torch.manual_seed(3)
    model = R_.RpnModel(cfg).cuda().train()
    xyz, inten, label_cls, label_reg = _small_batch(cfg, "cuda")
    for m in model.modules():                                              # dropout off: compare two evaluations
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    model.backbone.fc_keep = [1.0] * len(model.backbone.fc_keep)
    model.heads.drop = [0.0] * len(model.heads.drop)
    seg1, head1 = model(xyz, inten)
    geo = model.geometry(xyz)
    seg2, head2 = model(xyz, inten, geometry=geo)
    assert torch.equal(seg1, seg2) and torch.equal(head1, head2)
    opt = torch.optim.Adam(model.parameters(), lr=2e-3)
    losses = []
    for _ in range(12):
        opt.zero_grad(set_to_none=True)
        seg, head = model(xyz, inten, geometry=geo)
        loss, parts = model.loss(xyz, seg, head, label_cls, label_reg)
        loss.backward()
        for n, p_ in model.named_parameters():
            assert p_.grad is not None and torch.isfinite(p_.grad).all(), n
        opt.step()
        losses.append(float(loss.detach()))
    assert all(np.isfinite(losses)) and losses[-1] < 0.8 * losses[0], losses
    assert float(parts["num_foreground"]) > 50


@pytest.mark.gpu
def test_rpn_backbone_fused_nodes_equal_op_by_op_torch_form():
    """the same network, same weights: HIP fused nodes (group_concat + MFMA linear/BN/ReLU(+max-pool) chains,
    interpolate+concat) vs plain torch ops on the tensors the reference materialises (group -> concat -> 1x1 conv -> BN
    -> ReLU -> max; three_interpolate -> concat -> conv ...), forward values and parameter gradients"""
    import heterofusionrcnn_amd as hf
    from heterofusionrcnn_amd.modules import three_nn_weights
    cfg = _small_cfg()
    torch.manual_seed(11)
    model = R_.RpnModel(cfg).cuda().train()
    model.backbone.fc_keep = [1.0] * len(model.backbone.fc_keep)
    model.heads.drop = [0.0] * len(model.heads.drop)
    xyz, inten, label_cls, label_reg = _small_batch(cfg, "cuda", seed=4)

    def layer_ref(layer, x):                                            # Linear + bias -> BN(batch stats, eps 1e-3) -> ReLU
        z = F.linear(x, layer.fc.weight, layer.fc.bias)
        mu, var = z.mean(0), z.var(0, unbiased=False)
        return torch.relu((z - mu) / torch.sqrt(var + 1e-3) * layer.bn.weight + layer.bn.bias)

    def backbone_ref(bb, xyz, pts):
        xyzs, feats = [xyz], [pts]
        for m in bb.sa:
            new_xyz = hf.gather_point(xyzs[-1], hf.farthest_point_sample(m.npoint, xyzs[-1]))
            outs = []
            for sc, mlp in zip(m.scales, m.mlps):
                idx, _ = hf.query_ball_point(sc.radius, sc.nsample, xyzs[-1], new_xyz)
                gx = hf.group_point(xyzs[-1], idx) - new_xyz.unsqueeze(2)
                gp = hf.group_point(feats[-1], idx)
                g = torch.cat([gp, gx], -1) if m.msg else torch.cat([gx, gp], -1)
                b_, n_, k_, c_ = g.shape
                x = g.reshape(-1, c_)
                for layer in mlp:
                    x = layer_ref(layer, x)
                outs.append(x.reshape(b_, n_, k_, -1).max(2).values)
            xyzs.append(new_xyz)
            feats.append(outs[0] if len(outs) == 1 else torch.cat(outs, -1))
        up = feats[-1]
        for i, m in enumerate(bb.fp):
            d = len(bb.sa) - 1 - i
            dist, idx = hf.three_nn(xyzs[d], xyzs[d + 1])
            x = torch.cat([hf.three_interpolate(up, idx, three_nn_weights(dist)), feats[d]], 2)
            b_, n_, c_ = x.shape
            x = x.reshape(-1, c_)
            for layer in m.mlp:
                x = layer_ref(layer, x)
            up = x.reshape(b_, n_, -1)
        b_, n_, c_ = up.shape
        x = up.reshape(-1, c_)
        for layer in bb.fc:
            x = layer_ref(layer, x)
        return x.reshape(b_, n_, -1)

    seg, head = model(xyz, inten)
    loss, _ = model.loss(xyz, seg, head, label_cls, label_reg)
    params = [p_ for p_ in model.parameters()]
    g_fused = torch.autograd.grad(loss, params, allow_unused=True)
    # running statistics were updated by the first pass; the reference form below uses batch statistics only
    seg_r, head_r = model.heads(backbone_ref(model.backbone, xyz, inten))
    loss_r, _ = model.loss(xyz, seg_r, head_r, label_cls, label_reg)
    g_ref = torch.autograd.grad(loss_r, params, allow_unused=True)
    assert torch.allclose(seg, seg_r, rtol=1e-3, atol=1e-3) and torch.allclose(head, head_r, rtol=2e-3, atol=2e-3)
    assert abs(float(loss) - float(loss_r)) <= 1e-3 * abs(float(loss_r))
    for (n, _), a, b in zip(model.named_parameters(), g_fused, g_ref):
        if a is None or b is None:
            # a bias in front of a BatchNorm has an exactly-zero gradient; the fused node returns None / zeros for it
            assert b is None or float(b.abs().max()) < 1e-4, n
            continue
        scale = float(b.abs().max()) + 1e-6
        assert float((a - b).abs().max()) <= 2e-2 * scale + 1e-5, (n, float((a - b).abs().max()), scale)


@pytest.mark.gpu
@pytest.mark.parametrize("k", [1, 3])
def test_fused_rpn_loss_against_the_op_by_op_form(k):
    """hf_rpn_loss_fwd / _bwd against rpn_targets + rpn_loss (which the CPU tests pin to the literal reference graph): the
    loss, its three parts, the foreground count, and the gradients w.r.t. the segmentation logits and the head -- also with an
    upstream factor, with clipped softmax probabilities (|logit| = 40), and with no foreground point at all"""
    cfg = _small_cfg() if k == 3 else R_.rpn_stack_config2()
    xyz, _, label_cls, label_reg = _small_batch(cfg, "cuda", b=2, p=4096, seed=9)
    assert int((label_cls > 0).sum()) > 50
    g = torch.Generator().manual_seed(5)
    for case in ("plain", "upstream", "clipped", "no_foreground"):
        lab = torch.zeros_like(label_cls) if case == "no_foreground" else label_cls
        seg = torch.randn(2, 4096, cfg.num_classes + 1, generator=g).cuda()
        if case == "clipped":
            seg = seg * 40.0
        seg.requires_grad_(True)
        head = (2 * torch.randn(2, 4096, cfg.num_classes, cfg.head_width, generator=g)).cuda().requires_grad_(True)
        model = R_.RpnModel(cfg)                      # only its .loss is used: no device weights needed
        up = 0.37 if case == "upstream" else 1.0
        loss_f, parts_f = model.loss(xyz, seg, head, lab, label_reg, fused=True)
        gf = torch.autograd.grad(loss_f * up, (seg, head))
        loss_r, parts_r = model.loss(xyz, seg, head, lab, label_reg, fused=False)
        gr = torch.autograd.grad(loss_r * up, (seg, head), allow_unused=True)
        assert torch.allclose(loss_f, loss_r, rtol=2e-5, atol=1e-6), (case, float(loss_f), float(loss_r))
        for name in ("segmentation", "bin_classification", "regression", "num_foreground"):
            assert torch.allclose(parts_f[name].float(), parts_r[name].float(), rtol=2e-5, atol=1e-6), (case, name)
        for a, b in zip(gf, gr):
            b = torch.zeros_like(a) if b is None else b
            assert torch.allclose(a, b, rtol=1e-4, atol=1e-7 + 1e-5 * float(b.abs().max())), (case, float((a - b).abs().max()))


@pytest.mark.gpu
def test_full_size_pointnet_rpn_step_b8_properties():
    """hf/configs/rpn_cars_pointnet_paper.config at its own sizes, 8 frames of 16384 points (round 2's bench workload): the
    prefetched geometry gives bit-identical outputs to the inline geometry, the fused loss equals the op-by-op loss, every
    gradient is finite and non-trivial, three Adam steps reduce the loss"""
    import bench
    from heterofusionrcnn_amd.pipeline import GeometryPrefetcher
    cfg = R_.rpn_cars_pointnet_paper()
    rng = np.random.default_rng(21)
    xyz = torch.from_numpy(bench.kitti_uniform(rng, 8, bench.N0)).cuda()
    inten = torch.from_numpy(rng.uniform(-0.5, 0.5, (8, bench.N0, 1)).astype(np.float32)).cuda()
    gt_boxes, gt_cls = R_.synthetic_ground_truth(rng, 8, 12, cfg, ground_y=3.0)
    label_cls, label_reg = R_.point_labels(xyz, torch.from_numpy(gt_boxes).cuda(), torch.from_numpy(gt_cls).cuda())
    assert int((label_cls > 0).sum()) > 100
    torch.manual_seed(2)
    model = R_.RpnModel(cfg).cuda().train()
    model.backbone.fc_keep = [1.0] * len(model.backbone.fc_keep)          # dropout off: two evaluations must agree bit for bit
    model.heads.drop = [0.0] * len(model.heads.drop)
    pf = GeometryPrefetcher(model.geometry, depth=1, group=1)
    pf.submit(xyz)
    geo = pf.get()
    with torch.no_grad():
        seg_a, head_a = model(xyz, inten)
        for m in model.modules():                                         # the first pass moved the running statistics only
            pass
        seg_b, head_b = model(xyz, inten, geometry=geo)
    assert seg_a.shape == (8, bench.N0, 2) and head_a.shape == (8, bench.N0, 1, 76)
    assert torch.equal(seg_a, seg_b) and torch.equal(head_a, head_b)
    opt = torch.optim.Adam(model.parameters(), lr=1e-3, fused=True)
    losses = []
    for it in range(3):
        opt.zero_grad(set_to_none=True)
        seg, head = model(xyz, inten, geometry=geo)
        loss, parts = model.loss(xyz, seg, head, label_cls, label_reg)
        if it == 0:
            loss_ref, _ = model.loss(xyz, seg, head, label_cls, label_reg, fused=False)
            assert torch.allclose(loss, loss_ref, rtol=2e-5, atol=1e-6)
            assert float(parts["num_foreground"]) == float((label_cls > 0).sum())
        loss.backward()
        if it == 0:
            for n, p_ in model.named_parameters():
                assert p_.grad is not None and torch.isfinite(p_.grad).all(), n
            assert sum(float(p_.grad.abs().sum()) > 0 for p_ in model.parameters()) > 100
        opt.step()
        losses.append(float(loss.detach()))
    assert all(np.isfinite(losses)) and losses[-1] < losses[0], losses
