"""The captured train step (heterofusionrcnn_amd/graph_step.py) and the fusion step it contains.

CPU (-m "not gpu"): the path-drop decision against a literal restatement of create_path_drop_masks
(hf/core/models/rpn_model.py:1130-1193); flat gradients + one all-reduce over a world-size-2 gloo group against the mean
of the two ranks' gradients (hvd.DistributedOptimizer, hf/core/trainer.py:71).
GPU (-m gpu): hf_fuse_concat (+grad) against mul + cat; the graph replay against the eager step over 20 steps (dropout off:
same trajectory; dropout on: every replay draws fresh masks); the rpn_multiclass.config step at its full width and B = 8
with the image-feature fusion (finite, gradient reaches the image feature map, prefetched geometry == inline, one X-Conv
level at full width == its op-by-op form)."""
import dataclasses
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp
import torch.nn as nn
import torch.nn.functional as F

from heterofusionrcnn_amd import fusion
from heterofusionrcnn_amd.graph_step import FlatGrads, TrainStep, broadcast_parameters, tree_tensors


# ------------------------------------------------------------------------------------------------ CPU
def _literal_path_drop(p_img, p_bev, r):
    """rpn_model.py:1130-1193 step by step (tf.case chains written as ifs)"""
    img_chances = 1.0 if r[0] < p_img else 0.0
    bev_chances = 1.0 if r[1] < p_bev else 0.0
    third_flip = 1.0 if (bool(img_chances) or bool(bev_chances)) else 0.0
    img_second_flip = 1.0 if r[2] > 0.5 else 0.0
    bev_second_flip = 1.0 if r[2] <= 0.5 else 0.0
    final_img = img_chances if third_flip == 1 else img_second_flip
    final_bev = bev_chances if third_flip == 1 else bev_second_flip
    return final_img, final_bev


def test_path_drop_masks_against_literal_restatement():
    rng = np.random.default_rng(0)
    cases = [rng.random(3) for _ in range(200)] + [np.array([0.95, 0.95, 0.5]), np.array([0.95, 0.95, 0.51]), np.array([0.9, 0.9, 0.2])]
    seen = set()
    for r in cases:
        for (pi, pp) in ((0.9, 0.9), (0.5, 0.3), (1.0, 1.0), (0.0, 0.0)):
            img, pc = _literal_path_drop(pi, pp, r)
            got = fusion.path_drop_masks(pi, pp, torch.tensor(r, dtype=torch.float32))
            assert got.tolist() == [pc, img], (r, pi, pp)
            seen.add((img, pc))
    assert seen == {(1.0, 1.0), (1.0, 0.0), (0.0, 1.0)}      # never both dropped


def test_fuse_concat_cpu_form_and_mean_fusion():
    a, b = torch.randn(2, 5, 4), torch.randn(2, 5, 3)
    m = torch.tensor([0.0, 1.0])
    out = fusion.fuse_point_image_features(a, b, "concat", masks=m)
    assert torch.equal(out, torch.cat([a * 0.0, b], -1))
    c = torch.randn(2, 5, 4)
    assert torch.allclose(fusion.fuse_point_image_features(a, c, "mean"), (a + c) / 2)
    assert torch.allclose(fusion.fuse_point_image_features(a, c, "mean", masks=m), c)      # divided by img_mask + pc_mask = 1


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _flat_worker(rank, world, port, ret):
    from heterofusionrcnn_amd import dp
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    ctx = dp.init(backend="gloo")
    torch.manual_seed(7 + rank)                               # de-synchronised on purpose
    model = nn.Sequential(nn.Linear(6, 16), nn.ReLU(), nn.Linear(16, 3))
    broadcast_parameters(model)
    opt = torch.optim.SGD(model.parameters(), lr=dp.scaled_lr(0.05, world))
    g = torch.Generator().manual_seed(123)
    frames, target = torch.randn(8, 5, 6, generator=g), torch.randn(8, 5, 3, generator=g)
    mine = dp.shard_frames(8, rank, world)
    inputs = {"x": frames[mine], "y": target[mine]}
    step = TrainStep(model, opt, inputs, geometry={}, world=world, graph=False,
                     loss_fn=lambda m, inp, geo: ((m(inp["x"]) - inp["y"]) ** 2).mean())
    losses = [float(step()) for _ in range(3)]
    lo, hi = step.grads.flat.data_ptr(), step.grads.flat.data_ptr() + step.grads.flat.numel() * 4
    ret[rank] = {"params": torch.cat([p.detach().flatten() for p in model.parameters()]), "losses": losses,
                 "views": all(lo <= p.grad.data_ptr() < hi for p in model.parameters())}
    dp.shutdown(ctx)


def test_flat_gradient_all_reduce_two_ranks_gloo():
    world, port = 2, _free_port()
    ret = mp.Manager().dict()
    mp.spawn(_flat_worker, args=(world, port, ret), nprocs=world, join=True)
    assert torch.equal(ret[0]["params"], ret[1]["params"]) and ret[0]["views"] and ret[1]["views"]
    # one process, whole batch, same lr * world: the mean of the two shard losses' gradients = the gradient of the mean loss
    torch.manual_seed(7)
    model = nn.Sequential(nn.Linear(6, 16), nn.ReLU(), nn.Linear(16, 3))
    opt = torch.optim.SGD(model.parameters(), lr=0.1)
    g = torch.Generator().manual_seed(123)
    frames, target = torch.randn(8, 5, 6, generator=g), torch.randn(8, 5, 3, generator=g)
    for _ in range(3):
        opt.zero_grad()
        ((model(frames) - target) ** 2).mean().backward()
        opt.step()
    ref = torch.cat([p.detach().flatten() for p in model.parameters()])
    assert torch.allclose(ret[0]["params"], ref, rtol=1e-5, atol=1e-6)


class _ToyBackbone(nn.Module):
    """an encoder with skip connections into a decoder, the interface TrainStep cuts at: backbone.enc + forward(..., taps=)"""

    def __init__(self):
        super().__init__()
        self.enc = nn.ModuleList([nn.Linear(4, 8), nn.Linear(8, 8)])
        self.dec = nn.Linear(16, 6)

    def forward(self, xyz, intensity, geometry=None, taps=None):
        f1 = torch.tanh(self.enc[0](torch.cat([xyz, intensity], -1)))
        f2 = torch.tanh(self.enc[1](f1))
        if taps is not None:                                   # the decoder reads detached copies (a clean cut)
            c1, c2 = f1.detach().requires_grad_(True), f2.detach().requires_grad_(True)
            taps.extend([(f1, c1), (f2, c2)])
            f1, f2 = c1, c2
        return self.dec(torch.cat([f1, f2], -1))              # BOTH encoder outputs cross the cut


class _ToyRpn(nn.Module):
    def __init__(self):
        super().__init__()
        self.backbone = _ToyBackbone()
        self.head = nn.Linear(6, 2)

    def forward(self, xyz, intensity, geometry=None, img_fts=None, calib=None, taps=None):
        f = self.backbone(xyz, intensity, geometry, taps=taps)
        return self.head(f), f

    def loss(self, xyz, seg_logits, head, label_cls, label_reg):
        return ((seg_logits - label_reg[..., :2]) ** 2).mean() + 0.1 * (head ** 2).mean(), None


def _split_worker(rank, world, port, ret):
    from heterofusionrcnn_amd import dp
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    ctx = dp.init(backend="gloo")
    g = torch.Generator().manual_seed(5)
    xyz, inten, reg = torch.randn(8, 7, 3, generator=g), torch.randn(8, 7, 1, generator=g), torch.randn(8, 7, 7, generator=g)
    mine = dp.shard_frames(8, rank, world)
    out = {}
    for overlap in (True, False):
        torch.manual_seed(11 + rank)
        model = _ToyRpn()
        broadcast_parameters(model)
        opt = torch.optim.SGD(model.parameters(), lr=dp.scaled_lr(0.05, world))
        inputs = {"xyz": xyz[mine], "intensity": inten[mine], "label_cls": None, "label_reg": reg[mine]}
        step = TrainStep(model, opt, inputs, geometry={}, world=world, graph=False, overlap_exchange=overlap)
        assert step.split == overlap and step.chunks == (2 if overlap else 1)
        if overlap:                                            # chunk 0 = everything after the encoder, chunk 1 = the encoder
            enc = {id(p) for p in model.backbone.enc.parameters()}
            assert [id(p) in enc for p in step.grads.params] == [False] * len(step.late) + [True] * len(step.early)
        losses = [float(step()) for _ in range(4)]
        out[overlap] = (torch.cat([p.detach().flatten() for p in model.parameters()]), losses)
    ret[rank] = out
    dp.shutdown(ctx)


def test_chunked_gradient_exchange_two_ranks_gloo():
    """the backward pass cut at the encoder's outputs, the late parameters' chunk of the flat buffer
    reduced while the encoder's backward runs, then the encoder's chunk -- same trajectory as the single exchange, on both ranks"""
    world, port = 2, _free_port()
    ret = mp.Manager().dict()
    mp.spawn(_split_worker, args=(world, port, ret), nprocs=world, join=True)
    for r in range(world):
        assert torch.allclose(ret[r][True][0], ret[r][False][0], rtol=1e-5, atol=1e-6)
        assert np.allclose(ret[r][True][1], ret[r][False][1], rtol=1e-5)
    assert torch.equal(ret[0][True][0], ret[1][True][0])


def test_tree_tensors_order_is_stable():
    geo = {"pts": [torch.zeros(1), torch.ones(2)], "enc": (torch.ones(3),), "dec": []}
    a = [t.numel() for t in tree_tensors(geo)]
    b = [t.numel() for t in tree_tensors(dict(reversed(list(geo.items()))))]
    assert a == b == [3, 1, 2]


# ------------------------------------------------------------------------------------------------ GPU
def _small_multiclass(img_c=8, dropout=0.0, path_drop=(1.0, 1.0)):
    from heterofusionrcnn_amd import pointcnn as P_, rpn as R_
    small = P_.PointCnnConfig(xconv=((8, 1, -1, 32), (8, 1, 512, 32), (8, 1, 128, 64), (8, 1, 32, 128), (8, 1, 16, 128)),
                              fc=((32, dropout), (32, dropout)))
    return dataclasses.replace(R_.rpn_multiclass(img_c), pointcnn=small, rpn_fc=((64, dropout), (64, dropout)), path_drop=path_drop)


def _frames(cfg, b, p, seed, h=24, w=80):
    from heterofusionrcnn_amd import rpn as R_
    rng = np.random.default_rng(seed)
    xyz = np.stack([rng.uniform(-8, 8, (b, p)), rng.uniform(-1.0, 1.7, (b, p)), rng.uniform(2, 18, (b, p))], -1).astype(np.float32)
    boxes, cls = R_.synthetic_ground_truth(rng, b, 10, cfg, extent=((-7.0, 7.0), (3.0, 17.0)))
    xyz_t = torch.from_numpy(xyz).cuda()
    inp = {"xyz": xyz_t, "intensity": torch.from_numpy(rng.uniform(-0.5, 0.5, (b, p, 1)).astype(np.float32)).cuda()}
    inp["label_cls"], inp["label_reg"] = R_.point_labels(xyz_t, torch.from_numpy(boxes).cuda(), torch.from_numpy(cls).cuda())
    if cfg.img_channels:
        inp["img_fts"] = torch.from_numpy(rng.standard_normal((b, h, w, cfg.img_channels)).astype(np.float32)).cuda().requires_grad_(True)
        # a pinhole that maps x in [-8,8], z in [2,18] into most of the (h, w) map
        inp["calib"] = torch.tensor([[w / 2.0, 0, w / 2.0, 0], [0, h / 2.0, h / 2.0, 0], [0, 0, 1, 0]], device="cuda").repeat(b, 1, 1)
    return inp


@pytest.mark.gpu
def test_fuse_concat_kernel_against_mul_and_cat():
    g = torch.Generator().manual_seed(0)
    for rows, c1, c2 in ((1, 4, 4), (1000, 256, 32), (77, 5, 3), (4096, 256, 64)):
        a = torch.randn(rows, c1, generator=g).cuda().requires_grad_(True)
        b = torch.randn(rows, c2, generator=g).cuda().requires_grad_(True)
        go = torch.randn(rows, c1 + c2, generator=g).cuda()
        for masks in (None, torch.tensor([1.0, 0.0], device="cuda"), torch.tensor([0.0, 1.0], device="cuda")):
            out = fusion.fuse_point_image_features(a, b, "concat", masks=masks)
            ga, gb = torch.autograd.grad(out, (a, b), go)
            m0, m1 = (1.0, 1.0) if masks is None else masks.tolist()
            assert torch.equal(out, torch.cat([a * m0, b * m1], -1))
            assert torch.equal(ga, go[:, :c1] * m0) and torch.equal(gb, go[:, c1:] * m1)


@pytest.mark.gpu
def test_graph_replay_reproduces_the_eager_trajectory():
    """same weights, same frames, dropout and path drop off: 20 steps of the captured step against 20 eager steps.  The
    scatter gradients use atomics (summation order differs run to run), so the trajectories agree to rounding, not bit for bit."""
    from heterofusionrcnn_amd import rpn as R_
    cfg = _small_multiclass()
    inp = _frames(cfg, 2, 2048, seed=0)
    losses = {}
    for mode in ("eager", "graph"):
        torch.manual_seed(4)
        model = R_.RpnModel(cfg).cuda().train()
        opt = torch.optim.Adam(model.parameters(), lr=2e-3, fused=True, capturable=True)
        geo = model.geometry(inp["xyz"])
        step = TrainStep(model, opt, inp, geo, world=1, graph=(mode == "graph"), warmup=0 if mode == "eager" else 3)
        # the graph constructor's 3 warm-up steps are undone before the capture (parameters, BatchNorm buffers, Adam state):
        # both modes start from the seeded initial state, the eager loop runs no extra steps
        losses[mode] = [float(step(geometry=geo)) for _ in range(20)]
        if mode == "graph":
            assert step.graph is not None and step.opt_in_graph
            g = inp["img_fts"].grad
            assert g is not None and torch.isfinite(g).all() and float(g.abs().sum()) > 0
    e, g = np.array(losses["eager"]), np.array(losses["graph"])
    assert np.all(np.isfinite(e)) and np.all(np.isfinite(g))
    assert e[-1] < 0.9 * e[0] and g[-1] < 0.9 * g[0], (e, g)
    np.testing.assert_allclose(g, e, rtol=2e-2)
    np.testing.assert_allclose(g[:3], e[:3], rtol=1e-4)


@pytest.mark.gpu
@pytest.mark.parametrize("graph", [False, True])
def test_x_branch_on_a_side_stream_gives_the_same_step(graph):
    """pointcnn.CONCURRENT_X_BRANCH: the X-transformation branch of every X-Conv enqueued on a side HIP stream (forward and, through
    autograd's stream bookkeeping, backward) must not change a step: same losses over 6 steps, eager and replayed"""
    from heterofusionrcnn_amd import pointcnn, rpn as R_
    cfg = _small_multiclass()
    inp = _frames(cfg, 2, 2048, seed=3)
    losses = {}
    for flag in (False, True):
        pointcnn.CONCURRENT_X_BRANCH = flag
        try:
            torch.manual_seed(9)
            model = R_.RpnModel(cfg).cuda().train()
            opt = torch.optim.Adam(model.parameters(), lr=2e-3, fused=True, capturable=True)
            geo = model.geometry(inp["xyz"])
            step = TrainStep(model, opt, inp, geo, world=1, graph=graph, warmup=2)
            losses[flag] = [float(step(geometry=geo)) for _ in range(6)]
            torch.cuda.synchronize()
        finally:
            pointcnn.CONCURRENT_X_BRANCH = False
    np.testing.assert_allclose(losses[True], losses[False], rtol=2e-4)
    assert losses[True][-1] < losses[True][0]


@pytest.mark.gpu
def test_graph_replay_draws_fresh_dropout_and_path_drop_masks():
    from heterofusionrcnn_amd import rpn as R_
    cfg = _small_multiclass(dropout=0.5, path_drop=(0.5, 0.5))
    inp = _frames(cfg, 2, 2048, seed=1)
    torch.manual_seed(5)
    model = R_.RpnModel(cfg).cuda().train()
    opt = torch.optim.Adam(model.parameters(), lr=0.0, fused=True, capturable=True)      # frozen weights: only the masks change
    geo = model.geometry(inp["xyz"])
    step = TrainStep(model, opt, inp, geo, world=1, graph=True)
    vals = [round(float(step(geometry=geo)), 6) for _ in range(12)]
    assert len(set(vals)) >= 10, vals
    # new geometry / new frames go through the static slots
    inp2 = _frames(cfg, 2, 2048, seed=2)
    geo2 = model.geometry(inp2["xyz"])
    a = float(step(geometry=geo2, xyz=inp2["xyz"], intensity=inp2["intensity"], label_cls=inp2["label_cls"], label_reg=inp2["label_reg"]))
    assert np.isfinite(a)
    assert all(torch.equal(s, t) for s, t in zip(tree_tensors(step.geometry), tree_tensors(geo2)))


@pytest.mark.gpu
def test_rpn_multiclass_full_width_step_with_fusion_b8():
    """BASELINE config 4 at its own sizes: rpn_multiclass.config's PointCNN at full width, 8 frames of 16384 points, a
    (8,360,1200,32) image feature map, path drop, concat fusion, three-class heads, losses, backward, Adam -- captured and
    replayed; prefetched geometry == inline geometry; the gradient reaches the image feature map at the projected pixels."""
    import bench
    from heterofusionrcnn_amd import rpn as R_
    from heterofusionrcnn_amd.pipeline import GeometryPrefetcher
    cfg = R_.rpn_multiclass(bench.IMG_C)
    rng = np.random.default_rng(11)
    b = 8
    xyz = torch.from_numpy(bench.kitti_frustum(rng, b, bench.N0)).cuda()
    inp = {"xyz": xyz, "intensity": torch.from_numpy(rng.uniform(-0.5, 0.5, (b, bench.N0, 1)).astype(np.float32)).cuda()}
    gt_boxes, gt_cls = R_.synthetic_ground_truth(rng, b, 12, cfg, ground_y=3.0)
    inp["label_cls"], inp["label_reg"] = R_.point_labels(xyz, torch.from_numpy(gt_boxes).cuda(), torch.from_numpy(gt_cls).cuda())
    inp["img_fts"] = torch.randn(b, bench.IMG_H, bench.IMG_W, bench.IMG_C, device="cuda").requires_grad_(True)
    inp["calib"] = torch.from_numpy(bench.KITTI_P2).cuda().repeat(b, 1, 1).contiguous()
    torch.manual_seed(9)
    model = R_.RpnModel(cfg).cuda().train()
    geo = model.geometry(xyz)
    pf = GeometryPrefetcher(model.geometry, depth=1, group=1)
    pf.submit(xyz)
    geo_pf = pf.get()
    assert all(torch.equal(s, t) for s, t in zip(tree_tensors(geo), tree_tensors(geo_pf)))
    opt = torch.optim.Adam(model.parameters(), lr=1e-3, fused=True, capturable=True)
    step = TrainStep(model, opt, inp, geo, world=1, graph=True, warmup=2)
    losses = [float(step(geometry=geo_pf)) for _ in range(4)]
    assert all(np.isfinite(losses)), losses
    g = inp["img_fts"].grad
    assert g is not None and g.shape == inp["img_fts"].shape and torch.isfinite(g).all()
    # every point of the frustum-filtered frames has a pixel; gradient only at those pixels
    _, pix = fusion.project_gather(xyz, inp["calib"], inp["img_fts"].detach(), return_pixels=True)
    assert bool(((pix[..., 0] >= 0) & (pix[..., 0] < bench.IMG_W) & (pix[..., 1] >= 0) & (pix[..., 1] < bench.IMG_H)).all())
    touched = torch.zeros(b, bench.IMG_H, bench.IMG_W, dtype=torch.bool, device="cuda")
    bi = torch.arange(b, device="cuda")[:, None].expand(-1, bench.N0)
    touched[bi, pix[..., 1].long(), pix[..., 0].long()] = True
    assert float(g[~touched].abs().max()) == 0.0
    for n, p_ in model.named_parameters():
        assert torch.isfinite(p_).all(), n


@pytest.mark.gpu
def test_full_width_xconv_level_equals_op_by_op_form():
    """the second encoder level of rpn_multiclass.config at full width (16384 -> 4096 points, K = 8, 256 gathered channels,
    C = 256): the HIP route (grid kNN, gather into the concat buffer, fused X-apply + depthwise, split-K weight gradients,
    ELU-on-load BatchNorm) against plain torch ops on the tensors the reference materialises, values and gradients"""
    import heterofusionrcnn_amd as hf
    from heterofusionrcnn_amd import pointcnn as P_
    import bench
    torch.manual_seed(21)
    rng = np.random.default_rng(21)
    b, n, p, k, cprev, c = 2, 16384, 4096, 8, 256, 256
    pts = torch.from_numpy(bench.kitti_frustum(rng, b, n)).cuda()
    qrs = hf.gather_point(pts, hf.farthest_point_sample(p, pts))
    fts = torch.randn(b, n, cprev, device="cuda").requires_grad_(True)
    m = P_.XConv(k, 1, cprev, c, 64, 1, with_x=True, with_global=False).cuda().train()
    out = m(pts, fts, qrs)
    go = torch.randn_like(out)
    params = list(m.parameters())
    grads = torch.autograd.grad(out, [fts] + params, go)

    def bn(x, mod):
        x2 = x.reshape(-1, x.shape[-1])
        mu, var = x2.mean(0), x2.var(0, unbiased=False)
        return ((x - mu) / torch.sqrt(var + 1e-3)) * mod.bn.weight + mod.bn.bias

    def dense(x, mod):
        y = F.linear(x, mod.linear.weight)
        return bn(F.elu(y) if mod.post.activation else y, mod.post)

    def dw(x, mod):
        kk, cc, mm = mod.weight.shape
        y = torch.einsum("bpwc,wcm->bpcm", x, mod.weight).reshape(x.shape[0], x.shape[1], cc * mm)
        return bn(F.elu(y) if mod.post.activation else y, mod.post)

    _, idx = hf.knn_point(k, pts, qrs)
    bi = torch.arange(b, device="cuda")[:, None, None]
    idl = idx.long()
    local = pts[bi, idl] - qrs[:, :, None]
    f = torch.cat([dense(dense(local, m.lift0), m.lift1), fts[bi, idl]], -1)
    x0 = dense(local.reshape(b, p, 1, k * 3), m.x0).reshape(b, p, k, k)
    x1 = dw(x0, m.x1).reshape(b, p, k, k)
    x2 = dw(x1, m.x2).reshape(b, p, k, k)
    fx = torch.matmul(x2, f)
    y = torch.einsum("bpkc,kcm->bpcm", fx, m.conv.depthwise).reshape(b, p, -1)
    ref = bn(F.elu(F.linear(y, m.conv.pointwise.weight)), m.conv.post)
    ref_grads = torch.autograd.grad(ref, [fts] + params, go)
    torch.testing.assert_close(out, ref, rtol=2e-3, atol=2e-3)
    names = ["fts"] + [nm for nm, _ in m.named_parameters()]
    # a few gradients are analytically zero (x1's BatchNorm shift is removed again by x2's mean subtraction): both routes
    # return rounding noise there, so the floor of the bound is a fraction of the LARGEST parameter gradient
    gmax = max(float(r.abs().max()) for r in ref_grads[1:])
    for nm, a, r in zip(names, grads, ref_grads):
        scale = float(r.abs().max()) + 1e-6
        assert float((a - r).abs().max()) <= 5e-3 * scale + 1e-4 * gmax, (nm, float((a - r).abs().max()), scale, gmax)


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["weak", "strong"])
def test_bench_two_ranks_rehearsed_on_one_gpu(mode):
    """`bench.py --gpus 2` end to end with both ranks on this GPU and gloo collectives (RCCL needs one GPU per rank): the
    self-launch, rank-sharded frames, parameter broadcast, the gradient exchange (weak: DistributedDataParallel at 8 frames per
    rank; strong: the captured step at 4 frames per rank, flat gradient buffer, one all-reduce), barrier + max-over-ranks timing,
    ONE JSON line from rank 0"""
    import json, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--rehearse-on-one-gpu", "--scaling", mode, "--steps", "4",
                          "--warmup", "1", "--no-op-table", "--no-cpu-baseline", "--no-side-runs"], capture_output=True, text=True, timeout=500, env=env)
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert out.returncode == 0 and len(lines) == 1, out.stdout[-1500:] + out.stderr[-3000:]
    d = json.loads(lines[0])
    per = 8 if mode == "weak" else 4
    assert d["n_gpus"] == 2 and d["scaling"] == mode and d["config"]["frames_per_gpu"] == per and d["config"]["global_batch"] == 2 * per
    assert d["config"]["hip_graph"] == (mode == "strong") and np.isfinite(d["value"]) and d["value"] > 0
    assert "rehearsal" in d["config"]


@pytest.mark.gpu
def test_copy_many_mixed_dtypes_one_launch():
    """graph_step.copy_many (hf_copy_multi): 150 pairs of mixed dtypes, odd byte counts, unaligned views, an empty tensor and a
    non-contiguous pair (which goes to the framework) -- every destination equals its source, nothing beside them is touched"""
    from heterofusionrcnn_amd.graph_step import copy_many
    g = torch.Generator().manual_seed(3)
    dst, src, guards = [], [], []
    dtypes = [torch.float32, torch.int32, torch.int64, torch.uint8, torch.float64, torch.int16]
    for i in range(150):
        dt = dtypes[i % len(dtypes)]
        n = [0, 1, 3, 17, 255, 4096, 70001, 300000][i % 8]
        s = (torch.rand(n + 9, generator=g) * 100).to(dt).cuda()
        d = torch.full((n + 9,), 5, dtype=dt, device="cuda")
        off = i % 5                                  # views that start off the 16-byte grid
        src.append(s[off:off + n]); dst.append(d[off:off + n]); guards.append((d, off, n))
    a = torch.arange(24, device="cuda", dtype=torch.float32).view(4, 6)
    b = torch.zeros(6, 4, device="cuda")
    src.append(a.t()); dst.append(b)                 # non-contiguous source
    copy_many(dst, src)
    torch.cuda.synchronize()
    for d, s in zip(dst, src):
        assert torch.equal(d, s)
    for d, off, n in guards:
        assert (d[:off] == 5).all() and (d[off + n:] == 5).all()
