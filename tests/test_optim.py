"""optim.MultiTensorAdam (csrc/optim.hip: the Adam update of every parameter tensor in one launch) against torch.optim.Adam
(same placement of epsilon: tf_epsilon=False) and against a restatement of tf.train.AdamOptimizer's formula in fp64."""
import numpy as np
import pytest
import torch


def _params(seed, shapes, device):
    g = torch.Generator().manual_seed(seed)
    return [torch.randn(s, generator=g).to(device).requires_grad_(True) for s in shapes]


SHAPES = [(3,), (64, 3), (16385,), (257, 129), (1, 1), (40000,), (7, 5, 3), (512, 512)]   # chunk edges (16384), odd lengths, unaligned slices


@pytest.mark.gpu
def test_multi_tensor_adam_matches_torch_adam():
    from heterofusionrcnn_amd.optim import MultiTensorAdam
    a, b = _params(0, SHAPES, "cuda"), _params(0, SHAPES, "cuda")
    oa = MultiTensorAdam(a, lr=3e-3, betas=(0.9, 0.999), eps=1e-8, tf_epsilon=False)
    ob = torch.optim.Adam(b, lr=3e-3, betas=(0.9, 0.999), eps=1e-8)
    gen = torch.Generator().manual_seed(1)
    for step in range(12):
        for pa, pb in zip(a, b):
            g = torch.randn(pa.shape, generator=gen).cuda() * (10.0 ** (step % 3 - 1))
            pa.grad, pb.grad = g.clone(), g.clone()
        if step == 5:                                            # a tensor without a gradient is skipped for that step
            a[2].grad = None
            b[2].grad = None
        oa.step()
        ob.step()
    for pa, pb in zip(a, b):
        if pa.shape == (16385,):
            continue                                             # torch's per-tensor step count differs after the skipped step
        np.testing.assert_allclose(pa.detach().cpu().numpy(), pb.detach().cpu().numpy(), rtol=2e-5, atol=2e-6)


@pytest.mark.gpu
def test_multi_tensor_adam_tensorflow_formula_and_grad_scale():
    """tf.train.AdamOptimizer: lr_t = lr sqrt(1 - b2^t) / (1 - b1^t); p -= lr_t m / (sqrt(v) + eps); gradients scaled by 1 / world on load"""
    from heterofusionrcnn_amd.optim import MultiTensorAdam
    ps = _params(3, SHAPES[:5], "cuda")
    ref = [p.detach().cpu().double().numpy().copy() for p in ps]
    m = [np.zeros_like(r) for r in ref]
    v = [np.zeros_like(r) for r in ref]
    lr, b1, b2, eps, scale = 1e-2, 0.9, 0.999, 1e-8, 0.125
    opt = MultiTensorAdam(ps, lr=lr, betas=(b1, b2), eps=eps, tf_epsilon=True, grad_scale=scale)
    gen = torch.Generator().manual_seed(4)
    for t in range(1, 9):
        for i, p in enumerate(ps):
            g = torch.randn(p.shape, generator=gen)
            p.grad = g.cuda()
            gd = g.double().numpy() * scale
            m[i] = b1 * m[i] + (1 - b1) * gd
            v[i] = b2 * v[i] + (1 - b2) * gd * gd
            ref[i] -= lr * np.sqrt(1 - b2 ** t) / (1 - b1 ** t) * m[i] / (np.sqrt(v[i]) + eps)
        opt.step()
    for p, r in zip(ps, ref):
        np.testing.assert_allclose(p.detach().cpu().numpy(), r, rtol=3e-5, atol=3e-6)
    snap = opt.snapshot()
    before = [p.detach().clone() for p in ps]
    opt.step()
    opt.restore(snap)
    assert float(opt.step_count) == 8.0
    for p, q in zip(ps, before):
        assert not torch.equal(p, q)                             # parameters are the caller's to restore, the moments are ours


@pytest.mark.gpu
def test_multi_tensor_adam_inside_a_captured_graph():
    """gradients at fixed addresses (a captured step): the table is uploaded once per capture, every replay advances the counter"""
    from heterofusionrcnn_amd.optim import MultiTensorAdam
    a, b = _params(5, SHAPES[:4], "cuda"), _params(5, SHAPES[:4], "cuda")
    oa = MultiTensorAdam(a, lr=1e-2, tf_epsilon=False)
    ob = torch.optim.Adam(b, lr=1e-2)
    x = torch.randn(64, device="cuda")

    def loss(ps):
        return sum(((p * 1.5).sin() ** 2).sum() for p in ps) * x.mean()

    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for p in a:
            p.grad = None
        loss(a).backward()                                       # lazy initialisations outside the capture
        for p in a:
            p.grad = None
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for p in a:
            p.grad = None
        loss(a).backward()
        oa.step()
    for _ in range(5):
        g.replay()
        for p in b:
            p.grad = None
        loss(b).backward()
        ob.step()
    torch.cuda.synchronize()
    assert float(oa.step_count) == 5.0
    for pa, pb in zip(a, b):
        np.testing.assert_allclose(pa.detach().cpu().numpy(), pb.detach().cpu().numpy(), rtol=2e-5, atol=2e-6)


@pytest.mark.gpu
def test_multi_tensor_adam_gradients_at_new_addresses_every_step():
    """eager steps whose gradients never sit where the previous step's did (the old ones are kept alive): every step rewrites
    and uploads the table while the host is ahead of the device -- the staging ring must hand each upload its own bytes"""
    from heterofusionrcnn_amd.optim import MultiTensorAdam
    a, b = _params(9, SHAPES, "cuda"), _params(9, SHAPES, "cuda")
    oa = MultiTensorAdam(a, lr=1e-2, tf_epsilon=False)
    ob = torch.optim.Adam(b, lr=1e-2)
    big = torch.randn(4096, 4096, device="cuda")
    keep = []
    for step in range(12):
        for ps, opt in ((a, oa), (b, ob)):
            for p in ps:
                p.grad = None
            (sum(((p * 1.5).sin() ** 2).sum() for p in ps) * (1.0 + step)).backward()
            if ps is a:
                keep.append([p.grad for p in ps])
                for _ in range(3):
                    big = big @ big * 1e-4          # device work the host runs ahead of
            opt.step()
    torch.cuda.synchronize()
    assert len({k[0].data_ptr() for k in keep}) == len(keep)
    for pa, pb in zip(a, b):
        np.testing.assert_allclose(pa.detach().cpu().numpy(), pb.detach().cpu().numpy(), rtol=2e-5, atol=2e-6)


def test_multi_tensor_adam_has_no_cpu_path():
    from heterofusionrcnn_amd.optim import MultiTensorAdam
    with pytest.raises(RuntimeError, match="no CPU implementation"):
        MultiTensorAdam([torch.zeros(3, requires_grad=True)])
