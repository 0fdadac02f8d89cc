"""CPU tests (-m "not gpu") of the data-parallel harness: world_size 2 over gloo.

The GPU path uses the same code with backend "nccl" (RCCL); here the model is a small torch MLP
because the HIP ops have no CPU implementation by design."""
import os
import socket

import torch
import torch.multiprocessing as mp
import torch.nn as nn

from heterofusionrcnn_amd import dp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _make_model():
    torch.manual_seed(7)
    return nn.Sequential(nn.Linear(6, 16), nn.ReLU(), nn.Linear(16, 3))


def _worker(rank, world, port, ret):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    ctx = dp.init(backend="gloo")
    assert ctx.world == world and ctx.rank == rank and ctx.distributed and ctx.device.type == "cpu"
    model = _make_model()
    if rank == 1:  # de-synchronise on purpose: wrap_model must broadcast rank 0's weights
        with torch.no_grad():
            for p in model.parameters():
                p.add_(1.0)
    net = dp.wrap_model(model, ctx)
    opt = torch.optim.SGD(net.parameters(), lr=dp.scaled_lr(0.05, ctx.world))
    g = torch.Generator().manual_seed(123)
    frames = torch.randn(8, 5, 6, generator=g)          # 8 "frames" of 5 rows
    target = torch.randn(8, 5, 3, generator=g)
    mine = dp.shard_frames(8, rank, world)
    losses = []
    for _ in range(3):
        opt.zero_grad()
        loss = ((net(frames[mine]) - target[mine]) ** 2).mean()
        loss.backward()
        opt.step()
        losses.append(loss.item())
    dp.fence(ctx)
    slowest = dp.max_over_ranks(1.0 + rank, ctx)
    gathered = dp.gather_objects({"rank": rank, "frames": mine}, ctx)
    flat = torch.cat([p.detach().flatten() for p in model.parameters()])
    ret[rank] = {"params": flat, "slowest": slowest, "gathered": gathered, "losses": losses}
    dp.shutdown(ctx)


def test_two_rank_gloo_data_parallel():
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, ret), nprocs=world, join=True)
    r0, r1 = ret[0], ret[1]
    # replicas stay bit-identical: broadcast at wrap time + averaged gradients every step
    assert torch.equal(r0["params"], r1["params"])
    assert r0["slowest"] == 2.0 and r1["slowest"] == 2.0
    assert r1["gathered"] is None and [g["rank"] for g in r0["gathered"]] == [0, 1]
    assert sorted(r0["gathered"][0]["frames"] + r0["gathered"][1]["frames"]) == list(range(8))
    # equivalence with one process on the full batch: mean over ranks of per-shard mean losses ==
    # mean over the whole batch (equal shard sizes), lr scaled by world as the reference does
    model = _make_model()
    opt = torch.optim.SGD(model.parameters(), lr=0.05 * world)
    g = torch.Generator().manual_seed(123)
    frames = torch.randn(8, 5, 6, generator=g)
    target = torch.randn(8, 5, 3, generator=g)
    for _ in range(3):
        opt.zero_grad()
        ((model(frames) - target) ** 2).mean().backward()
        opt.step()
    single = torch.cat([p.detach().flatten() for p in model.parameters()])
    torch.testing.assert_close(r0["params"], single, rtol=1e-5, atol=1e-6)


def test_shard_and_schedule_helpers():
    for n, w in ((8, 2), (13, 4), (3, 8), (0, 2)):
        shards = [dp.shard_frames(n, r, w) for r in range(w)]
        assert sorted(sum(shards, [])) == list(range(n))
        assert max(map(len, shards)) - min(map(len, shards)) <= 1
    assert dp.scaled_lr(1e-3, 8) == 8e-3                      # optimizer_builder.py:105
    assert dp.steps_per_rank(240000, 8) == 30000              # trainer.py:147
    ctx = dp.DPContext(0, 1, 0, torch.device("cpu"))
    assert not ctx.distributed and dp.max_over_ranks(3.5, ctx) == 3.5 and dp.gather_objects("x", ctx) == ["x"]
    m = _make_model()
    assert dp.wrap_model(m, ctx) is m
