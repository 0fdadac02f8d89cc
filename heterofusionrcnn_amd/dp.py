"""Batch-data-parallel harness: one process per GPU, torch.distributed over RCCL (backend "nccl" on
ROCm), launched by `python -m torch.distributed.run`.  Replaces the reference's
mpirun + Horovod/NCCL launch (hf/experiments/mpi_run_training.sh:16-19) and keeps its semantics:

  hvd.init()                              -> init()                      (run_training.py:86)
  hvd.local_rank() pins the GPU           -> init() sets the device      (trainer.py:112-118)
  hvd.broadcast_global_variables(0)       -> DDP's initial broadcast     (trainer.py:73,144)
  hvd.DistributedOptimizer (avg all-reduce)-> DDP bucketed all-reduce    (trainer.py:71)
  lr * hvd.size()                         -> scaled_lr()                 (optimizer_builder.py:105)
  iterations / hvd.size()                 -> steps_per_rank()            (trainer.py:147)

The path itself shards over frames with no data-path collective (every op indexes the batch in
blockIdx); the only exchange is the gradient all-reduce of the train step.  BatchNorm statistics stay
per replica, as in the reference (pointfly.py:371-380).  Unlike the reference (every rank samples the
whole dataset with its own unseeded RNG, kitti_dataset.py:776-799) frames are sharded rank-strided.
"""
import os

import torch
import torch.distributed as dist


class DPContext:
    def __init__(self, rank, world, local_rank, device):
        self.rank, self.world, self.local_rank, self.device = rank, world, local_rank, device

    @property
    def distributed(self):
        return self.world > 1 or (dist.is_available() and dist.is_initialized())


def init(backend=None, share_gpu=False):
    """Read RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* from the environment (torch.distributed.run sets
    them) and join the process group.  backend: "nccl" (= RCCL) on GPUs, "gloo" on CPU.
    share_gpu: every rank uses GPU 0 and the collectives go over gloo (CUDA tensors staged through the host) -- a rehearsal of
    the N > 1 code path on a one-GPU box; RCCL itself needs one GPU per rank."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    use_cuda = torch.cuda.is_available()
    if backend is None:
        backend = "nccl" if use_cuda else "gloo"
    if share_gpu and use_cuda:
        backend = "gloo"
        device = torch.device("cuda", 0)
    else:
        device = torch.device("cuda", local_rank) if (use_cuda and backend == "nccl") else torch.device("cpu")
    if device.type == "cuda":
        torch.cuda.set_device(device)
    # HF_FORCE_DDP=1 joins a one-rank group too, so the whole DDP code path can be exercised on a single GPU
    force = os.environ.get("HF_FORCE_DDP", "") == "1" and "MASTER_PORT" in os.environ
    if (world > 1 or force) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        kwargs = {"device_id": device} if device.type == "cuda" else {}
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **kwargs)
    return DPContext(rank, world, local_rank, device)


def shutdown(ctx):
    if ctx.distributed and dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


def shard_frames(num_frames, rank, world):
    """rank-strided shard of frame ids 0..num_frames-1: disjoint, complete, sizes differ by at most one"""
    return list(range(rank, num_frames, world))


def scaled_lr(base_lr, world):
    """optimizer_builder.py:105: learning rate times the number of replicas"""
    return base_lr * world


def steps_per_rank(total_iterations, world):
    """trainer.py:147: the global iteration budget is split over the replicas"""
    return total_iterations // world


def wrap_model(model, ctx, bucket_cap_mb=64):
    """DistributedDataParallel = initial broadcast from rank 0 + bucketed gradient all-reduce (mean),
    overlapped with backward.  The SA/FP stack is ~1 MB of fp32 parameters: one bucket."""
    if not ctx.distributed:
        return model
    from torch.nn.parallel import DistributedDataParallel
    if ctx.device.type == "cuda":
        # per-replica BatchNorm statistics as under Horovod (hf/core/trainer.py:71 averages gradients only):
        # no per-step broadcast of the running buffers
        return DistributedDataParallel(model, device_ids=[ctx.device.index], bucket_cap_mb=bucket_cap_mb,
                                       gradient_as_bucket_view=True, broadcast_buffers=False)
    return DistributedDataParallel(model, bucket_cap_mb=bucket_cap_mb, broadcast_buffers=False)


def fence(ctx):
    """barrier + device synchronisation on both sides of a timed region"""
    if ctx.device.type == "cuda":
        torch.cuda.synchronize()
    if ctx.distributed:
        dist.barrier()
    if ctx.device.type == "cuda":
        torch.cuda.synchronize()


def max_over_ranks(value, ctx):
    """the job's time for a region is the slowest rank's"""
    if not ctx.distributed:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=ctx.device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_objects(obj, ctx):
    """sharded inference (BASELINE config 5): per-frame results to rank 0, no tensor collective"""
    if not ctx.distributed:
        return [obj]
    out = [None] * ctx.world if ctx.rank == 0 else None
    dist.gather_object(obj, out, dst=0)
    return out
