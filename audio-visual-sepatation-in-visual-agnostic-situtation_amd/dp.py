"""Data parallelism: one process per GPU, identical replicas, ONE RCCL all-reduce of the flat
gradient buffer per step over xGMI (replaces nn.DataParallel at main.py:660-662, which broadcasts
all parameters, gathers outputs and reduces gradients to GPU 0 every step).  BatchNorm statistics
stay rank-local, which is what DataParallel's per-replica BN computes too.
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None, force=False):
    """Reads RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* (torchrun contract).  Returns (rank, world, device).
    `force` (or AVSEP_DP_FORCE=1): create the process group even at world size 1 (a 1-rank RCCL group)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    use_cuda = torch.cuda.is_available()
    # one process per GPU; on a box with fewer GPUs than ranks (rehearsals of the N > 1 path on a 1-GPU box, together
    # with AVSEP_DP_BACKEND=gloo: RCCL refuses two ranks on one device) the ranks wrap around the visible devices
    if use_cuda:
        local %= torch.cuda.device_count()
    backend = backend or os.environ.get("AVSEP_DP_BACKEND") or None
    device = torch.device("cuda", local) if use_cuda else torch.device("cpu")
    if use_cuda:
        torch.cuda.set_device(device)
    force = force or os.environ.get("AVSEP_DP_FORCE", "0") == "1"
    if (world > 1 or force) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group(backend or ("nccl" if use_cuda else "gloo"), rank=rank, world_size=world)
    return rank, world, device


def shard_range(global_batch, rank, world):
    """Rank r owns samples [r*B, (r+1)*B) of the global batch (main.py:772: num_gpus x batch_size_per_gpu)."""
    if global_batch % world:
        raise ValueError(f"global batch {global_batch} is not divisible by world size {world}")
    per = global_batch // world
    return rank * per, (rank + 1) * per


def allreduce_mean_(flat, world, group=None):
    """In-place mean over ranks of one flat buffer (sum over RCCL/gloo, then scale)."""
    if world > 1:
        dist.all_reduce(flat, group=group)
        flat.mul_(1.0 / world)
    return flat


def broadcast_params_(flat_param, src=0, group=None):
    """Make replicas identical once at start-up (the reference re-broadcasts every step)."""
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.broadcast(flat_param, src=src, group=group)
    return flat_param


def reduce_scalars(values, world, group=None):
    """Average a few logging scalars (err, match_loss) across ranks: the only other traffic."""
    t = torch.stack([v.detach().float().reshape(()) for v in values])
    if world > 1:
        dist.all_reduce(t, group=group)
        t = t / world
    return t
