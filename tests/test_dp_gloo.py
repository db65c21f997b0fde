"""not-gpu: the data-parallel path on 2 gloo ranks (CPU).  The flat-bucket all-reduce must give every
rank the MEAN of the per-shard gradients, which equals the single-process gradient of the mean loss
over the concatenated batch (what DataParallel's err.mean() at main.py:562 computes)."""
import os
import socket
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    import avsep_amd as P
    r, w, dev = P.dp.init_from_env(backend="gloo")
    assert (r, w, dev.type) == (rank, world, "cpu")
    torch.manual_seed(0)                                  # identical replicas
    lin = torch.nn.Linear(6, 3)
    flat_p = torch.cat([p.data.reshape(-1) for p in lin.parameters()])
    P.dp.broadcast_params_(flat_p)
    g = torch.Generator().manual_seed(1)
    X, Y = torch.randn(8, 6, generator=g), torch.randn(8, 3, generator=g)
    lo, hi = P.dp.shard_range(8, rank, world)
    loss = ((lin(X[lo:hi]) - Y[lo:hi]) ** 2).mean()
    loss.backward()
    flat = torch.cat([p.grad.reshape(-1) for p in lin.parameters()])
    P.dp.allreduce_mean_(flat, world)
    scal = P.dp.reduce_scalars([loss], world)
    torch.save({"flat": flat, "loss": scal}, out + f".{rank}")
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gradient_mean(tmp_path):
    port, out = _free_port(), str(tmp_path / "r")
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    res = [torch.load(out + f".{r}") for r in range(2)]
    assert torch.equal(res[0]["flat"], res[1]["flat"])
    torch.manual_seed(0)
    lin = torch.nn.Linear(6, 3)
    g = torch.Generator().manual_seed(1)
    X, Y = torch.randn(8, 6, generator=g), torch.randn(8, 3, generator=g)
    loss = ((lin(X) - Y) ** 2).mean()
    loss.backward()
    ref = torch.cat([p.grad.reshape(-1) for p in lin.parameters()])
    assert torch.allclose(res[0]["flat"], ref, atol=1e-6)
    assert abs(res[0]["loss"].item() - loss.item()) < 1e-6


def test_shard_range_errors():
    sys.path.insert(0, ROOT)
    import avsep_amd as P
    assert P.dp.shard_range(256, 3, 8) == (96, 128)
    try:
        P.dp.shard_range(10, 0, 4)
        assert False
    except ValueError:
        pass
