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


def _toy_nets():
    """Three parameter groups shaped like create_optimizer's (sound / frame_fc / frame_features), CPU tensors."""
    torch.manual_seed(0)
    sound = torch.nn.Sequential(torch.nn.Conv2d(1, 4, 3, padding=1), torch.nn.Conv2d(4, 2, 3, padding=1))
    fc = torch.nn.Conv2d(3, 2, 3, padding=1)
    feats = torch.nn.Sequential(torch.nn.Conv2d(3, 3, 3, padding=1), torch.nn.Conv2d(3, 3, 1))
    return sound, fc, feats


def _toy_loss(nets, X, V, Y, use_vis):
    sound, fc, feats = nets
    out = sound(X)
    if use_vis:
        out = out * fc(feats(V)).mean(dim=(2, 3), keepdim=True)
    return ((out - Y) ** 2).mean()


def _flat_worker(rank, world, port, out, overlap):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    import avsep_amd as P
    from avsep_amd.net_wrapper import FlatSGD
    P.dp.init_from_env(backend="gloo")
    nets = _toy_nets()
    opt = FlatSGD([{"params": list(nets[0].parameters()), "lr": 1e-3, "name": "sound"},
                   {"params": list(nets[1].parameters()), "lr": 1e-3, "name": "frame_fc"},
                   {"params": list(nets[2].parameters()), "lr": 1e-4, "name": "frame_features"}],
                  world_size=world, overlap=bool(overlap), require_gpu=False)
    g = torch.Generator().manual_seed(1)
    X, V, Y = torch.randn(8, 1, 6, 6, generator=g), torch.randn(8, 3, 6, 6, generator=g), torch.randn(8, 2, 6, 6, generator=g)
    lo, hi = P.dp.shard_range(8, rank, world)
    res = []
    for use_vis in (True, False, True):
        opt.zero_grad()
        loss = _toy_loss(nets, X[lo:hi], V[lo:hi], Y[lo:hi], use_vis)
        opt.arm_early_reduce(1)
        loss.backward()
        active, scale = opt.reduce_gradients(None if use_vis else ("sound",))
        # per-parameter views of the flat buffer (every tensor starts on a 256-byte boundary: FlatSGD.ALIGN)
        res.append({"grad": torch.cat([(gv * scale).reshape(-1) for _, gv in opt._views]),
                    "ranges": [list(g_["range"]) for g_ in active]})
    torch.save({"res": res, "early": opt.early_reductions}, out + f".{rank}")
    dist.barrier()
    dist.destroy_process_group()


def test_flat_sgd_buckets_over_two_gloo_ranks(tmp_path):
    """FlatSGD's data-parallel half (net_wrapper.py: reduce_gradients, the early all-reduce of the first group armed by
    arm_early_reduce, the `only=` range of an audio-only step) on 2 gloo ranks: every rank ends up with the MEAN of the
    per-shard gradients = the single-process gradient of the concatenated batch, for AV, AO and AV steps, with and without
    the overlapped early all-reduce; the groups outside an audio-only step's range are left untouched (zero)."""
    sys.path.insert(0, ROOT)
    for overlap in (0, 1):
        port, out = _free_port(), str(tmp_path / f"f{overlap}")
        mp.spawn(_flat_worker, args=(2, port, out, overlap), nprocs=2, join=True)
        res = [torch.load(out + f".{r}") for r in range(2)]
        assert res[0]["early"] == (3 if overlap else 0)
        nets = _toy_nets()
        params = [p for n in nets for p in n.parameters()]
        g = torch.Generator().manual_seed(1)
        X, V, Y = torch.randn(8, 1, 6, 6, generator=g), torch.randn(8, 3, 6, 6, generator=g), torch.randn(8, 2, 6, 6, generator=g)
        import avsep_amd as P
        A = P.net_wrapper.FlatSGD.ALIGN
        n_sound = sum((p.numel() + A - 1) // A * A for p in nets[0].parameters())
        for step, use_vis in enumerate((True, False, True)):
            for p in params:
                p.grad = None
            _toy_loss(nets, X, V, Y, use_vis).backward()
            ref = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in params])
            for r in range(2):
                got = res[r]["res"][step]
                assert got["ranges"] == ([[0, n_sound]] if not use_vis else got["ranges"]) and got["ranges"][0][0] == 0
                assert len(got["ranges"]) == (3 if use_vis else 1)
                assert torch.allclose(got["grad"], ref, atol=1e-6), (overlap, step, r)


def test_shard_range_errors():
    sys.path.insert(0, ROOT)
    import avsep_amd as P
    assert P.dp.shard_range(256, 3, 8) == (96, 128)
    try:
        P.dp.shard_range(10, 0, 4)
        assert False
    except ValueError:
        pass
