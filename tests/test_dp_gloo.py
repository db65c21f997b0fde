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


def _reporting_worker(rank, world, port, out):
    """Two gloo ranks driving FlatSGD the way the real networks do: ParamGrads nodes that PLACE gradients (grad_dest /
    scratch_dest / finish -> node_finished) plus one parameter handed back to autograd, with arm_early_reduce(1 | 2)."""
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    import avsep_amd as P
    from avsep_amd.net_wrapper import FlatSGD, attach_grad_sink
    from avsep_amd.models.audio_net import ParamGrads
    P.dp.init_from_env(backend="gloo")
    torch.manual_seed(0)
    snd = torch.nn.Sequential(torch.nn.Conv2d(1, 4, 3), torch.nn.BatchNorm2d(4), torch.nn.Conv2d(4, 2, 1))
    vis = torch.nn.Conv2d(3, 2, 1)
    sp, vp = list(snd.parameters()), list(vis.parameters())
    opt = FlatSGD([{"params": sp, "lr": 1e-3, "name": "sound"}, {"params": vp, "lr": 1e-3, "name": "frame_fc"}],
                  world_size=world, overlap=True, require_gpu=False)
    attach_grad_sink(opt, snd, vis)
    assert opt._reports
    gen = torch.Generator().manual_seed(100 + rank)
    res = []

    class Node(torch.autograd.Function):
        """One autograd node owning `params`: backward places every gradient but `returned` (handed back to autograd)."""
        @staticmethod
        def forward(ctx, net, group, grads, returned, x, *params):
            ctx.args = (net, group, grads, returned, params)
            return x.sum() * 0.0 + sum((p * 0).sum() for p in params)

        @staticmethod
        def backward(ctx, dy):
            net, group, grads, returned, params = ctx.args
            pg = ParamGrads(net)
            early_before = opt.early_reductions
            for i, p in enumerate(params):
                if i in returned:
                    pg.d[p] = grads[i].clone()             # handed back: AccumulateGrad + the post-accumulate hook follow
                else:
                    pg.add(p, grads[i].clone())
            out = pg.finish(list(params), group)
            assert opt.early_reductions == early_before or group == "sound"
            return (None, None, None, None, None, *out)
    for nodes in (1, 2):
        opt.zero_grad()
        x = torch.zeros(1, requires_grad=True)
        per_node = [[torch.randn(p.shape, generator=gen) for p in sp] for _ in range(nodes)]
        gv = [torch.randn(p.shape, generator=gen) for p in vp]
        loss = sum(Node.apply(snd, "sound", per_node[k], {1} if k == 0 else set(), x, *sp) for k in range(nodes))
        loss = loss + Node.apply(vis, "frame_fc", gv, set(), x, *vp)
        opt.arm_early_reduce(nodes)
        before = opt.early_reductions
        loss.backward()
        started_early = opt.early_reductions - before
        active, scale = opt.reduce_gradients(None)
        local = [sum(per_node[k][i] for k in range(nodes)) for i in range(len(sp))] + gv
        res.append({"early": started_early, "local": torch.cat([t.reshape(-1) for t in local]),
                    "reduced": torch.cat([(g_ * scale).reshape(-1) for _, g_ in opt._views])})
    torch.save(res, out + f".{rank}")
    dist.barrier()
    dist.destroy_process_group()


def test_early_allreduce_with_reporting_nodes_over_two_gloo_ranks(tmp_path):
    """FlatSGD.node_finished / _pending / _nodes_left with REPORTING nodes (attach_grad_sink: the networks' ParamGrads
    place gradients themselves and report): with one and with two U-Net nodes per step, one parameter of the first node
    handed back to autograd, the early all-reduce of the first group starts exactly once per step — after the LAST node
    and after the handed-back gradient was accumulated — and every rank's flat gradient equals the mean over the ranks."""
    sys.path.insert(0, ROOT)
    port, out = _free_port(), str(tmp_path / "rep")
    mp.spawn(_reporting_worker, args=(2, port, out), nprocs=2, join=True)
    res = [torch.load(out + f".{r}") for r in range(2)]
    for step in range(2):
        mean = (res[0][step]["local"] + res[1][step]["local"]) / 2
        for r in range(2):
            assert res[r][step]["early"] == 1, (step, r, res[r][step]["early"])
            assert torch.allclose(res[r][step]["reduced"], mean, atol=1e-6), (step, r)
