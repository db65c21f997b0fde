"""not-gpu: the CPU oracle replayed against the golden vectors that oracle/gen_golden.py captured
from the reference itself (the reference is absent at test time)."""
import argparse

import numpy as np
import torch
import pytest
import torch.nn.functional as F

from conftest import assert_close
from oracle import nets as O, criterion as OC, step as OS, sopp as OSP


def _args(**kw):
    a = argparse.Namespace(num_mix=2, log_freq=1, weighted_loss=1, binary_mask=1, output_activation="sigmoid",
                           img_activation="relu", not_pool_vis=False, fusion_type="hidsep", match_weight=0.1,
                           lr_sound=1e-3, lr_frame=1e-4, fix_vis=False, beta1=0.9, weight_decay=1e-4)
    for k, v in kw.items():
        setattr(a, k, v)
    return a


def test_prepare(golden):
    G = golden("prepare")
    for tag, kw in [("bin_w", dict(binary_mask=1, weighted_loss=1, log_freq=1)),
                    ("ratio_now", dict(binary_mask=0, weighted_loss=0, log_freq=1)),
                    ("nolog", dict(binary_mask=1, weighted_loss=1, log_freq=0))]:
        out = OS.prepare({"mag_mix": G["mag_mix"].clone(), "mags": [G["mags0"].clone(), G["mags1"].clone()]}, _args(**kw))
        mags, mix, logm, gt, w = out
        assert_close(mix, G[f"{tag}.mag_mix"], 1e-6)
        assert_close(logm, G[f"{tag}.log_mag_mix"], 1e-6)
        assert_close(w, G[f"{tag}.weights"], 1e-6)
        for n in range(2):
            assert_close(mags[n], G[f"{tag}.mags{n}"], 1e-6)
            assert_close(gt[n], G[f"{tag}.gt_masks{n}"], 1e-6)
    assert torch.equal(torch.from_numpy(OS.warpgrid(1, 8, 5, True)), G["warpgrid_8x5"])
    assert torch.equal(torch.from_numpy(OS.warpgrid(1, 8, 5, False)), G["unwarpgrid_8x5"])
    # edge quirk: an all-ones input is halved on the border rows/cols
    ones = torch.ones(1, 1, 512, 16)
    _, mix, *_ = OS.prepare({"mag_mix": ones - 1e-10, "mags": [ones.clone(), ones.clone()]}, _args())
    assert abs(mix[0, 0, 0, 0].item() - 0.25) < 1e-5 and abs(mix[0, 0, 100, 0].item() - 0.5) < 1e-5


def test_fusion(golden):
    G = golden("fusion")
    for ftype in ("hidsep", "CoLoc_Sel", "MixVis"):
        for att in ("cos", "sig"):
            tag = f"{ftype}.{att}"
            x = G[f"{tag}.x"].clone().requires_grad_(True)
            nv = 1 if ftype == "MixVis" else 2
            vs = [G[f"{tag}.v{i}"].clone().requires_grad_(True) for i in range(nv)]
            y, (ml, maps) = O.Fusion(ftype, att)(x, vs)
            ((y * G[f"{tag}.cot"]).sum() + 0.7 * ml.sum() + 0.01 * (maps ** 2).sum()).backward()
            assert_close(y, G[f"{tag}.y"], 2e-5, tag)
            assert_close(ml.reshape(-1), G[f"{tag}.match"], 2e-5, tag)
            assert_close(maps, G[f"{tag}.maps"], 2e-5, tag)
            assert_close(x.grad, G[f"{tag}.dx"], 2e-5, tag)
            for i in range(nv):
                assert_close(vs[i].grad, G[f"{tag}.dv{i}"], 2e-5, tag)
    for seed in (0, 1, 5):
        assert torch.equal(O.ao_swap(G["ao.x"], G[f"ao.draws{seed}"]), G[f"ao.y{seed}"])
    assert torch.equal(O.ao_swap(G["ao.x"], torch.zeros(4, dtype=torch.bool)), G["ao.y_allzero"])


def test_n_source_generalisation_reduces_to_the_reference(golden):
    """BASELINE.json configs[4] (3 sources) is beyond the reference (C = 2 is hard-coded, fusion_net.py:35,43-46); the
    build-defined generalisation (oracle Fusion._coloc_n / ao_permute_n / per-target PIT weights) must BE the reference's
    CoLoc when C = 2: checked against the reference-generated goldens, values and gradients."""
    G = golden("fusion")
    for att in ("cos", "sig"):
        tag = f"hidsep.{att}"
        x = G[f"{tag}.x"].clone().requires_grad_(True)
        vs = [G[f"{tag}.v{i}"].clone().requires_grad_(True) for i in range(2)]
        y, (ml, maps) = O.Fusion("hidsep", att)._coloc_n(x, vs)
        ((y * G[f"{tag}.cot"]).sum() + 0.7 * ml.sum() + 0.01 * (maps ** 2).sum()).backward()
        assert_close(y, G[f"{tag}.y"], 2e-5, tag)
        assert_close(ml.reshape(-1), G[f"{tag}.match"], 2e-5, tag)
        assert_close(maps, G[f"{tag}.maps"], 2e-5, tag)
        assert_close(x.grad, G[f"{tag}.dx"], 2e-5, tag)
        for i in range(2):
            assert_close(vs[i].grad, G[f"{tag}.dv{i}"], 2e-5, tag)
    # three sources: shapes, the remainder rule (D = 32 -> Dc = 10, channels 30..31 of the tile are zero) and the
    # defining properties: the winning permutation's score is the largest, a permutation of the visual inputs permutes
    # the attended vectors and leaves the match loss unchanged
    gen = torch.Generator().manual_seed(3)
    x = torch.randn(4, 32, 2, 2, generator=gen)
    vs = [torch.rand(4, 10, 3, 5, generator=gen) for _ in range(3)]
    fu = O.Fusion("hidsep", "sig")
    y, (ml, maps) = fu(x, vs)
    assert y.shape == (4, 64, 2, 2) and maps.shape == (4, 3, 3, 5)
    assert torch.equal(y[:, 30:32], torch.zeros(4, 2, 2, 2)) and torch.equal(y[:, 32:], x)
    y2, (ml2, maps2) = fu(x, [vs[2], vs[0], vs[1]])
    assert_close(ml2.reshape(1), ml.reshape(1), 1e-6)
    assert_close(y2[:, 0:10], y[:, 20:30], 1e-6); assert_close(y2[:, 10:20], y[:, 0:10], 1e-6)
    fu.num_src = 3
    draws = torch.tensor([0, 3, 5, 1])
    ya = O.ao_permute_n(x, draws, 3)
    g = torch.amax(x, dim=(2, 3))
    import itertools
    table = list(itertools.permutations(range(3)))
    for b in range(4):
        for c in range(3):
            blk = table[int(draws[b])][c]
            assert torch.equal(ya[b, 10 * c:10 * c + 10, 0, 0], g[b, 10 * blk:10 * blk + 10])
    assert torch.equal(ya[:, 30:32], torch.zeros(4, 2, 2, 2))


def test_unet(golden):
    G = golden("unet")
    for tag, downs, ngf, ftype, att in [("u5", 5, 8, "hidsep", "sig"), ("u6sel", 6, 4, "CoLoc_Sel", "sig")]:
        net = O.Unet(fc_dim=2, num_downs=downs, ngf=ngf, fusion_type=ftype, att_type=att)
        sd = {k[len(tag) + 3:]: v for k, v in G.items() if k.startswith(tag + ".w.")}
        assert list(net.state_dict().keys()) == list(sd.keys())
        net.load_state_dict(sd)
        for k, b in net.named_buffers():
            b.copy_(torch.ones_like(b) if k.endswith("running_var") else torch.zeros_like(b))
        vs = [G[f"{tag}.v{i}"].clone().requires_grad_(True) for i in range(2)]
        net.train()
        y, (ml, maps) = net(G[f"{tag}.x"], vs)
        ((y * G[f"{tag}.cot"]).sum() + 0.3 * ml).backward()
        full = tag == "u5"
        assert_close(y if full else y[:, :, ::8, ::8], G[f"{tag}.y"], 2e-5, tag)
        assert_close(ml.reshape(1), G[f"{tag}.match"], 2e-5)
        for k, p in net.named_parameters():
            assert_close(p.grad, G[f"{tag}.g.{k}"], 5e-4, k)
        for k, b in net.named_buffers():
            assert_close(b.double(), G[f"{tag}.b.{k}"].double(), 1e-5, k)


def test_criterion_and_pit(golden):
    G = golden("criterion")
    preds, tg, w = [G["p0"], G["p1"]], [G["t0"], G["t1"]], G["w"]
    for kind in ("bce", "l1", "l2"):
        c = OC.build_criterion(kind)
        assert_close(c(preds, tg, w).reshape(1), G[f"{kind}.list"], 1e-6)
        assert_close(c(preds[0], tg[0]).reshape(1), G[f"{kind}.tensor_now"], 1e-6)
    pit = OC.build_criterion("l1", use_pit=True)     # arch is ignored with use_pit (models/__init__.py:130-131)
    assert pit.kind == "bce"
    loss, perms = pit(G["pit.P"], G["pit.T"], G["pit.W"])
    assert_close(loss, G["pit.loss"], 1e-6)
    assert [tuple(p) for p in perms] == [tuple(p) for p in G["pit.perms"].tolist()]
    assert perms[2] == (0, 1)                                  # exact tie keeps the first permutation
    assert torch.equal(pit.reorder_tensor(G["pit.P"], perms), G["pit.reordered"])
    assert_close(pit.loss_matrix(G["pit.P"], G["pit.T"], G["pit.W"]), G["pit.mat"], 1e-6)


def test_synthesizer(golden):
    G = golden("synthesizer")
    for name, mod in (("innerprod", O.InnerProd(8)), ("bias", O.Bias())):
        mod.load_state_dict({k[len(name) + 3:]: v for k, v in G.items() if k.startswith(name + ".w.")})
        for fn, arg in (("forward", G["fi"]), ("forward_nosum", G["fi"]), ("forward_pixelwise", G["fim"])):
            assert_close(getattr(mod, fn)(arg, G["fs"]), G[f"{name}.{fn}"], 1e-5, f"{name}.{fn}")


def test_sopp(golden):
    G = golden("sopp")
    aud, sep = [G["aud0"], G["aud1"]], [G["sep0"], G["sep1"]]
    for cname in ("AttModel", "MatchAtt"):
        for at in ("cos", "sig"):
            tag, m = f"{cname}.{at}", OSP.AttModule(cname, at)
            assert_close(m(aud, None, None)[0], G[tag + ".ao.ctx"], 1e-6)
            ctx, (ml, maps) = m(aud, G["mix"], None)
            assert_close(ctx, G[tag + ".infer.ctx"], 1e-5)
            assert_close(ml, G[tag + ".infer.match"], 1e-5)
            assert_close(maps, G[tag + ".infer.maps"], 1e-5)
            ctx, meta = m(aud, G["mix"], sep)
            assert_close(ctx, G[tag + ".train.ctx"], 1e-5)
            for i, t in enumerate(meta):
                assert_close(t, G[tag + f".train.meta{i}"], 1e-5)
    net = O.Unet(fc_dim=6, num_downs=5, ngf=4, extra_size=6)
    net.load_state_dict({k[7:]: v for k, v in G.items() if k.startswith("unet.w.")})
    net.train()
    y, (extra,) = net(G["unet.x"])
    assert_close(y, G["unet.basis"], 2e-5)
    assert_close(extra, G["unet.extra"], 2e-5)


def test_train_steps(golden):
    """AV, AO, AV train steps (main.py:557-569 semantics incl. SGD skipping None grads)."""
    G = golden("step")
    seed = int(G["seed"][0])
    torch.manual_seed(seed)
    gen = torch.Generator().manual_seed(seed)
    snd = O.Unet(fc_dim=2, num_downs=5, ngf=8, fusion_type="hidsep", att_type="sig")
    O.wide_init(snd, gen)
    trunk = O.resnet18_trunk()
    torch.nn.Linear(512, 10)                       # the golden run consumed the RNG for a discarded fc here
    fc = torch.nn.Conv2d(512, 32, 3, padding=1)
    frm = O.VisualNet(fc_dim=32, pool_type="maxpool", dilate_scale=16)
    frm.features.load_state_dict(trunk.state_dict())
    frm.fc.load_state_dict(fc.state_dict())
    args = _args(log_freq=0)
    wrap = OS.NetWrapper((snd, frm), OC.build_criterion("bce", True), OC.build_criterion("bce"))
    opt = OS.create_optimizer((snd, frm), args)
    for it, use_vis in enumerate([True, False, True]):
        snd.levels()[-1].fusion.ao_draws = G[f"it{it}.draws"]
        batch = {"mag_mix": G["mag_mix"].clone(), "mags": [G["mags0"].clone(), G["mags1"].clone()],
                 "frames": [G["frames0"], G["frames1"]]}
        err, match, outs = OS.train_step(wrap, batch, opt, use_vis, args)
        assert abs(err - G[f"it{it}.err"].item()) < 2e-5
        if use_vis:
            assert abs(match - G[f"it{it}.match"].item()) < 2e-5
    assert_close(snd.state_dict()["unet_block.up_forward.2.weight"], G["final.sound.last_w"], 2e-4)
    assert_close(frm.state_dict()["fc.bias"], G["final.frame.fc_b"], 2e-4)


def test_reference_mixvis_step_raises_as_shipped(golden):
    """main.py:181 hands PitWrapper the un-stacked [B,1,F,T] weight: the reference's own forward_avmiximg, executed by
    oracle/gen_golden.py through the reference's NetWrapper.forward, raises (recorded type + message).  The oracle's
    forward_avmiximg (oracle/step.py) therefore carries a build-defined repair — forward_ao's per-target weight stacking
    (main.py:103) — and with the ORIGINAL un-stacked weight the oracle's PitWrapper fails the same way."""
    G = golden("step")
    assert bool(G["mixvis.raised"][0])
    typ = bytes(G["mixvis.exc_type"].numpy()).decode()
    msg = bytes(G["mixvis.exc_msg"].numpy()).decode()
    assert typ == "RuntimeError" and "must match the size of tensor" in msg, (typ, msg)
    from oracle import criterion as OC
    crit = OC.build_criterion("bce", True)
    B, Fq, T = 2, 64, 64
    g = torch.Generator().manual_seed(0)
    pred, gt = torch.rand(B, Fq, T, 2, generator=g), (torch.rand(B, Fq, T, 2, generator=g) > 0.5).float()
    weight = torch.rand(B, 1, Fq, T, generator=g)
    with pytest.raises(RuntimeError, match="must match the size of tensor"):
        crit(pred, gt, weight)                                   # what main.py:181 does
    err, perms = crit(pred, gt, torch.stack([weight[:, 0]] * 2, -1))   # the repair
    assert err.shape[0] == B and torch.isfinite(err).all()
