"""-m gpu: the HIP U-Net / fusion / train step against the golden vectors generated from the
reference (tests/golden/*.npz) and against the CPU oracle on seeded inputs.
Tolerances (stated per assert) are fp32 summation-order noise through 10-14 conv layers with
train-mode BatchNorm; the north-star bound is mask MSE <= 1e-4."""
import argparse
import math

import pytest
import torch

from conftest import assert_close, rel_err

pytestmark = pytest.mark.gpu


def _pkg():
    import avsep_amd
    return avsep_amd


def _load_unet(P, G, tag, downs, ngf, ftype, att, dev, **kw):
    net = P.models.Unet(fc_dim=2, num_downs=downs, ngf=ngf, fusion_type=ftype, att_type=att, **kw)
    sd = {k[len(tag) + 3:]: v for k, v in G.items() if k.startswith(tag + ".w.")}
    assert list(net.state_dict().keys()) == list(sd.keys()), "state_dict keys/order differ from the reference"
    net.load_state_dict(sd)
    # the golden state_dict was captured after the reference's train passes: restart the BN buffers
    for k, b in net.named_buffers():
        b.copy_(torch.ones_like(b) if k.endswith("running_var") else torch.zeros_like(b))
    return net.to(dev)


@pytest.mark.parametrize("tag,downs,ngf,ftype,att,fuse", [
    ("u5", 5, 8, "hidsep", "sig", True), ("u5", 5, 8, "hidsep", "sig", False),
    ("u7", 7, 4, "hidsep", "cos", True), ("u6sel", 6, 4, "CoLoc_Sel", "sig", True)])
def test_unet_golden(dev, golden, tag, downs, ngf, ftype, att, fuse):
    P = _pkg()
    G = golden("unet")
    net = _load_unet(P, G, tag, downs, ngf, ftype, att, dev, fuse_upsample=fuse)
    x = G[f"{tag}.x"].to(dev)
    vs = [G[f"{tag}.v{i}"].to(dev).requires_grad_(True) for i in range(2)]
    cot = G[f"{tag}.cot"].to(dev)
    net.train()
    y, (ml, maps) = net(x, vs)
    ((y * cot).sum() + 0.3 * ml).backward()
    full = tag == "u5"
    yy = y if full else y[:, :, ::8, ::8]
    assert_close(yy, G[f"{tag}.y"], 2e-4, "logits")
    assert abs(y.double().sum().item() - G[f"{tag}.y_sum"].item()) <= 2e-4 * G[f"{tag}.y_abs"].item()
    assert_close(ml.reshape(1), G[f"{tag}.match"], 2e-4, "match loss")
    assert_close(maps, G[f"{tag}.maps"], 2e-4, "att maps")
    for i in range(2):
        assert_close(vs[i].grad, G[f"{tag}.dv{i}"], 2e-3, f"dv{i}")
    for k, p in net.named_parameters():
        assert p.grad is not None, k
        assert_close(p.grad, G[f"{tag}.g.{k}"], 3e-3, "grad " + k)
    # BatchNorm running statistics after exactly one train-mode forward
    for k, b in net.named_buffers():
        ref = G[f"{tag}.b.{k}"]
        if ref.dtype.is_floating_point:
            assert_close(b, ref, 1e-4, "buffer " + k)
        else:
            assert int(b) == int(ref), k
    net.ao_draws = G[f"{tag}.draws"]
    yao, meta = net(x, None)
    assert meta == (None, None)
    assert_close(yao if full else yao[:, :, ::8, ::8], G[f"{tag}.yao"], 2e-4, "AO logits")
    net.eval()
    with torch.no_grad():
        yev, _ = net(x, [v.detach() for v in vs])
    assert_close(yev if full else yev[:, :, ::8, ::8], G[f"{tag}.yev"], 3e-4, "eval logits")


def test_fusion_golden(dev, golden):
    P = _pkg()
    G = golden("fusion")
    from avsep_amd.models import fusion_net as FN
    for ftype in ("hidsep", "CoLoc_Sel", "MixVis"):
        for att in ("cos", "sig"):
            tag = f"{ftype}.{att}"
            mod = FN.get_fusion_net(ftype)(att_type=att)
            x = G[f"{tag}.x"].to(dev).requires_grad_(True)
            vs = [G[f"{tag}.v{i}"].to(dev).requires_grad_(True) for i in range(1 if ftype == "MixVis" else 2)]
            y, (ml, maps) = mod(x, vs)
            # golden objective also had a 0.01*sum(maps^2) term; the att_maps output is
            # non-differentiable on the HIP path (it is a visualisation output in the reference's
            # training loop), so compare forward values and the (y, match) gradients only
            ((y * G[f"{tag}.cot"].to(dev)).sum() + 0.7 * ml.sum()).backward()
            assert_close(y, G[f"{tag}.y"], 1e-5, tag + " y")
            assert_close(ml.reshape(1), G[f"{tag}.match"], 1e-5, tag + " match")
            assert_close(maps, G[f"{tag}.maps"], 1e-5, tag + " maps")
    x = G["ao.x"].to(dev)
    mod = FN.CoLoc(att_type="cos")
    for seed in (0, 1, 5):
        mod.ao_draws = G[f"ao.draws{seed}"]
        y, _ = mod(x, None)
        assert torch.equal(y.cpu(), G[f"ao.y{seed}"])
    mod.ao_draws = torch.zeros(4, dtype=torch.bool)
    assert torch.equal(mod(x, None)[0].cpu(), G["ao.y_allzero"])


def test_fusion_grad_vs_oracle(dev):
    """gradients of (y, match) against the CPU oracle's autograd, both attention kernels."""
    from oracle import nets as O
    from avsep_amd.models import fusion_net as FN
    g = torch.Generator().manual_seed(31)
    B, D, H, W = 3, 64, 5, 4
    for ftype in ("hidsep", "CoLoc_Sel", "MixVis"):
        for att in ("cos", "sig"):
            x = torch.randn(B, D, 2, 2, generator=g)
            vs = [torch.randn(B, D // 2, H, W * (2 if ftype == "MixVis" else 1), generator=g).relu()
                  for _ in range(1 if ftype == "MixVis" else 2)]
            cot = torch.randn(B, 2 * D, 2, 2, generator=g)
            xo, vo = x.clone().requires_grad_(True), [v.clone().requires_grad_(True) for v in vs]
            yo, (mlo, _) = O.Fusion(ftype, att)(xo, vo)
            ((yo * cot).sum() + 0.7 * mlo.sum()).backward()
            xd = x.to(dev).requires_grad_(True)
            vd = [v.to(dev).requires_grad_(True) for v in vs]
            y, (ml, _) = FN.get_fusion_net(ftype)(att_type=att)(xd, vd)
            ((y * cot.to(dev)).sum() + 0.7 * ml.sum()).backward()
            assert_close(xd.grad, xo.grad, 2e-5, f"{ftype}.{att} dx")
            for a, b in zip(vd, vo):
                assert_close(a.grad, b.grad, 2e-5, f"{ftype}.{att} dv")


@pytest.mark.parametrize("att", ["cos", "sig"])
@pytest.mark.parametrize("C,D", [(2, 64), (3, 64), (3, 512), (4, 66)])
def test_fusion_n_kernel(dev, C, D, att):
    """csrc/fusion_n.hip (CoLoc for C = 2..4 sources, BASELINE configs[4]): values and gradients of (y, match) and of the
    audio-only permutation branch against the CPU oracle's autograd (oracle/nets.py:Fusion._coloc_n / ao_permute_n; D not
    a multiple of C exercises the zero remainder channels); with C = 2 the generalised kernel must equal the two-source
    kernel of csrc/fusion.hip (kind 0), which is pinned by the reference goldens."""
    import math
    from oracle import nets as O
    P = _pkg()
    FN = P.models.fusion_net
    g = torch.Generator().manual_seed(100 * C + D)
    B, H, W = 3, 5, 4
    Dc = D // C
    x = torch.randn(B, D, 2, 2, generator=g)
    vs = [torch.randn(B, Dc, H, W, generator=g).relu() for _ in range(C)]
    cot = torch.randn(B, 2 * D, 2, 2, generator=g)
    mod = FN.get_fusion_net("hidsep")(att_type=att)
    mod.num_src = C
    xd = x.to(dev).requires_grad_(True)
    vd = [v.to(dev).requires_grad_(True) for v in vs]
    fus = mod._run_forward_n(xd.detach(), [v.detach() for v in vd], None)          # the kernel, whatever C is
    if C == 2:
        two = mod.run_forward(xd.detach(), [v.detach() for v in vd], None)         # csrc/fusion.hip
        for k in ("feat", "att_maps", "match_part", "best", "sel_idx", "pool_idx"):
            assert torch.equal(fus[k], two[k]), k
        dfeat = torch.randn(B, D, generator=g).to(dev)
        dm = torch.tensor(0.7, device=dev)
        dxa, dxb = torch.zeros_like(xd), torch.zeros_like(xd)
        da = mod._run_backward_n(xd.detach(), [v.detach() for v in vd], fus, dfeat, dxa, dm)
        db = mod.run_backward(xd.detach(), [v.detach() for v in vd], two, dfeat, dxb, None, dm)
        assert_close(dxa, dxb, 1e-6, "dx C=2")
        for a, b in zip(da, db):
            assert_close(a, b, 1e-6, "dv C=2")
        return
    xo, vo = x.clone().requires_grad_(True), [v.clone().requires_grad_(True) for v in vs]
    yo, (mlo, atto) = O.Fusion("hidsep", att)(xo, vo)
    ((yo * cot).sum() + 0.7 * mlo.sum()).backward()
    y, (ml, attm) = mod(xd, vd)
    ((y * cot.to(dev)).sum() + 0.7 * ml.sum()).backward()
    assert_close(y, yo, 1e-5, "y"); assert_close(attm, atto, 1e-5, "att maps")
    assert abs(ml.item() - mlo.item()) < 1e-5
    assert_close(xd.grad, xo.grad, 2e-5, "dx")
    for a, b in zip(vd, vo):
        assert_close(a.grad, b.grad, 2e-5, "dv")
    # audio-only: every permutation index once
    draws = torch.arange(B) * 2 % math.factorial(C)
    ofus = O.Fusion("hidsep", att)
    ofus.num_src, ofus.ao_draws, mod.ao_draws = C, draws, draws
    xo2, xd2 = x.clone().requires_grad_(True), x.to(dev).requires_grad_(True)
    yo2, _ = ofus(xo2, None)
    (yo2 * cot).sum().backward()
    y2, _ = mod(xd2, None)
    (y2 * cot.to(dev)).sum().backward()
    assert torch.equal(y2.cpu(), yo2.detach()), "AO tiles"
    assert_close(xd2.grad, xo2.grad, 1e-6, "AO dx")


def test_synthesizer_entry_points_golden(dev, golden):
    """InnerProd / Bias (models/synthesizer_net.py:6-70): forward (GEMV kernel), forward_nosum and forward_pixelwise (the
    inference helpers: avsep_innerprod_nosum / avsep_innerprod_pixelwise, the [P x K] x [K x HW] mask contraction on the
    f32 MFMA) against the fixtures generated from the reference; under autograd the helpers must give the same values
    and differentiate; a full-size pixelwise case (K = 32, 14x14 visual positions, 256x256 audio positions) against
    torch.bmm on the CPU."""
    P = _pkg()
    G = golden("synthesizer")
    for name, mod in (("innerprod", P.models.synthesizer_net.InnerProd(8)), ("bias", P.models.synthesizer_net.Bias())):
        mod.load_state_dict({k[len(name) + 3:]: v for k, v in G.items() if k.startswith(name + ".w.")})
        mod = mod.to(dev)
        for fn, arg in (("forward", G["fi"]), ("forward_nosum", G["fi"]), ("forward_pixelwise", G["fim"])):
            with torch.no_grad():
                assert_close(getattr(mod, fn)(arg.to(dev), G["fs"].to(dev)), G[f"{name}.{fn}"], 1e-5, f"{name}.{fn} (kernel)")
            a, snd = arg.to(dev).requires_grad_(True), G["fs"].to(dev).requires_grad_(True)
            out = getattr(mod, fn)(a, snd)
            assert_close(out, G[f"{name}.{fn}"], 1e-5, f"{name}.{fn} (autograd)")
            out.sum().backward()
            assert a.grad is not None and snd.grad is not None
    g = torch.Generator().manual_seed(4)
    mod = P.models.synthesizer_net.InnerProd(32)
    with torch.no_grad():
        mod.scale.copy_(torch.rand(32, generator=g) + 0.5)
        mod.bias.fill_(0.3)
    imgs, snd = torch.randn(2, 32, 14, 14, generator=g), torch.randn(2, 32, 256, 256, generator=g)
    with torch.no_grad():
        ref = torch.bmm((imgs.view(2, 32, 196).transpose(1, 2) * mod.scale), snd.view(2, 32, -1)).view(2, 14, 14, 256, 256) + 0.3
        out = mod.to(dev).forward_pixelwise(imgs.to(dev), snd.to(dev))
    assert_close(out, ref, 2e-6, "forward_pixelwise, full size")


def _args(**kw):
    a = argparse.Namespace(num_mix=2, log_freq=0, weighted_loss=1, binary_mask=1, output_activation="sigmoid",
                           img_activation="relu", not_pool_vis=False, fusion_type="hidsep", match_weight=0.1,
                           lr_sound=1e-3, lr_frame=1e-4, fix_vis=False, beta1=0.9, weight_decay=1e-4,
                           stft_frame=1022, stft_hop=256)
    for k, v in kw.items():
        setattr(a, k, v)
    return a


def test_train_steps_golden(dev, golden):
    """Three train_steps (AV, AO, AV) of the reference's main.py on small nets: losses and the
    parameters after SGD must match the golden run."""
    P = _pkg()
    from oracle import nets as O
    G = golden("step")
    seed = int(G["seed"][0])
    torch.manual_seed(seed)
    gen = torch.Generator().manual_seed(seed)
    # same construction order / RNG consumption as oracle/gen_golden.py:build_small_nets
    osnd = O.Unet(fc_dim=2, num_downs=5, ngf=8, fusion_type="hidsep", att_type="sig")
    O.wide_init(osnd, gen)
    # the golden run built the trunk, then a (discarded) Linear(512,10), then the fc conv: same RNG order
    trunk = O.resnet18_trunk()
    torch.nn.Linear(512, 10)
    fc = torch.nn.Conv2d(512, 32, 3, padding=1)
    ofrm = O.VisualNet(fc_dim=32, pool_type="maxpool", dilate_scale=16)
    ofrm.features.load_state_dict(trunk.state_dict())
    ofrm.fc.load_state_dict(fc.state_dict())
    snd = P.models.Unet(fc_dim=2, num_downs=5, ngf=8, fusion_type="hidsep", att_type="sig")
    frm = P.models.ResnetDilated(None, fc_dim=32, pool_type="maxpool")
    snd.load_state_dict(osnd.state_dict())
    frm.load_state_dict(ofrm.state_dict())
    snd, frm = snd.to(dev), frm.to(dev)
    args = _args()
    mb = P.ModelBuilder()
    wrap = P.NetWrapper((snd, frm), mb.build_criterion("bce", use_pit=True), mb.build_criterion("bce"))
    opt = P.create_optimizer((snd, frm), args)

    def batch():
        return {"mag_mix": G["mag_mix"].to(dev), "mags": [G["mags0"].to(dev), G["mags1"].to(dev)],
                "frames": [G["frames0"].to(dev), G["frames1"].to(dev)]}
    for it, use_vis in enumerate([True, False, True]):
        snd.ao_draws = G[f"it{it}.draws"]
        err, match, outs = P.net_wrapper.train_step_async(wrap, batch(), opt, use_vis, args)
        assert abs(err.item() - G[f"it{it}.err"].item()) <= 2e-4 * max(1.0, abs(G[f"it{it}.err"].item())), it
        if use_vis:
            assert abs(match.item() - G[f"it{it}.match"].item()) <= 2e-4
        for n in range(2):
            mse = ((outs["pred_masks"][n].detach().cpu() - G[f"it{it}.pred{n}"]) ** 2).mean().item()
            assert mse <= 1e-6, f"it{it} mask MSE {mse}"     # north-star bound is 1e-4
    for pre, net in (("sound", snd), ("frame", frm)):
        for k, p in net.state_dict().items():
            if p.dtype.is_floating_point and "running" not in k:
                ref_abs = G[f"final.{pre}.{k}.abs"].item()
                assert abs(p.double().sum().item() - G[f"final.{pre}.{k}.sum"].item()) <= 1e-3 * max(ref_abs, 1e-6), k
    assert_close(snd.state_dict()["unet_block.up_forward.2.weight"], G["final.sound.last_w"], 1e-3, "last conv after 3 steps")
    assert_close(frm.state_dict()["fc.bias"], G["final.frame.fc_b"], 1e-3, "fc bias after 3 steps")


def test_full_size_properties(dev):
    """BASELINE configs[1] tile size (256x256, unet7, 64 ngf) at small batch: size-independent
    properties — AV output invariant to swapping BOTH the visual order and nothing else changes the
    permutation-symmetric quantities; AO pass is deterministic given the draw; finite gradients."""
    P = _pkg()
    mb = P.ModelBuilder()
    torch.manual_seed(0)
    net = mb.build_sound(arch="unet7", fc_dim=2, fusion_type="hidsep", att_type="sig").to(dev)
    B = 2
    x = torch.randn(B, 1, 256, 256, device=dev) * 2 - 4
    vs = [torch.rand(B, 256, 14, 14, device=dev) for _ in range(2)]
    net.train()
    y1, (m1, a1) = net(x, vs)
    y2, (m2, a2) = net(x, vs[::-1])
    assert y1.shape == (B, 2, 256, 256) and a1.shape == (B, 2, 14, 14)
    # the match loss is symmetric under swapping the two visual streams (both permutations are scored)
    assert abs(m1.item() - m2.item()) < 1e-6
    y1.sum().backward()
    for k, p in net.named_parameters():
        assert p.grad is not None and torch.isfinite(p.grad).all(), k
    net.ao_draws = torch.tensor([True, False])
    ya, _ = net(x, None)
    yb, _ = net(x, None)
    assert torch.equal(ya, yb)


def test_mixvis_step_vs_oracle(dev):
    """forward_avmiximg (main.py:162-192): one train step on small nets against the CPU oracle."""
    P = _pkg()
    from oracle import nets as O, step as OS, criterion as OC
    torch.manual_seed(5)
    gen = torch.Generator().manual_seed(5)
    osnd = O.Unet(fc_dim=2, num_downs=5, ngf=8, fusion_type="MixVis", att_type="sig")
    O.wide_init(osnd, gen)
    ofrm = O.VisualNet(fc_dim=32, pool_type="maxpool", dilate_scale=16)
    snd = P.models.Unet(fc_dim=2, num_downs=5, ngf=8, fusion_type="MixVis", att_type="sig")
    frm = P.models.ResnetDilated(None, fc_dim=32, pool_type="maxpool")
    snd.load_state_dict(osnd.state_dict())
    frm.load_state_dict(ofrm.state_dict())
    snd, frm = snd.to(dev), frm.to(dev)
    args = _args(fusion_type="MixVis")
    srcs = [torch.rand(2, 1, 64, 64, generator=gen) ** 2 for _ in range(2)]
    frames = [torch.randn(2, 3, 2, 64, 64, generator=gen) for _ in range(2)]

    def batch(d):
        return {"mag_mix": (srcs[0] + srcs[1]).to(d), "mags": [s.clone().to(d) for s in srcs], "frames": [f.to(d) for f in frames]}
    mb = P.ModelBuilder()
    wrap = P.NetWrapper((snd, frm), mb.build_criterion("bce", use_pit=True), mb.build_criterion("bce"))
    opt = P.create_optimizer((snd, frm), args)
    owrap = OS.NetWrapper((osnd, ofrm), OC.build_criterion("bce", True), OC.build_criterion("bce"))
    oopt = OS.create_optimizer((osnd, ofrm), args)
    for it in range(2):
        err, match, outs = P.net_wrapper.train_step_async(wrap, batch(dev), opt, True, args)
        oerr, omatch, oouts = OS.train_step(owrap, batch("cpu"), oopt, True, args)
        assert abs(err.item() - oerr) < 2e-4, (it, err.item(), oerr)
        assert abs(match.item() - omatch) < 2e-4
        for n in range(2):
            assert ((outs["pred_masks"][n].detach().cpu() - oouts["pred_masks"][n].detach()) ** 2).mean().item() < 1e-6
        assert_close(outs["maps"], oouts["maps"], 2e-4, "maps")


def test_sopp_variant_vs_oracle(dev, golden):
    """SoP++ operators (A17-A20): basis U-Net (golden from the reference's SoP++/audio_net.py), InnerProd
    fwd/bwd, attention module, and the four stage forwards + backward against the CPU oracle."""
    P = _pkg()
    from oracle import nets as O, sopp as OSP, criterion as OC
    from avsep_amd import sopp as PS
    G = golden("sopp")
    net = P.models.Unet(fc_dim=6, num_downs=5, ngf=4, extra_size=6)
    sd = {k[7:]: v for k, v in G.items() if k.startswith("unet.w.")}
    assert list(net.state_dict().keys()) == list(sd.keys())
    net.load_state_dict(sd)
    net = net.to(dev).train()
    y, (extra,) = net(G["unet.x"].to(dev))
    assert_close(y, G["unet.basis"], 2e-4, "sopp basis")
    assert_close(extra, G["unet.extra"], 2e-4, "sopp extra")
    # attention modules against the reference's goldens
    aud, sep = [G["aud0"].to(dev), G["aud1"].to(dev)], [G["sep0"].to(dev), G["sep1"].to(dev)]
    for cname, cls in (("AttModel", P.models.AttModel), ("MatchAtt", P.models.MatchAtt)):
        for at in ("cos", "sig"):
            m, tag = cls(att_type=at), f"{cname}.{at}"
            assert_close(m(aud, None, None)[0], G[tag + ".ao.ctx"], 1e-5)
            ctx, meta = m(aud, G["mix"].to(dev), sep)
            assert_close(ctx, G[tag + ".train.ctx"], 1e-5)
            for i, t in enumerate(meta):
                assert_close(t, G[tag + f".train.meta{i}"], 1e-5)
    # stage math on small nets, product vs oracle (same weights, same inputs)
    torch.manual_seed(9)
    gen = torch.Generator().manual_seed(9)
    K = 8
    osnd = O.Unet(fc_dim=K, num_downs=5, ngf=8, extra_size=K)
    O.wide_init(osnd, gen)
    ofrm = O.VisualNet(fc_dim=K, pool_type="maxpool", dilate_scale=16)
    osyn, opit = O.InnerProd(K), OSP.AttModule("AttModel", "sig")
    with torch.no_grad():
        osyn.scale.copy_(torch.rand(K, generator=gen) + 0.5)
    snd = P.models.Unet(fc_dim=K, num_downs=5, ngf=8, extra_size=K)
    frm = P.models.ResnetDilated(None, fc_dim=K, pool_type="maxpool")
    syn = P.ModelBuilder().build_synthesizer("linear", fc_dim=K)
    snd.load_state_dict(osnd.state_dict()); frm.load_state_dict(ofrm.state_dict()); syn.load_state_dict(osyn.state_dict())
    snd, frm, syn = snd.to(dev), frm.to(dev), syn.to(dev)
    args = _args(sound_activation="no", fusion_type="Base", att_type="sig")
    pit = P.models.get_attmodule(args)(att_type="sig")
    mb = P.ModelBuilder()
    wrap = PS.NetWrapper((snd, frm, syn, pit), mb.build_criterion("bce", use_pit=True), mb.build_criterion("bce"))
    owrap = OSP.SopNetWrapper((osnd, ofrm, osyn, opit), OC.build_criterion("bce", True), OC.build_criterion("bce"))
    srcs = [torch.rand(2, 1, 64, 64, generator=gen) ** 2 for _ in range(2)]
    frames = [torch.randn(2, 3, 2, 64, 64, generator=gen) for _ in range(2)]

    def batch(d):
        return {"mag_mix": (srcs[0] + srcs[1]).to(d), "mags": [s.clone().to(d) for s in srcs], "frames": [f.to(d) for f in frames]}
    for use_vis, stage in ((True, 1), (True, 2), (True, 3), (False, 0)):
        for m in (snd, frm, syn, osnd, ofrm, osyn):
            m.zero_grad()
        wrap.train(); owrap.train()
        err, outs = wrap(batch(dev), args, use_vis, stage)
        oerr, oouts = owrap(batch("cpu"), args, use_vis, stage)
        err.mean().backward(); oerr.mean().backward()
        assert abs(err.mean().item() - oerr.mean().item()) < 2e-4, (stage, err, oerr)
        for n in range(2):
            assert ((outs["pred_masks"][n].detach().cpu() - oouts["pred_masks"][n].detach()) ** 2).mean().item() < 1e-6
        assert_close(syn.scale.grad, osyn.scale.grad, 3e-3, f"stage {stage} dscale")
        assert_close(snd.unet_block.up_forward.at(2).weight.grad, osnd.unet_block.up_conv.weight.grad, 3e-3, f"stage {stage} last conv grad")
        k = "unet_block.mid_forward.mid_forward.mid_forward.mid_forward.down_forward.1.weight"
        assert_close(dict(snd.named_parameters())[k].grad, dict(osnd.named_parameters())[k].grad, 3e-3, f"stage {stage} bottleneck conv grad")


def test_sopp_three_stage_schedule_vs_oracle(dev):
    """SoP++/main.py:670-688 + :593-606: train_step_3stage picks the AV stage from the iteration number (stage 1 below
    train_steps[0], 2 below train_steps[1], 3 up to train_steps[2]) and steps the SoP++ optimizer groups; six
    iterations (AV on the even ones, as the shipped flags schedule them) against the oracle's stage forwards driven by
    the same rule + torch.optim.SGD over the reference's groups: every loss, then the parameters."""
    P = _pkg()
    from avsep_amd import sopp as PS
    from oracle import nets as O, criterion as OC, sopp as OSP
    torch.manual_seed(19)
    gen = torch.Generator().manual_seed(19)
    K = 8
    osnd = O.Unet(fc_dim=K, num_downs=5, ngf=8, extra_size=K)
    O.wide_init(osnd, gen)
    ofrm = O.VisualNet(fc_dim=K, pool_type="maxpool", dilate_scale=16)
    osyn, opit = O.InnerProd(K), OSP.AttModule("AttModel", "sig")
    snd = P.models.Unet(fc_dim=K, num_downs=5, ngf=8, extra_size=K)
    frm = P.models.ResnetDilated(None, fc_dim=K, pool_type="maxpool")
    syn = P.ModelBuilder().build_synthesizer("linear", fc_dim=K)
    snd.load_state_dict(osnd.state_dict()); frm.load_state_dict(ofrm.state_dict()); syn.load_state_dict(osyn.state_dict())
    snd, frm, syn = snd.to(dev), frm.to(dev), syn.to(dev)
    args = _args(sound_activation="no", fusion_type="Base", att_type="sig", lr_synthesizer=1e-3, train_steps=[2, 4, 6])
    pit = P.models.get_attmodule(args)(att_type="sig").to(dev)
    mb = P.ModelBuilder()
    wrap = PS.NetWrapper((snd, frm, syn, pit), mb.build_criterion("bce", use_pit=True), mb.build_criterion("bce"))
    owrap = OSP.SopNetWrapper((osnd, ofrm, osyn, opit), OC.build_criterion("bce", True), OC.build_criterion("bce"))
    opt = PS.create_optimizer((snd, frm, syn, pit), args)
    groups = [{"params": osnd.parameters(), "lr": args.lr_sound}, {"params": osyn.parameters(), "lr": args.lr_synthesizer}]
    if list(opit.parameters()):
        groups.append({"params": opit.parameters(), "lr": args.lr_synthesizer})
    groups += [{"params": ofrm.features.parameters(), "lr": args.lr_frame}, {"params": ofrm.fc.parameters(), "lr": args.lr_sound}]
    oopt = torch.optim.SGD(groups, momentum=args.beta1, weight_decay=args.weight_decay)
    srcs = [torch.rand(2, 1, 64, 64, generator=gen) ** 2 for _ in range(2)]
    frames = [torch.randn(2, 3, 2, 64, 64, generator=gen) for _ in range(2)]

    def batch(d):
        return {"mag_mix": (srcs[0] + srcs[1]).to(d), "mags": [s.clone().to(d) for s in srcs], "frames": [f.to(d) for f in frames]}
    assert [PS.stage_of(i, args.train_steps) for i in range(7)] == [1, 1, 2, 2, 3, 3, 3]
    with pytest.raises(ValueError):
        PS.stage_of(7, args.train_steps)
    for i in range(7):
        use_vis = i % 2 == 0
        err, match = PS.train_step_3stage(wrap, batch(dev), opt, use_vis, i, args)
        owrap.train()
        oopt.zero_grad(set_to_none=True)
        oerr, oouts = owrap(batch("cpu"), args, use_vis, PS.stage_of(i, args.train_steps))
        oerr.mean().backward()
        oopt.step()
        assert abs(err - oerr.mean().item()) < 3e-4 * max(1.0, abs(err)), (i, err, oerr.mean().item())
        if use_vis:
            om = oouts.get("match_loss")                     # stage 1 has no matching term
            assert abs(match - (0.0 if om is None else om.mean().item())) < 3e-4
    osd = osnd.state_dict()
    for k, v in snd.state_dict().items():
        if v.dtype.is_floating_point and "running" not in k:
            assert_close(v, osd[k], 3e-3, "after 7 three-stage steps: " + k)
    assert_close(syn.scale, osyn.scale, 1e-3, "synthesizer scale")


def test_shared_encoder_pair_equals_two_passes(dev, golden):
    """forward_pair (one encoder, two decoders) == two forward() calls: outputs, every gradient, and the
    BatchNorm running statistics (two momentum updates)."""
    P = _pkg()
    G = golden("unet")
    x = G["u5.x"].to(dev)
    v = [G["u5.v0"].to(dev), G["u5.v1"].to(dev)]
    res = {}
    for mode in ("pair", "two"):
        net = _load_unet(P, G, "u5", 5, 8, "hidsep", "sig", dev).train()
        va = [t.clone().requires_grad_(True) for t in v]
        if mode == "pair":
            (fa, (ma, _)), (fb, (mb, _)) = net.forward_pair(x, va[::-1], va)
        else:
            fa, (ma, _) = net(x, va[::-1])
            fb, (mb, _) = net(x, va)
        ((fa * G["u5.cot"].to(dev)).sum() + 2 * (fb * G["u5.cot"].to(dev)).sum() + 0.3 * ma + 0.7 * mb).backward()
        res[mode] = dict(fa=fa.detach(), fb=fb.detach(), ma=ma.detach(), mb=mb.detach(), dv=[t.grad for t in va],
                         g={k: p.grad for k, p in net.named_parameters()}, b={k: b.clone() for k, b in net.named_buffers()})
    a, b = res["pair"], res["two"]
    assert torch.equal(a["fa"], b["fa"]) and torch.equal(a["fb"], b["fb"])
    assert_close(a["ma"], b["ma"], 1e-6); assert_close(a["mb"], b["mb"], 1e-6)
    for i in range(2):
        assert_close(a["dv"][i], b["dv"][i], 1e-5, "dv")
    for k in a["g"]:
        assert_close(a["g"][k], b["g"][k], 2e-5, "grad " + k)
    for k in a["b"]:
        assert_close(a["b"][k].double(), b["b"][k].double(), 1e-6, "buffer " + k)


@pytest.mark.parametrize("log_freq,backend", [(1, "hip"), (0, "hip")])
def test_config1_full_size_step_vs_oracle(dev, log_freq, backend):
    _full_size_step(dev, log_freq, backend, "f32", 1e-4)


def test_config2_bf16_full_hip_step_vs_oracle(dev):
    """BASELINE.json configs[2] arithmetic: bf16 conv operands (rounded while they are staged), fp32 accumulation, BatchNorm
    statistics, loss, master weights and SGD; visual trunk on this library ("full HIP path").  Same full-size AV + AO
    steps as configs[1], against the fp32 CPU oracle: the north-star bound  mask MSE <= 1e-4  must hold; the loss may
    differ by the bf16 operand rounding (bound 2e-3 of the loss, measured value printed)."""
    _full_size_step(dev, 1, "hip", "bf16", 2e-3)


@pytest.mark.parametrize("prec,err_tol", [("f32", 1e-4), ("bf16", 3e-3)])
def test_config5_three_sources_five_frames_step_vs_oracle(dev, prec, err_tol):
    """BASELINE.json configs[4]: 3-source mix, 512x256 STFT tiles (log_freq 0), 5 frames per source.  The reference
    hard-codes two sources (fusion_net.py:35,43-46; main.py:103,109); the N-source generalisation is build-defined
    (DESIGN.md §9), restated in the oracle (Fusion._coloc_n, ao_permute_n, per-target PIT weights) and pinned to the
    reference at N = 2 by tests/test_oracle_golden.py.  Here the HIP path (3 x 5 frames through the visual trunk, two
    U-Net passes with reversed / natural visual order, 3! PIT permutations on the audio-only step) meets that oracle
    at full tile size: loss, match loss, mask MSE <= 1e-4."""
    import numpy as np
    P = _pkg()
    from oracle import nets as O, step as OS, criterion as OC, stft as OST
    a = P.arguments.train_music_args()
    a.stft_pad_mode = "reflect"
    a.log_freq, a.num_mix, a.num_frames, a.num_channels = 0, 3, 5, 3
    a.vis_channels = 512 // 3                                    # remainder rule: 2 bottleneck channels stay unmatched
    raw = P.synth.make_batch(2, a.num_mix, a.num_frames, 224, a.audLen, seed=78)
    mags = [torch.from_numpy(np.stack([OST.stft_mag_phase(w.numpy())[0] for w in src]))[:, None] for src in raw["audios"]]
    mix = torch.from_numpy(np.stack([OST.stft_mag_phase(w.numpy())[0] for w in raw["audio_mix"]]))[:, None]
    torch.manual_seed(12)
    gen = torch.Generator().manual_seed(12)
    osnd = O.build_sound(a.arch_sound, a.num_channels, a.fusion_type, a.att_type)
    O.wide_init(osnd, gen)
    ofrm = O.build_frame(a.arch_frame, a.vis_channels, a.img_pool)
    mb = P.ModelBuilder()
    snd = mb.build_sound(arch=a.arch_sound, fc_dim=a.num_channels, fusion_type=a.fusion_type, att_type=a.att_type)
    frm = mb.build_frame(arch=a.arch_frame, fc_dim=a.vis_channels, pool_type=a.img_pool)
    assert [tuple(v.shape) for v in snd.state_dict().values()][:-2] == \
        [tuple(v.shape) for v in mb.build_sound(arch=a.arch_sound, fc_dim=2, fusion_type=a.fusion_type,
                                                 att_type=a.att_type).state_dict().values()][:-2], \
        "only the last conv (one logit per source) may depend on the number of sources"
    snd.load_state_dict(osnd.state_dict()); frm.load_state_dict(ofrm.state_dict())
    snd, frm = snd.to(dev), frm.to(dev)
    frm.backend = "hip"
    wrap = P.NetWrapper((snd, frm), mb.build_criterion(a.loss, use_pit=True), mb.build_criterion(a.loss))
    opt = P.create_optimizer((snd, frm), a)
    owrap = OS.NetWrapper((osnd, ofrm), OC.build_criterion(a.loss, True), OC.build_criterion(a.loss))
    oopt = OS.create_optimizer((osnd, ofrm), a)
    FN, launched = P.models.fusion_net, []
    orig_call = FN.call
    FN.call = lambda name, *args: (launched.append(name), orig_call(name, *args))[1]     # which fusion entry points run
    P.kernels.set_precision(prec)         # bf16: 512x256 (non-square) tiles and 15 frames per mixture on B16 images
    try:
        _config5_steps(P, OS, dev, a, raw, mix, mags, snd, osnd, wrap, owrap, opt, oopt, err_tol)
    finally:
        FN.call = orig_call
        P.kernels.set_precision("f32")
    # the N-source fusion is the HIP kernel pair of csrc/fusion_n.hip (two AV passes + one AO pass, forward and backward),
    # and the two AV passes share one encoder (one U-Net autograd node)
    assert launched.count("avsep_fusion_n_av_fwd") == 2 and launched.count("avsep_fusion_n_av_bwd") == 2, launched
    assert launched.count("avsep_fusion_n_ao_fwd") == 1 and launched.count("avsep_fusion_n_ao_bwd") == 1, launched
    assert not [n for n in launched if not n.startswith("avsep_fusion_n_")], launched


def _config5_steps(P, OS, dev, a, raw, mix, mags, snd, osnd, wrap, owrap, opt, oopt, err_tol=1e-4):
    for use_vis in (True, False):
        draws = torch.tensor([4, 1])                              # permutation indices (itertools order) of the AO step
        snd.ao_draws = draws
        osnd.levels()[-1].fusion.ao_draws = draws
        gb = {"audios": [w.to(dev) for w in raw["audios"]], "audio_mix": raw["audio_mix"].to(dev),
              "frames": [f.to(dev) for f in raw["frames"]]}
        cb = {"mag_mix": mix.clone(), "mags": [m.clone() for m in mags], "frames": raw["frames"]}
        err, match, outs = P.net_wrapper.train_step_async(wrap, gb, opt, use_vis, a)
        oerr, omatch, oouts = OS.train_step(owrap, cb, oopt, use_vis, a)
        assert len(outs["pred_masks"]) == 3 and outs["pred_masks"][0].shape == (2, 1, 512, 256)
        mse = max(((x.detach().cpu() - y.detach()) ** 2).mean().item()
                  for x, y in zip(outs["pred_masks"], oouts["pred_masks"]))
        print(f"configs[4] 3 sources {'AV' if use_vis else 'AO'}: err hip={err.item():.6f} oracle={oerr:.6f} mask-MSE={mse:.2e}")
        assert mse <= 1e-4, mse
        assert abs(err.item() - oerr) <= err_tol * max(1.0, abs(oerr)), (use_vis, err.item(), oerr)
        if use_vis:
            # the match term is a sum over 3! permutation scores of 3 attention-map maxima each (|.| ~ 2): bf16 bound 5e-3 of it
            assert abs(match.item() - omatch) <= (1e-4 if err_tol <= 1e-4 else 5e-3 * max(1.0, abs(omatch))), (match.item(), omatch)
        else:
            assert list(oouts["perms"]) == list(wrap._last_perms), "PIT must pick the same permutation of the 3 sources"
        if use_vis:
            assert wrap.unet_nodes == 1, "shared encoder for N = 3"


def _full_size_step(dev, log_freq, backend, prec, err_tol):
    """(log_freq 0 keeps the 512x256 tiles of configs[4]; backend "hip" runs the visual trunk on this library too.)
    BASELINE.json configs[0]/[1] shapes at batch 2: 65535-sample waveforms -> STFT 1022/256 -> 512x256 ->
    log-frequency warp to 256x256, 3 frames of 224x224 per source, unet7 (64 ngf) + hidsep(sig) + resnet18dilated,
    BCE, SGD.  One AV and one AO train step of the HIP path against the CPU oracle on identical inputs and weights
    (reference weight init for the U-Net, i.e. N(0, 1e-3) convs, plus a wide-init variant): loss, match loss and the
    north-star bound  mask MSE <= 1e-4  (measured ~1e-12)."""
    import numpy as np
    P = _pkg()
    from oracle import nets as O, step as OS, criterion as OC, stft as OST
    P.kernels.set_precision(prec)
    try:
        _full_size_step_body(dev, log_freq, backend, prec, err_tol, P, O, OS, OC, OST, np)
    finally:
        P.kernels.set_precision("f32")


def _full_size_step_body(dev, log_freq, backend, prec, err_tol, P, O, OS, OC, OST, np):
    a = P.arguments.train_music_args()
    a.stft_pad_mode = "reflect"
    a.log_freq = log_freq
    raw = P.synth.make_batch(2, a.num_mix, a.num_frames, 224, a.audLen, seed=77)
    mags = [torch.from_numpy(np.stack([OST.stft_mag_phase(w.numpy())[0] for w in src]))[:, None] for src in raw["audios"]]
    mix = torch.from_numpy(np.stack([OST.stft_mag_phase(w.numpy())[0] for w in raw["audio_mix"]]))[:, None]
    for wide in ((False, True) if log_freq else (True,)):
        torch.manual_seed(11)
        gen = torch.Generator().manual_seed(11)
        osnd = O.build_sound(a.arch_sound, a.num_channels, a.fusion_type, a.att_type)
        if wide:
            O.wide_init(osnd, gen)
        ofrm = O.build_frame(a.arch_frame, a.vis_channels, a.img_pool)
        mb = P.ModelBuilder()
        snd = mb.build_sound(arch=a.arch_sound, fc_dim=a.num_channels, fusion_type=a.fusion_type, att_type=a.att_type)
        frm = mb.build_frame(arch=a.arch_frame, fc_dim=a.vis_channels, pool_type=a.img_pool)
        snd.load_state_dict(osnd.state_dict()); frm.load_state_dict(ofrm.state_dict())
        snd, frm = snd.to(dev), frm.to(dev)
        assert frm.backend == backend == "hip"
        wrap = P.NetWrapper((snd, frm), mb.build_criterion(a.loss, use_pit=True), mb.build_criterion(a.loss))
        opt = P.create_optimizer((snd, frm), a)
        owrap = OS.NetWrapper((osnd, ofrm), OC.build_criterion(a.loss, True), OC.build_criterion(a.loss))
        oopt = OS.create_optimizer((osnd, ofrm), a)
        for use_vis in (True, False):
            draws = torch.tensor([True, False])
            snd.ao_draws = draws
            osnd.levels()[-1].fusion.ao_draws = draws
            gb = {"audios": [w.to(dev) for w in raw["audios"]], "audio_mix": raw["audio_mix"].to(dev),
                  "frames": [f.to(dev) for f in raw["frames"]]}          # the HIP path runs its own STFT
            cb = {"mag_mix": mix.clone(), "mags": [m.clone() for m in mags], "frames": raw["frames"]}
            err, match, outs = P.net_wrapper.train_step_async(wrap, gb, opt, use_vis, a)
            oerr, omatch, oouts = OS.train_step(owrap, cb, oopt, use_vis, a)
            mse = max(((x.detach().cpu() - y.detach()) ** 2).mean().item()
                      for x, y in zip(outs["pred_masks"], oouts["pred_masks"]))
            print(f"full-size {prec}/{backend} wide={wide} {'AV' if use_vis else 'AO'}: err hip={err.item():.6f} "
                  f"oracle={oerr:.6f} |d|={abs(err.item() - oerr):.2e} mask-MSE={mse:.2e}")
            assert mse <= 1e-4, f"mask MSE {mse} (wide={wide}, use_vis={use_vis})"
            assert abs(err.item() - oerr) <= err_tol * max(1.0, abs(oerr)), (wide, use_vis, err.item(), oerr)
            if use_vis:
                assert abs(match.item() - omatch) <= max(1e-4, err_tol)


# ---- the dispatch bench.py times (batch 64), under the full-size oracle parity test --------------------------------------
# Kernel family, tile shape, workgroup size and split-K of a convolution call depend on the size of its grid, i.e. on the
# batch: at batch 2 most layers of the step do NOT run the Winograd / 512-thread bf16 instantiations that make up the
# benched batch-64 step.  These tests run the full-size AV + AO step at batch 8 with every descriptor PLANNED for batch
# 64 (avsep_conv_desc.plan_n: decisions of the batch-64 call, grids of the batch-8 call), assert layer by layer that the
# launched variant is the variant of the real batch-64 descriptor, and compare with the CPU oracle at batch 8.
BENCH_BATCH, DISPATCH_TEST_BATCH = 64, 8


class _DispatchLog:
    """Records (geometry, mode, launched variant, variant of the same call at the bench batch) of every conv call."""

    def __init__(self, K, scale):
        self.K, self.scale, self.rows, self.orig = K, scale, [], {}
        for name in ("fwd", "dgrad", "wgrad", "dgrad_up2x"):
            self._wrap(name)

    def _wrap(self, name):
        orig, log = getattr(self.K.Conv, name), self
        self.orig[name] = orig

        def wrapped(cv, *a, **kw):
            mode = "dgrad" if name == "dgrad_up2x" else name
            with_stats = bool(len(a) > 2 and a[2] is not None) or kw.get("stats") is not None
            launched = cv.kernel_variant(mode, with_stats)
            bench = log.K.lib.ConvDesc.from_buffer_copy(cv.d)              # the descriptor bench.py builds for this layer
            bench.N, bench.plan_n = cv.N * log.scale, 0
            import ctypes
            buf = ctypes.create_string_buffer(128)
            assert log.K.lib.load().avsep_conv_kernel_variant(ctypes.byref(bench), {"fwd": 0, "dgrad": 1, "wgrad": 2}[mode],
                                                               int(with_stats), buf, 128) == 0
            geom = (cv.Cin, cv.H, cv.W, cv.Cout, cv.KH, cv.d.stride, cv.d.dil, cv.d.up2x)
            log.rows.append((geom, mode, launched, buf.value.decode()))
            return orig(cv, *a, **kw)
        setattr(self.K.Conv, name, wrapped)

    def close(self):
        for name, orig in self.orig.items():
            setattr(self.K.Conv, name, orig)


_ORACLE_CACHE = {}


def _oracle_full_size_b8(P, O, OS, OC, OST, np):
    """One AV then one AO oracle train step (CPU, fp32) at the dispatch-test batch; computed once for both precisions."""
    if "r" in _ORACLE_CACHE:
        return _ORACLE_CACHE["r"]
    a = P.arguments.train_music_args()
    a.stft_pad_mode = "reflect"
    raw = P.synth.make_batch(DISPATCH_TEST_BATCH, a.num_mix, a.num_frames, 224, a.audLen, seed=78)
    mags = [torch.from_numpy(np.stack([OST.stft_mag_phase(w.numpy())[0] for w in src]))[:, None] for src in raw["audios"]]
    mix = torch.from_numpy(np.stack([OST.stft_mag_phase(w.numpy())[0] for w in raw["audio_mix"]]))[:, None]
    torch.manual_seed(13)
    gen = torch.Generator().manual_seed(13)
    osnd = O.build_sound(a.arch_sound, a.num_channels, a.fusion_type, a.att_type)
    O.wide_init(osnd, gen)
    ofrm = O.build_frame(a.arch_frame, a.vis_channels, a.img_pool)
    init = ({k: v.clone() for k, v in osnd.state_dict().items()}, {k: v.clone() for k, v in ofrm.state_dict().items()})
    owrap = OS.NetWrapper((osnd, ofrm), OC.build_criterion(a.loss, True), OC.build_criterion(a.loss))
    oopt = OS.create_optimizer((osnd, ofrm), a)
    draws = torch.arange(DISPATCH_TEST_BATCH) % 3 == 0
    # the AV step once more in float64 from the same weights: the exact gradient both fp32 implementations approximate
    import copy
    s64, f64 = copy.deepcopy(osnd).double(), copy.deepcopy(ofrm).double()
    w64 = OS.NetWrapper((s64, f64), OC.build_criterion(a.loss, True), OC.build_criterion(a.loss))
    w64.train()
    # prepare() in fp32 (identical tiles, binary targets and weights for all three runs), everything after it in float64
    mags32, mix32, logmix32, gt32, wts32 = OS.prepare({"mag_mix": mix.clone(), "mags": [m.clone() for m in mags]}, a)
    data64 = ([m.double() for m in mags32], mix32.double(), logmix32.double(), [g.double() for g in gt32], wts32.double())
    err64, _ = w64.forward_av(data64, [f.double() for f in raw["frames"]], a)
    err64.mean().backward()
    grads64 = {("sound." + k): p.grad.detach().clone() for k, p in s64.named_parameters() if p.grad is not None}
    grads64.update({("frame." + k): p.grad.detach().clone() for k, p in f64.named_parameters() if p.grad is not None})
    del s64, f64, w64, data64
    steps = []
    for use_vis in (True, False):
        osnd.levels()[-1].fusion.ao_draws = draws
        cb = {"mag_mix": mix.clone(), "mags": [m.clone() for m in mags], "frames": raw["frames"]}
        oerr, omatch, oouts = OS.train_step(owrap, cb, oopt, use_vis, a)
        # the gradients this step's SGD update consumed (train_step zeroes them at the START of a step): main.py:557-569
        ograds = {("sound." + k): p.grad.detach().clone() for k, p in osnd.named_parameters() if p.grad is not None}
        ograds.update({("frame." + k): p.grad.detach().clone() for k, p in ofrm.named_parameters() if p.grad is not None})
        steps.append((oerr, omatch, [m.detach().clone() for m in oouts["pred_masks"]], ograds, grads64 if use_vis else None))
    _ORACLE_CACHE["mags"] = (mix, mags)
    _ORACLE_CACHE["r"] = (a, raw, init, draws, steps)
    return _ORACLE_CACHE["r"]


def _check_flat_grads(prec, nets, ograds, ograds64):
    """Every parameter gradient of the AV step (the views of FlatSGD.flat_grad the backward kernels wrote) against the CPU
    oracle.  A weight gradient is a cancelling sum over 10^5..10^7 products, so the fp32 CPU oracle itself is only good to
    1e-4..5e-3 of a tensor's norm (a ReLU / max-pool decision on a pre-activation of ~1e-7 flips between any two fp32
    implementations): the arbiter is the oracle in FLOAT64, run once from the same weights and inputs.  Per tensor, with
    e(x) = ||x - g64||_2 / ||g64||_2:
      fp32:  the distance of the reference's own fp32 arithmetic from the exact gradient is the yardstick, r = e(hip) / e(oracle
             fp32).  Attribution (tools/grad_attribution.py, profiles/r05_grad_attribution_*.txt: the same check with the
             Winograd kernels off, on one stream, from the HIP STFT): neither the Winograd transforms (the direct-form kernels
             are slightly WORSE: decoder 1.5-2x against 0.3-0.9x with F(2x2)) nor the order of the statistics atomics carry the
             5-19x a few tensors showed in round 4 — those were single ReLU / max-pool decisions flipping at the bottleneck (they
             moved to other tensors, or vanished, whenever any kernel's rounding changed; with the F(4x4,3x3) kernels the same
             tensors sit at 1.2-2.3x), i.e. one noise realisation, while the F(4x4,3x3) kernels raise the SMOOTH error of everything
             downstream of them to ~2x the oracle's own (median r 1.08 -> 1.75; trunk convs 1.1x -> 2.0x, decoder 0.3-0.9x ->
             1.6-1.9x).  Bounds: per parameter family (decoder / encoder / trunk convs and BatchNorms, fc) the MEDIAN r <= 8
             (measured 2.2-3.9 with the F(4x4) kernels, run-to-run +-1: the encoder inherits whatever decision flipped at the
             bottleneck) — a kernel family that is systematically off moves its family's median by orders of magnitude; and per
             tensor e(hip) <= max(25 * e(oracle fp32), 1e-3), which leaves room for one flipped decision (19x seen);
      bf16:  every operand and every stored activation / gradient carries a 2^-9 rounding and this network doubles a
             relative error per decoder level on the way back (the fp32 errors above grow the same way); the visual trunk only
             receives the gradient that went through the whole decoder and the fusion.  What arrives there is one noise
             realisation: two equally valid roundings of the same activations (the accumulation order of ONE trunk layer changed
             when its kernel was replaced) moved every trunk cosine from 0.80-0.87 to 0.47-0.58.  The bounds are therefore: the
             tensors at the head of the backward pass (outermost up conv, first BatchNorm below it) within 2e-2, the decoder
             (`up_forward`) cosine >= 0.85, every tensor cosine >= 0.3 and the median >= 0.5 — a wiring check; each kernel's
             arithmetic is pinned exactly by tests/test_gpu_ops.py (measured values are printed)."""
    rows, bad = [], []
    for prefix, net in nets:
        for k, p in net.named_parameters():
            og = ograds.get(prefix + k)
            if og is None:
                continue
            assert p.grad is not None, prefix + k
            g = p.grad.detach().double().cpu()
            ref = ograds64[prefix + k]
            e_hip = ((g - ref).norm() / ref.norm().clamp_min(1e-300)).item()
            e_o32 = ((og.double() - ref).norm() / ref.norm().clamp_min(1e-300)).item()
            cos = (g.flatten() @ ref.flatten() / (g.norm() * ref.norm()).clamp_min(1e-300)).item()
            rows.append((prefix + k, e_hip, e_o32, cos))
            if prec == "f32":
                ok = e_hip <= max(25.0 * e_o32, 1e-3)
            else:
                head = k in ("unet_block.up_forward.2.weight", "unet_block.up_forward.2.bias", "unet_block.mid_forward.up_forward.3.weight",
                             "unet_block.mid_forward.up_forward.3.bias") and prefix == "sound."
                ok = cos >= (0.85 if (prefix == "sound." and "up_forward" in k) else 0.3) and (not head or e_hip <= 2e-2)
            if not ok:
                bad.append(rows[-1])
    if prec == "f32":
        fam = {}
        for name, e_hip, e_o32, _ in rows:
            sound = name.startswith("sound.")
            conv = name.endswith("weight") and (".down_forward.0." in name or ".down_forward.1." in name or "up_forward.2." in name or
                                                ".conv" in name or name.endswith("features.0.weight") or ".fc." in name or "downsample.0" in name)
            key = ("U-Net decoder " if "up_forward" in name else "U-Net encoder " if sound else "trunk ") + ("conv" if conv else "BatchNorm")
            fam.setdefault(key, []).append(e_hip / max(e_o32, 1e-300))
        for key, v in sorted(fam.items()):
            v.sort()
            print(f"    family {key:24s} n={len(v):3d}  median e_hip/e_oracle32 {v[len(v) // 2]:.2f}  worst {v[-1]:.2f}")
            assert v[len(v) // 2] <= 8.0, (key, v)
    by_err = sorted(rows, key=lambda r: -r[1])
    med = sorted(r[1] for r in rows)[len(rows) // 2]
    med_cos = sorted(r[3] for r in rows)[len(rows) // 2]
    print(f"benched dispatch {prec}: {len(rows)} parameter gradients vs the float64 oracle: relative L2 error median {med:.2e}, "
          f"worst {by_err[0][1]:.2e} ({by_err[0][0]}); cosine median {med_cos:.4f}, lowest {min(r[3] for r in rows):.4f}")
    for name, e_hip, e_o32, cos in by_err[:10]:
        print(f"    {name:70s} hip {e_hip:.2e}  oracle-fp32 {e_o32:.2e}  cos {cos:.4f}")
    assert not bad, f"{len(bad)} gradients off: {bad[:6]}"
    if prec != "f32":
        assert med_cos >= 0.5, med_cos
    return len(rows)


@pytest.mark.parametrize("prec,err_tol", [("f32", 1e-4), ("bf16", 2e-3)])
def test_benched_dispatch_full_size_step_vs_oracle(dev, prec, err_tol):
    """Full-size (256x256 tiles, 3x224^2 frames, unet7 + resnet18dilated) AV + AO train step on the kernel instantiations
    bench.py times at batch 64, against the CPU oracle: mask MSE <= 1e-4 (north star), loss |d| <= 1e-4 (fp32) /
    2e-3 (bf16 operands), and per layer launched variant == variant of the batch-64 descriptor."""
    import numpy as np
    P = _pkg()
    from oracle import nets as O, step as OS, criterion as OC, stft as OST
    K = P.kernels
    a, raw, init, draws, osteps = _oracle_full_size_b8(P, O, OS, OC, OST, np)
    scale = BENCH_BATCH // DISPATCH_TEST_BATCH
    K.set_precision(prec)
    K.plan_batch_scale = scale
    log = _DispatchLog(K, scale)
    try:
        mb = P.ModelBuilder()
        snd = mb.build_sound(arch=a.arch_sound, fc_dim=a.num_channels, fusion_type=a.fusion_type, att_type=a.att_type)
        frm = mb.build_frame(arch=a.arch_frame, fc_dim=a.vis_channels, pool_type=a.img_pool)
        snd.load_state_dict(init[0]); frm.load_state_dict(init[1])
        snd, frm = snd.to(dev), frm.to(dev)
        wrap = P.NetWrapper((snd, frm), mb.build_criterion(a.loss, use_pit=True), mb.build_criterion(a.loss))
        opt = P.create_optimizer((snd, frm), a)
        for (oerr, omatch, omasks, ograds, ograds64), use_vis in zip(osteps, (True, False)):
            snd.ao_draws = draws
            gb = {"audios": [w.to(dev) for w in raw["audios"]], "audio_mix": raw["audio_mix"].to(dev),
                  "frames": [f.to(dev) for f in raw["frames"]]}
            err, match, outs = P.net_wrapper.train_step_async(wrap, gb, opt, use_vis, a)
            mse = max(((x.detach().cpu() - y) ** 2).mean().item() for x, y in zip(outs["pred_masks"], omasks))
            print(f"benched dispatch {prec} {'AV' if use_vis else 'AO'}: err hip={err.item():.6f} oracle={oerr:.6f} "
                  f"|d|={abs(err.item() - oerr):.2e} mask-MSE={mse:.2e}")
            assert mse <= 1e-4, f"mask MSE {mse} ({prec}, use_vis={use_vis})"
            assert abs(err.item() - oerr) <= err_tol * max(1.0, abs(oerr)), (prec, use_vis, err.item(), oerr)
            if use_vis:
                assert abs(match.item() - omatch) <= max(1e-4, err_tol)
        # ---- gradients: what the batch-64 instantiations of the data / weight gradient kernels write into the flat buffer (U-Net
        # d1-d7, u1-u7, BatchNorm gamma / beta, trunk layer1-4 and fc).  A fresh copy of the model runs the AV forward + backward on
        # the ORACLE's STFT magnitudes: through 14 BatchNorm + (Leaky)ReLU layers the gradient is so sensitive to its input that the
        # 1e-5 difference between the HIP STFT (fp32 DFT) and numpy's float64 FFT alone moves it by 2e-3 (measured), 10x the
        # distance between the fp32 and float64 oracles — that input noise must not mask a kernel's error
        mb = P.ModelBuilder()
        snd = mb.build_sound(arch=a.arch_sound, fc_dim=a.num_channels, fusion_type=a.fusion_type, att_type=a.att_type)
        frm = mb.build_frame(arch=a.arch_frame, fc_dim=a.vis_channels, pool_type=a.img_pool)
        snd.load_state_dict(init[0]); frm.load_state_dict(init[1])
        snd, frm = snd.to(dev), frm.to(dev)
        wrap = P.NetWrapper((snd, frm), mb.build_criterion(a.loss, use_pit=True), mb.build_criterion(a.loss))
        opt = P.create_optimizer((snd, frm), a)
        omix, omags = _ORACLE_CACHE["mags"]
        gb = {"mag_mix": omix.to(dev), "mags": [m.to(dev) for m in omags], "frames": [f.to(dev) for f in raw["frames"]]}
        wrap.train()
        opt.zero_grad()
        with K.pack_scope():
            err, _ = wrap.forward(gb, a, True)
            err.mean().backward()
        _, _, _, ograds, ograds64 = osteps[0]
        n = _check_flat_grads(prec, (("sound.", snd), ("frame.", frm)), ograds, ograds64)
        assert n >= 95, n
    finally:
        log.close()
        K.plan_batch_scale = 1
        K.set_precision("f32")
    assert len(log.rows) > 150, len(log.rows)          # AV: 2 decoder passes + 20 trunk convs, fwd + dgrad + wgrad; AO: one pass
    wrong = [r for r in log.rows if r[2] != r[3]]
    assert not wrong, "launched variant != variant of the batch-%d call: %s" % (BENCH_BATCH, wrong[:5])
    fams = {}
    for geom, mode, launched, _ in log.rows:
        fams.setdefault(launched.split(":")[0], set()).add((geom, mode))
    print("families:", {k: len(v) for k, v in fams.items()})
    # the families profiles/r02_layers_*.txt lists for the batch-64 step
    if prec == "f32":
        for must in ("wino4_kernel", "winow4_kernel", "wino_kernel", "wgrad4d_kernel", "conv3x3_kernel", "head_fwd_kernel", "head_dgrad_kernel",
                     "head_wgrad_kernel"):
            assert must in fams, (must, sorted(fams))
        # every 3x3/s1 conv with >= 64 input channels on an even map >= 8x8 is a Winograd launch, in all three modes
        for geom, mode, launched, _ in log.rows:
            cin, h, w, cout, k, s, dil, up = geom
            if k == 3 and s == 1 and not up and cin >= 64 and cout >= 64 and h >= 8 and h % 2 == 0:
                assert launched.split(":")[0] in ("wino4_kernel", "wino_kernel", "winow_kernel", "winow4_kernel"), (geom, mode, launched)
    else:
        for must in ("convbf_kernel", "wgradb_kernel"):
            assert must in fams, (must, sorted(fams))
        big = [r for r in log.rows if r[2].startswith("convbf_kernel") and r[2].endswith("x256")]
        assert len(big) >= 20, "the 512-thread (256-pixel) bf16 tiles must be on the tested path: %d" % len(big)


# (Cin, H, W, Cout, k, stride, pad, dil) of the step's convolutions that have bf16 backward kernels, at their true sizes
_BENCHED_BACKWARD_SHAPES = [
    (64, 128, 128, 128, 4, 2, 1, 1), (128, 64, 64, 256, 4, 2, 1, 1), (256, 32, 32, 512, 4, 2, 1, 1), (512, 16, 16, 512, 4, 2, 1, 1),
    (256, 128, 128, 64, 3, 1, 1, 1), (512, 64, 64, 128, 3, 1, 1, 1), (1024, 32, 32, 256, 3, 1, 1, 1), (1024, 16, 16, 512, 3, 1, 1, 1),
    (1024, 8, 8, 512, 3, 1, 1, 1),
    (64, 56, 56, 64, 3, 1, 1, 1), (64, 56, 56, 128, 3, 2, 1, 1), (128, 28, 28, 128, 3, 1, 1, 1), (128, 28, 28, 256, 3, 2, 1, 1),
    (256, 14, 14, 256, 3, 1, 1, 1), (256, 14, 14, 512, 3, 1, 1, 1), (512, 14, 14, 512, 3, 1, 2, 2), (512, 14, 14, 256, 3, 1, 1, 1),
    (64, 56, 56, 128, 1, 2, 0, 1), (256, 14, 14, 512, 1, 1, 0, 1),
]


@pytest.mark.parametrize("shape", _BENCHED_BACKWARD_SHAPES)
def test_bf16_backward_kernels_at_benched_shapes(dev, shape):
    """The check of bf16 mode's BACKWARD arithmetic that can fail for a wrong kernel (the whole-step gradient comparison of
    _check_flat_grads[bf16] is dominated by the chaos of a 14-level network: a wiring check).  Every convolution geometry of the
    step, at its true size and on the kernel instantiation of the batch-64 step (plan_n), gets operands that are EXACTLY
    representable in bf16 (so the bf16 kernels' operand rounding is the identity) and its bf16-mode data and weight gradients
    must equal the fp32 kernels' of this library (pinned against torch in tests/test_gpu_ops.py) up to fp32 accumulation order:
    2e-5 of the tensor's maximum, as test_conv_bf16_operands does at small shapes."""
    P = _pkg()
    K = P.kernels
    Cin, H, W, Cout, k, s, p, d = shape
    N = 8
    g = torch.Generator().manual_seed(sum(shape))
    r16 = lambda z: z.to(torch.bfloat16).float().to(dev)
    x = r16(torch.randn(N, Cin, H, W, generator=g))
    w = r16(torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5)
    K.plan_batch_scale = BENCH_BATCH // N
    try:
        cb = K.Conv(x, Cout, k, s, p, d, prec="bf16")
        cf = K.Conv(x, Cout, k, s, p, d, prec="f32")
        dy = r16(torch.randn(N, Cout, cb.Ho, cb.Wo, generator=g))
        fams = (cb.kernel_name("dgrad"), cb.kernel_name("wgrad"))
        assert fams[1] == "wgradb_kernel", fams                      # every listed geometry has a bf16 weight-gradient kernel
        dx16, dx32 = cb.dgrad(cb.pack(w, 1), dy), cf.dgrad(cf.pack(w, 1), dy)
        dw16, dw32 = cb.wgrad(dy)[0], cf.wgrad(dy)[0]
        e_dx = rel_err(K.to_f32(dx16), dx32)
        e_dw = rel_err(dw16, dw32)
        print(f"{shape}: {fams[0]} dgrad {e_dx:.2e}, {fams[1]} wgrad {e_dw:.2e}")
        # (the data gradient of a 1024-channel 3x3 layer sums 9216 products per output: 1.9e-5 measured, accumulation order only)
        assert e_dx <= 4e-5 and e_dw <= 2e-5, (shape, fams, e_dx, e_dw)
    finally:
        K.plan_batch_scale = 1


def test_sdr_on_synthetic_val_trains_in_both_precisions(dev):
    """bench.py's "SDR on synthetic val" leg (BASELINE.json's metric, second half) at a reduced length: the full-size model
    trained from the same seed in fp32 and in bf16 mode on the seeded synthetic stream with the shipped AV / audio-only schedule,
    evaluated on a held-out seeded set with the reference's evaluate() protocol (main.py:421-503).  The model must LEARN in both
    precisions — training loss down by >= 8 % (measured 13-28 %), validation SDR up by >= 4 dB from the untrained masks
    (measured +7 ... +20 dB; the thresholded masks of a 300-step model make the metric itself noisy) — and bf16 must track fp32:
    final training losses within 25 % (measured <= 10.6 %: a 15-step window of random batches), validation SDR within 9 dB (the
    SAME fp32 command has ended up to 6.2 dB apart between runs at this length — audio-only +3.9 ... +10.1 dB over seven runs,
    DESIGN.md 8d, profiles/r05_sdr_on_synthetic_val_*.json; a bf16 path that does not train shows up as no SDR gain (>= 14 dB
    short) and no loss decrease, far outside every one of these margins)."""
    import os
    import sys
    P = _pkg()
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    r = bench.sdr_on_synthetic_val(P, dev, 1234, steps=300, batch=8, val_batches=3, precisions=("f32", "bf16"))
    for prec in ("f32", "bf16"):
        x = r[prec]
        print(prec, {k: (round(x["before"][k]["sdr"], 2), round(x["after"][k]["sdr"], 2)) for k in ("val_av", "val_ao")},
              x["train_loss_av"], x["train_loss_ao"])
        for k in ("av", "ao"):
            t = x["train_loss_" + k]
            assert t["last"] <= 0.92 * t["first"], (prec, k, t)
        for k in ("val_av", "val_ao"):
            assert x["after"][k]["sdr"] >= x["before"][k]["sdr"] + 4.0, (prec, k, x["before"][k], x["after"][k])
            assert all(v == v for v in x["after"][k].values())
    for k in ("av", "ao"):
        a, b = r["f32"]["train_loss_" + k]["last"], r["bf16"]["train_loss_" + k]["last"]
        assert abs(a - b) <= 0.25 * a, (k, a, b)
    for k in ("val_av", "val_ao"):
        assert abs(r["bf16_minus_f32_after"][k]["sdr"]) <= 9.0, r["bf16_minus_f32_after"]


def test_eval_path_vs_oracle(dev):
    """N1: un-warp -> threshold -> mask*mag -> iSTFT -> SI-SDR / SDR on the GPU against the numpy/torch-CPU oracle."""
    import numpy as np
    import torch.nn.functional as F
    P = _pkg()
    from oracle import stft as OST, step as OS
    a = P.arguments.train_music_args()
    a.stft_pad_mode = "reflect"
    raw = P.synth.make_batch(2, 2, 1, 32, a.audLen, seed=5)
    gb = {"audios": [w.to(dev) for w in raw["audios"]], "audio_mix": raw["audio_mix"].to(dev)}
    mb = P.ModelBuilder()
    wrap = P.NetWrapper((mb.build_sound(arch="unet5", fc_dim=2, fusion_type="hidsep", att_type="sig"),
                         mb.build_frame(arch="resnet18dilated", fc_dim=256)), None, None)
    wrap.attach_stft(gb, a)
    gen = torch.Generator().manual_seed(3)
    masks = [torch.rand(2, 1, 256, 256, generator=gen) for _ in range(2)]
    out = P.evaluate.calc_metrics(gb, {"pred_masks": [m.to(dev) for m in masks]}, a)
    grid = torch.from_numpy(OS.warpgrid(2, 512, 256, warp=False))
    for n in range(2):
        lin = (F.grid_sample(masks[n], grid, align_corners=False) > a.mask_thres).float().numpy()
        for b in range(2):
            mag, ph = OST.stft_mag_phase(raw["audio_mix"][b].numpy())
            wav = OST.istft_reconstruction(mag * lin[b, 0], ph)
            ref = raw["audios"][n][b, :len(wav)].numpy()
            assert np.abs(out["pred_wavs"][n, b].cpu().numpy() - wav).max() < 2e-4
            assert abs(out["si_sdr"][b, n].item() - OST.si_sdr(wav, ref)) < 1e-2
            assert abs(out["sdr_plain"][b, n].item() - OST.sdr_plain(wav, ref)) < 1e-2
    # BSS-eval SDR / SIR / SAR (what the reference's get_metrics reports) on the reconstructed waveforms
    from oracle import bss_eval as OB
    L = out["pred_wavs"].shape[-1]
    for b in range(2):
        refs = np.stack([raw["audios"][n][b, :L].numpy() for n in range(2)])
        ests = np.stack([out["pred_wavs"][n, b].cpu().numpy() for n in range(2)])
        for key, ref in zip(("sdr", "sir", "sar"), OB.bss_eval_sources(refs, ests)):
            for n in range(2):
                assert abs(out[key][b, n].item() - ref[n]) < 0.05, (key, b, n, out[key][b, n].item(), ref[n])


@pytest.mark.parametrize("backend", ["hip"])
@pytest.mark.parametrize("B,T,HW,dilate,arch", [(2, 2, 64, 16, "resnet18dilated"), (1, 3, 224, 16, "resnet18dilated"),
                                                (2, 1, 96, 8, "resnet18dilated"), (2, 1, 64, 16, "resnet18fc")])
def test_visual_trunk_hip_backend(dev, B, T, HW, dilate, arch, backend):
    """The ResNet-18 (dilated) trunk + fc conv on the HIP kernels (models/vision_hip.py) against the oracle's
    VisualNet: features, every parameter gradient and the BatchNorm running statistics after one train-mode pass,
    then the eval-mode features.  224x224 frames reach the halo-patch 3x3 kernels; the small cases the generic path.
    The reference for the gradients is the oracle in float64: through 20 conv layers with tiny-batch BatchNorm a
    single ReLU / max-pool decision flip moves whole gradient tensors by per cents of their max (measured against
    float64 on the 224x224 case: this path <= 6.4e-3, the oracle itself in float32 on the CPU <= 4.7e-2), so the
    check is statistical (median and worst tensor), while everything before the first flip (fc) agrees to ~1e-5."""
    P = _pkg()
    import oracle as O
    import oracle.nets as ON
    torch.manual_seed(5)
    if arch == "resnet18dilated":
        onet = ON.VisualNet(fc_dim=16, pool_type="maxpool", dilate_scale=dilate)
        net = P.models.ResnetDilated(None, fc_dim=16, pool_type="maxpool", dilate_scale=dilate)
    else:
        onet = ON.VisualNet(fc_dim=16, pool_type="maxpool", dilate_scale=0)
        net = P.models.ResnetFC(None, fc_dim=16, pool_type="maxpool")
    assert list(net.state_dict().keys()) == list(onet.state_dict().keys())
    net.load_state_dict(onet.state_dict())
    net = net.to(dev)
    assert net.backend == backend == "hip"      # the package has no other trunk (MIOpen comparison: tools/miopen_compare)
    gen = torch.Generator().manual_seed(9)
    x = torch.randn(B, 3, T, HW, HW, generator=gen)
    import copy
    o64 = copy.deepcopy(onet).double()
    onet.train(); net.train(); o64.train()
    yo = onet.forward_multiframe(x, pool=False)
    cot = torch.randn(yo.shape, generator=gen)
    y64 = o64.forward_multiframe(x.double(), pool=False)
    (y64 * cot.double()).sum().backward()
    y = net.forward_multiframe(x.to(dev), pool=False)
    (y * cot.to(dev)).sum().backward()
    assert_close(y, yo, 2e-4, "features")
    og = dict(o64.named_parameters())
    errs = {}
    for k, p in net.named_parameters():
        assert p.grad is not None, k
        errs[k] = rel_err(p.grad, og[k].grad.float())
    # a ReLU decision on a pre-activation of ~1e-6 (they occur: min |pre| over a block is 1e-6..1e-4 here) is not
    # defined at float32 precision; on these tiny maps one such flip moves a whole gradient tensor by several per
    # cent.  The bulk must agree tightly, no tensor may be far off.
    worst = max(errs, key=errs.get)
    assert sorted(errs.values())[len(errs) // 2] <= 2e-2, sorted(errs.values())[len(errs) // 2]
    assert errs[worst] <= 0.15, (worst, errs[worst])
    assert_close(net.fc.weight.grad, og["fc.weight"].grad.float(), 1e-4, "grad fc.weight (before any flip)")
    ob = dict(onet.named_buffers())
    for k, b in net.named_buffers():
        if b.dtype.is_floating_point:
            assert_close(b, ob[k], 1e-4, "buffer " + k)
        else:
            assert int(b) == int(ob[k]), k
    onet.eval(); net.eval()
    with torch.no_grad():
        assert_close(net.forward_multiframe(x.to(dev), pool=True), onet.forward_multiframe(x, pool=True), 3e-4, "eval")
        assert_close(net(x[:, :, 0].to(dev), pool=False), onet(x[:, :, 0], pool=False), 3e-4, "eval single frame")


@pytest.mark.parametrize("ftype", ["hidsep", "MixVis"])
def test_inference_wrapper_vs_oracle(dev, ftype):
    """inference.py:29-160: AO, AV (two single frames, 5-D clips cut to the first frame), duet (one frame list, no
    img_activation) and MixVis forwards of the inference wrapper against the oracle restatement; eval-mode nets."""
    P = _pkg()
    from oracle import nets as O, inference as OI
    torch.manual_seed(3)
    gen = torch.Generator().manual_seed(3)
    osnd = O.Unet(fc_dim=2, num_downs=5, ngf=8, fusion_type=ftype, att_type="sig")
    O.wide_init(osnd, gen)
    ofrm = O.VisualNet(fc_dim=32, pool_type="maxpool", dilate_scale=16)
    snd = P.models.Unet(fc_dim=2, num_downs=5, ngf=8, fusion_type=ftype, att_type="sig")
    frm = P.models.ResnetDilated(None, fc_dim=32, pool_type="maxpool")
    snd.load_state_dict(osnd.state_dict()); frm.load_state_dict(ofrm.state_dict())
    snd, frm = snd.to(dev).eval(), frm.to(dev).eval()
    osnd.eval(); ofrm.eval()
    args = _args(fusion_type=ftype)
    wrap = P.inference.NetWrapper((snd, frm))
    mag = torch.rand(2, 1, 128, 64, generator=gen) ** 2
    phase = torch.rand(2, 1, 128, 64, generator=gen)
    clips = [torch.randn(2, 3, 2, 64, 64, generator=gen) for _ in range(2)]

    def cmp(out, ref, keys):
        for n in range(2):
            assert_close(out["pred_masks"][n], ref["pred_masks"][n], 3e-4, f"mask {n}")
        assert_close(out["mag_mix"], ref["mag_mix"], 1e-5, "warped mixture")
        for k in keys:
            assert_close(out[k], ref[k], 3e-4, k)
    with torch.no_grad():
        if ftype == "MixVis":
            out = wrap((mag.to(dev), phase.to(dev)), [c.to(dev) for c in clips], args, True)
            cmp(out, OI.forward((osnd, ofrm), (mag, phase), [c.clone() for c in clips], args, True), ["maps"])
            return
        snd.ao_draws = osnd.levels()[-1].fusion.ao_draws = torch.tensor([True, False])
        out = wrap((mag.to(dev), phase.to(dev)), None, args, False)
        cmp(out, OI.forward((osnd, ofrm), (mag, phase), None, args, False), [])
        assert out["phase_mix"].shape == phase.shape
        out = wrap((mag.to(dev), phase.to(dev)), [c.to(dev) for c in clips], args, True)
        cmp(out, OI.forward((osnd, ofrm), (mag, phase), [c.clone() for c in clips], args, True), ["maps", "match_loss"])
        duet = [torch.randn(2, 3, 64, 128, generator=gen)]
        out = wrap((mag.to(dev), phase.to(dev)), [duet[0].to(dev)], args, True)
        cmp(out, OI.forward((osnd, ofrm), (mag, phase), duet, args, True), ["maps", "match_loss"])


def test_bf16_mode_eval_and_audio_only_on_b16_images(dev):
    """bf16 mode OUTSIDE the benched AV train step, at channel widths that really take the B16 kernels (unet5 with ngf 32:
    32 ... 256 channels on 128x128 tiles; ResNet trunk 64 ... 512): a train-mode AO step (PIT, LeakyReLU / skip-gradient
    path of the encoder backward on B16 images), then eval-mode AV and AO forwards of the inference wrapper (running
    statistics instead of batch statistics through the same folded-affine staging) against the fp32 CPU oracle: mask MSE
    <= 1e-4 (the north-star bound), and the B16 kernel families are asserted to have been launched."""
    P = _pkg()
    from oracle import nets as O, step as OS, criterion as OC, inference as OI
    K = P.kernels
    torch.manual_seed(4)
    gen = torch.Generator().manual_seed(4)
    osnd = O.Unet(fc_dim=2, num_downs=5, ngf=32, fusion_type="hidsep", att_type="sig")
    O.wide_init(osnd, gen)
    ofrm = O.VisualNet(fc_dim=128, pool_type="maxpool", dilate_scale=16)
    snd = P.models.Unet(fc_dim=2, num_downs=5, ngf=32, fusion_type="hidsep", att_type="sig")
    frm = P.models.ResnetDilated(None, fc_dim=128, pool_type="maxpool")
    snd.load_state_dict(osnd.state_dict()); frm.load_state_dict(ofrm.state_dict())
    snd, frm = snd.to(dev), frm.to(dev)
    args = _args(fusion_type="hidsep")
    B = 4
    srcs = [torch.rand(B, 1, 128, 128, generator=gen) ** 2 for _ in range(2)]
    frames = [torch.randn(B, 3, 2, 96, 96, generator=gen) for _ in range(2)]
    mb = P.ModelBuilder()
    wrap = P.NetWrapper((snd, frm), mb.build_criterion("bce", use_pit=True), mb.build_criterion("bce"))
    opt = P.create_optimizer((snd, frm), args)
    owrap = OS.NetWrapper((osnd, ofrm), OC.build_criterion("bce", True), OC.build_criterion("bce"))
    oopt = OS.create_optimizer((osnd, ofrm), args)
    seen = set()
    orig = {n: getattr(K.Conv, n) for n in ("fwd", "dgrad", "wgrad")}

    def spy(name):
        def f(cv, *a, **kw):
            seen.add(cv.kernel_name(name, True))
            return orig[name](cv, *a, **kw)
        return f
    K.set_precision("bf16")
    for n in orig:
        setattr(K.Conv, n, spy(n))
    try:
        for use_vis in (True, False):
            draws = torch.tensor([True, False, False, True])
            snd.ao_draws = draws
            osnd.levels()[-1].fusion.ao_draws = draws
            gb = {"mag_mix": (srcs[0] + srcs[1]).to(dev), "mags": [x.clone().to(dev) for x in srcs], "frames": [f.to(dev) for f in frames]}
            cb = {"mag_mix": srcs[0] + srcs[1], "mags": [x.clone() for x in srcs], "frames": frames}
            err, _, outs = P.net_wrapper.train_step_async(wrap, gb, opt, use_vis, args)
            oerr, _, oouts = OS.train_step(owrap, cb, oopt, use_vis, args)
            mse = max(((a.detach().cpu() - b.detach()) ** 2).mean().item() for a, b in zip(outs["pred_masks"], oouts["pred_masks"]))
            print(f"bf16 / B16 mid-size {'AV' if use_vis else 'AO'} train step: err hip={err.item():.6f} oracle={oerr:.6f} mask-MSE={mse:.2e}")
            assert mse <= 1e-4 and abs(err.item() - oerr) <= 5e-3 * max(1.0, abs(oerr))
        assert {"convbf_kernel", "wgradb_kernel"} <= seen, seen
        snd.eval(); frm.eval(); osnd.eval(); ofrm.eval()
        iw = P.inference.NetWrapper((snd, frm))
        mag, phase = srcs[0] + srcs[1], torch.rand(B, 1, 128, 128, generator=gen)
        with torch.no_grad():
            for use_vis in (False, True):
                clips = [f.clone() for f in frames] if use_vis else None
                out = iw((mag.to(dev), phase.to(dev)), [c.to(dev) for c in clips] if use_vis else None, args, use_vis)
                ref = OI.forward((osnd, ofrm), (mag, phase), clips, args, use_vis)
                mse = max(((out["pred_masks"][n].cpu() - ref["pred_masks"][n]) ** 2).mean().item() for n in range(2))
                print(f"bf16 / B16 eval-mode {'AV' if use_vis else 'AO'} forward: mask-MSE={mse:.2e}")
                assert mse <= 1e-4
    finally:
        for n, f in orig.items():
            setattr(K.Conv, n, f)
        K.set_precision("f32")


def test_checkpoint_resume_on_gpu(dev, tmp_path):
    """checkpoint() -> rebuild through ModelBuilder(weights=...) + optimizer state: the resumed run takes the same
    next step as the uninterrupted one (the reference drops the momentum; optim_latest.pth is this build's extra)."""
    P = _pkg()
    a = _args()
    a.ckpt, a.best_err = str(tmp_path), float("inf")
    mb = P.ModelBuilder()

    def make(ws="", wf=""):
        torch.manual_seed(11)
        snd = mb.build_sound(arch="unet5", fc_dim=2, weights=ws, fusion_type="hidsep", att_type="sig").to(dev)
        frm = mb.build_frame(arch="resnet18dilated", fc_dim=256, pool_type="maxpool", weights=wf).to(dev)
        wrap = P.NetWrapper((snd, frm), mb.build_criterion("bce", use_pit=True), mb.build_criterion("bce"))
        return snd, frm, wrap, P.create_optimizer((snd, frm), a)
    gen = torch.Generator().manual_seed(2)
    srcs = [torch.rand(2, 1, 64, 64, generator=gen) ** 2 for _ in range(2)]
    frames = [torch.randn(2, 3, 1, 64, 64, generator=gen) for _ in range(2)]

    def batch():
        return {"mag_mix": (srcs[0] + srcs[1]).to(dev), "mags": [s.clone().to(dev) for s in srcs],
                "frames": [f.to(dev) for f in frames]}
    snd, frm, wrap, opt = make()
    for _ in range(2):
        P.train_step(wrap, batch(), opt, True, a)
    P.checkpoint.checkpoint((snd, frm), {"val_ao": {"si_sdr": [1.5]}}, 2, a, optimizer=opt)
    e3, _ = P.train_step(wrap, batch(), opt, True, a)
    ws, wf = P.checkpoint.resume_paths(a)
    snd2, frm2, wrap2, opt2 = make(ws, wf)
    assert P.checkpoint.load_optimizer(opt2, a) == 2
    e3b, _ = P.train_step(wrap2, batch(), opt2, True, a)
    assert abs(e3 - e3b) < 1e-6, (e3, e3b)
    for (k, p), (_, q) in zip(snd.named_parameters(), snd2.named_parameters()):
        # atomics-order noise only (a dropped momentum buffer shows as ~1e-1); BN biases here are ~1e-9 (pure noise)
        assert (q - p).abs().max().item() <= 1e-4 * max(p.abs().max().item(), 1e-3), k


def test_loader_workers_after_gpu_init(dev, tmp_path):
    """DataLoader WORKERS (fork()ed children) after this process has initialised HIP and launched kernels: the collate
    runs GPU-free in the children (no pin_memory / hipHostMalloc on an inherited HIP runtime), the parent's pin thread
    pins the batches, and they are identical to the workers=0 batches (train.py's default path is --workers 32)."""
    P = _pkg()
    from test_dataset import _make_disk_dataset
    from avsep_amd import dataset as PD
    x = torch.ones(8, device=dev)
    P.kernels.channel_stats(torch.rand(2, 4, 8, 8, device=dev), P.kernels.zeros_stats(4, x))   # HIP is live in this process
    torch.cuda.synchronize()
    lst = _make_disk_dataset(str(tmp_path))
    a = P.ArgParser().parse_train_arguments(
        ["--num_frames", "2", "--stride_frames", "2", "--imgSize", "64", "--audLen", "16383", "--margin", "1.0",
         "--train_repeat", "1"], verbose=False)
    got = {}
    for workers in (0, 2):
        loader = PD.make_loader([lst], a, "val", batch_size=2, shuffle=False, workers=workers)
        got[workers] = [b for _, b in zip(range(3), loader)]
    for b0, b2 in zip(got[0], got[2]):
        assert b2["audio_mix"].is_pinned() and b2["frames"][0].is_pinned()
        assert torch.equal(b0["audio_mix"], b2["audio_mix"]) and torch.equal(b0["frames"][1], b2["frames"][1])
        assert b0["id"] == b2["id"]
        dev_b = PD.to_device(b2, dev)
        assert torch.equal(dev_b["audios"][0].cpu(), b0["audios"][0])


def test_loader_to_gpu_step(dev, tmp_path):
    """Loader contract end to end: on-disk wav/jpg dataset -> pinned collate -> async H2D -> STFT on the GPU
    (NetWrapper.attach_stft replaces the loader-side librosa STFT) -> one AV and one AO train step; the GPU STFT of
    the loaded mixture equals the oracle STFT of the same waveform."""
    P = _pkg()
    from test_dataset import _make_disk_dataset
    from oracle import stft as OST
    from avsep_amd import dataset as PD
    lst = _make_disk_dataset(str(tmp_path))
    a = P.ArgParser().parse_train_arguments(
        ["--num_frames", "2", "--stride_frames", "2", "--imgSize", "64", "--audLen", "16383", "--margin", "1.0",
         "--train_repeat", "1", "--log_freq", "1", "--arch_sound", "unet5", "--num_channels", "2",
         "--fusion_type", "hidsep", "--att_type", "sig", "--not_pool_vis", "--img_activation", "relu",
         "--match_weight", "0.1"], verbose=False)
    loader = PD.make_loader([lst], a, "train", batch_size=2, shuffle=False)
    host = next(iter(loader))
    assert host["audio_mix"].is_pinned()
    batch = PD.to_device(host, dev)
    torch.manual_seed(0)
    snd = P.models.Unet(fc_dim=2, num_downs=5, ngf=8, fusion_type="hidsep", att_type="sig").to(dev)
    frm = P.models.ResnetDilated(None, fc_dim=32, pool_type="maxpool").to(dev)
    mb = P.ModelBuilder()
    wrap = P.NetWrapper((snd, frm), mb.build_criterion("bce", use_pit=True), mb.build_criterion("bce"))
    opt = P.create_optimizer((snd, frm), a)
    wrap.attach_stft(batch, a)
    mag_ref, _ = OST.stft_mag_phase(host["audio_mix"][0].numpy())
    assert_close(batch["mag_mix"][0, 0], torch.from_numpy(mag_ref), 2e-4, "GPU STFT of the loaded mixture")
    for use_vis in (True, False):
        b = PD.to_device(host, dev)
        err, match = P.train_step(wrap, b, opt, use_vis, a)
        assert err == err and 0.0 < err < 5.0, err


def _dp_setup(P, dev, world, shard):
    """Identical replica (seed 3) + the shard's slice of the global batch (seed 20 + shard)."""
    a = _args(log_freq=0)
    torch.manual_seed(3)
    snd = P.models.Unet(fc_dim=2, num_downs=5, ngf=8, fusion_type="hidsep", att_type="sig").to(dev)
    frm = P.models.ResnetDilated(None, fc_dim=32, pool_type="maxpool").to(dev)
    snd.ao_draws = torch.tensor([True, False])              # pin the audio-only coin: replicas and reference agree
    mb = P.ModelBuilder()
    wrap = P.NetWrapper((snd, frm), mb.build_criterion("bce", use_pit=True), mb.build_criterion("bce"))
    opt = P.create_optimizer((snd, frm), a, world_size=world)
    return a, snd, frm, wrap, opt


def _dp_batch(dev, shard):
    gen = torch.Generator().manual_seed(20 + shard)
    srcs = [torch.rand(2, 1, 64, 64, generator=gen) ** 2 for _ in range(2)]
    frames = [torch.randn(2, 3, 1, 64, 64, generator=gen) for _ in range(2)]
    return {"mag_mix": (srcs[0] + srcs[1]).to(dev), "mags": [s.clone().to(dev) for s in srcs], "frames": [f.to(dev) for f in frames]}


def _dp_worker(rank, world, port, out, overlap):
    import os
    import sys
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port), AVSEP_DP_BACKEND="gloo", AVSEP_DP_OVERLAP=overlap)
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import avsep_amd as P
    import torch.distributed as dist
    r, w, dev = P.dp.init_from_env()
    a, snd, frm, wrap, opt = _dp_setup(P, dev, w, rank)
    reduced, orig = [], opt.reduce_gradients

    def spy(only=None):                                      # the mean gradient every rank is about to apply
        active, scale = orig(only)
        reduced.append(((opt.flat_grad * scale).cpu(), [list(g["range"]) for g in active]))
        return active, scale
    opt.reduce_gradients = spy
    losses = []
    for it in range(4):
        err, _, _ = P.net_wrapper.train_step_async(wrap, _dp_batch(dev, rank), opt, it % 2 == 0, a)
        losses.append(float(err))
    flat = torch.cat([p.detach().reshape(-1) for p in list(snd.parameters()) + list(frm.parameters())]).cpu()
    torch.save({"losses": losses, "params": flat, "early": opt.early_reductions, "reduced": reduced[:2]}, f"{out}.{overlap}.{rank}")
    dist.barrier()
    dist.destroy_process_group()


def _dp_reference(dev):
    """Single process: the two shards' gradients computed one after the other from the same parameters (BatchNorm
    statistics stay per shard, as on two ranks), their MEAN, then the SGD update with that mean — for an AV step and the
    AO step that follows it.  Returns [(mean flat gradient, active ranges)] in FlatSGD's buffer order."""
    P = _pkg()
    a, snd, frm, wrap, opt = _dp_setup(P, dev, 1, 0)
    out = []
    for use_vis in (True, False):
        grads = []
        for shard in range(2):
            opt.zero_grad()
            wrap.train()
            err, _ = wrap.forward(_dp_batch(dev, shard), a, use_vis)
            err.mean().backward()
            opt._collect()
            grads.append(opt.flat_grad.clone())
        mean = (grads[0] + grads[1]) / 2
        opt.flat_grad.copy_(mean)
        only = None if use_vis else ("sound",)
        ranges = [list(g["range"]) for g in opt.param_groups if only is None or g["name"] in only]
        out.append((mean.cpu(), ranges))
        opt.step(only=only)                                  # what the two ranks apply; zero_grad() must not run first
    return out


def test_data_parallel_two_ranks_early_allreduce(dev, tmp_path):
    """N > 1 path on the GPU (2 ranks sharing the device over gloo; RCCL refuses two ranks on one GPU), REAL model:
      * the gradient every rank applies (flat buffer after FlatSGD.reduce_gradients, x 1/world) equals the MEAN of the
        two shards' single-process gradients to <= 1e-5, for an AV step (all three parameter groups) and for the AO step
        after it (U-Net group only) — what DataParallel's err.mean() at main.py:562 computes;
      * replicas stay identical, and issuing the U-Net's all-reduce early (overlapping the visual backward) gives the
        same parameters as the single all-reduce in step()."""
    import socket
    import torch.multiprocessing as mp
    res = {}
    for overlap in ("0", "1"):
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        mp.spawn(_dp_worker, args=(2, port, str(tmp_path / "r"), overlap), nprocs=2, join=True)
        res[overlap] = [torch.load(str(tmp_path / "r") + f".{overlap}.{r}") for r in range(2)]
        assert torch.equal(res[overlap][0]["params"], res[overlap][1]["params"]), "replicas diverged"
    assert res["0"][0]["early"] == 0 and res["1"][0]["early"] == 4        # every AV and AO step took the early path
    assert_close(res["1"][0]["params"], res["0"][0]["params"], 1e-5, "overlapped vs single all-reduce")
    for x, y in zip(res["0"][0]["losses"], res["1"][0]["losses"]):
        assert abs(x - y) < 1e-5
    ref = _dp_reference(dev)
    for overlap in ("0", "1"):
        for rank in range(2):
            for step, ((got, ranges), (want, want_ranges)) in enumerate(zip(res[overlap][rank]["reduced"], ref)):
                assert ranges == want_ranges, (overlap, rank, step, ranges, want_ranges)
                for lo, hi in ranges:
                    assert_close(got[lo:hi], want[lo:hi], 1e-5,
                                 f"all-reduced gradient vs mean of the shard gradients (overlap {overlap}, rank {rank}, "
                                 f"{'AV' if step == 0 else 'AO'} step, range {lo}:{hi})")



def _rccl_worker(rank, port, out):
    """Fresh process: a 1-rank `nccl` (= RCCL) group first, then the REAL full-size model with the collectives forced."""
    import os
    import sys
    import time
    os.environ.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch.distributed as dist
    import avsep_amd as P
    r, w, dev = P.dp.init_from_env(backend="nccl", force=True)          # before any kernel of this library runs
    assert dist.is_initialized() and dist.get_backend() == "nccl" and dist.get_world_size() == 1
    a = P.arguments.train_music_args()
    a.stft_pad_mode = "reflect"
    raw = P.synth.make_batch(4, a.num_mix, a.num_frames, 224, a.audLen, seed=91, device=dev)
    calls = []
    orig_ar = dist.all_reduce

    def spy(t, *args, **kw):
        calls.append(t.numel())
        return orig_ar(t, *args, **kw)
    dist.all_reduce = spy
    res = {}
    for forced in (True, False):
        torch.manual_seed(5)
        mb = P.ModelBuilder()
        snd = mb.build_sound(arch=a.arch_sound, fc_dim=a.num_channels, fusion_type=a.fusion_type, att_type=a.att_type)
        frm = mb.build_frame(arch=a.arch_frame, fc_dim=a.vis_channels, pool_type=a.img_pool)
        snd, frm = snd.to(dev), frm.to(dev)
        wrap = P.NetWrapper((snd, frm), mb.build_criterion(a.loss, use_pit=True), mb.build_criterion(a.loss))
        opt = P.create_optimizer((snd, frm), a, world_size=1, force_collective=forced)
        losses = []
        for use_vis in (True, True, False):
            snd.ao_draws = torch.tensor([True, False, False, True])
            gb = {"audios": list(raw["audios"]), "audio_mix": raw["audio_mix"], "frames": list(raw["frames"])}
            err, _, _ = P.net_wrapper.train_step_async(wrap, gb, opt, use_vis, a)
            losses.append(float(err))
        res[forced] = {"losses": losses, "params": opt.flat_param.detach().cpu().clone(), "early": opt.early_reductions,
                       "ranges": [list(g["range"]) for g in opt.param_groups]}
        if forced:
            res["calls"] = list(calls)
            # the step's one buffer through RCCL, on its own: 10 all-reduces of the full flat gradient
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(10):
                orig_ar(opt.flat_grad)
            torch.cuda.synchronize()
            res["allreduce_ms"] = (time.perf_counter() - t0) * 100.0
            res["bytes"] = opt.flat_grad.numel() * 4
    dist.all_reduce = orig_ar
    torch.save(res, out)
    dist.barrier()
    dist.destroy_process_group()


def test_rccl_one_rank_group_carries_the_flat_gradient(dev, tmp_path):
    """RCCL readiness on one GPU (main.py:660-662's replacement): a child process initialises a 1-rank `nccl` process
    group before anything else, then runs two AV steps and one AO step of the REAL full-size model (unet7 + resnet18dilated,
    256x256 tiles, 3x224^2 frames) with the collectives FORCED — FlatSGD issues the early all-reduce of the U-Net range from
    the autograd node (overlapping the visual trunk's backward, RCCL's own stream against this library's 512-thread
    workgroups) and the late one in step().  A 1-rank sum is the identity and scale = 1/world = 1, so the parameters must
    equal those of the same steps without any collective (to the 1e-6 run-to-run noise of the atomics); the all-reduce sizes must be the U-Net range (early)
    and the visual ranges (late) on AV steps, and only the U-Net range on the AO step."""
    import socket
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / "rccl.pt")
    mp.spawn(_rccl_worker, args=(port, out), nprocs=1, join=True)
    res = torch.load(out)
    f, n = res[True], res[False]
    assert f["early"] == 3 and n["early"] == 0, (f["early"], n["early"])
    (a0, a1), rest = f["ranges"][0], f["ranges"][1:]
    unet, vis = a1 - a0, max(r[1] for r in rest) - min(r[0] for r in rest)
    assert res["calls"] == [unet, vis, unet, vis, unet], (res["calls"], unet, vis)
    # a 1-rank sum is the identity; the two runs still differ by the order of the fp64 statistics atomics (~1e-7)
    for x, y in zip(f["losses"], n["losses"]):
        assert abs(x - y) <= 1e-6 * max(1.0, abs(y)), (f["losses"], n["losses"])
    assert_close(f["params"], n["params"], 1e-6, "parameters after 3 steps, collectives forced vs none")
    assert all(math.isfinite(x) for x in f["losses"])
    print(f"RCCL 1-rank group: {res['bytes'] / 1e6:.1f} MB flat gradient, {res['allreduce_ms']:.3f} ms per all-reduce; "
          f"early all-reduces {f['early']}, sizes {res['calls']}")


@pytest.mark.parametrize("nsrc,prec", [(2, "f32"), (3, "f32"), (2, "bf16")])
def test_trunk_passes_on_forked_streams_equal_back_to_back(dev, nsrc, prec):
    """NetWrapper._frame_features issues the visual trunk's pass over source n > 0 on its own HIP stream (main.py:117-121 calls
    net_frame once per source: the passes are independent).  What they share — packed weight images, the BatchNorm running
    statistics (updated in source order), the flat gradient buffer and its scratch — is ordered by events: three train steps
    (AV, AO, AV) must leave the same losses, parameters, momentum and BatchNorm buffers as the same steps with every pass on one
    stream, up to the run-to-run noise of the fp64 statistics atomics."""
    P = _pkg()
    K = P.kernels
    from oracle import nets as O
    gen = torch.Generator().manual_seed(21)
    osnd = O.Unet(fc_dim=nsrc, num_downs=5, ngf=8, fusion_type="hidsep", att_type="sig")
    O.wide_init(osnd, gen)
    ofrm = O.VisualNet(fc_dim=64 // nsrc, pool_type="maxpool", dilate_scale=16)
    args = _args(num_mix=nsrc)
    srcs = [torch.rand(3, 1, 64, 64, generator=gen) ** 2 for _ in range(nsrc)]
    frames = [torch.randn(3, 3, 2, 64, 64, generator=gen) for _ in range(nsrc)]
    mb = P.ModelBuilder()
    res = {}
    K.set_precision(prec)
    try:
        for fork in (True, False):
            torch.manual_seed(5)                       # the audio-only step of three sources draws a permutation per sample
            snd = P.models.Unet(fc_dim=nsrc, num_downs=5, ngf=8, fusion_type="hidsep", att_type="sig")
            frm = P.models.ResnetDilated(None, fc_dim=64 // nsrc, pool_type="maxpool")
            snd.load_state_dict(osnd.state_dict()); frm.load_state_dict(ofrm.state_dict())
            snd, frm = snd.to(dev), frm.to(dev)
            wrap = P.NetWrapper((snd, frm), mb.build_criterion("bce", use_pit=True), mb.build_criterion("bce"))
            wrap.fork_sources = snd.fork_pair = fork            # the trunk's passes and the U-Net's two decoder passes
            snd.encoder_bwd_on_side = fork
            opt = P.create_optimizer((snd, frm), args)
            losses = []
            for it, use_vis in enumerate((True, False, True)):
                if nsrc == 2:
                    snd.ao_draws = torch.tensor([it % 2 == 0, True, False])
                b = {"mag_mix": sum(srcs).to(dev), "mags": [s.clone().to(dev) for s in srcs], "frames": [f.to(dev) for f in frames]}
                err, match, outs = P.net_wrapper.train_step_async(wrap, b, opt, use_vis, args)
                losses.append(err.item())
            torch.cuda.synchronize()
            assert (len(wrap.__dict__.get("_src_streams", [])) == nsrc) == fork      # one stream per source (issued before the STFT)
            res[fork] = (losses, {k: v.detach().double().cpu() for k, v in list(snd.state_dict().items()) + list(frm.state_dict().items())},
                         opt.flat_buf.detach().double().cpu())
            if fork:
                # a caller on a stream of its own: the end-of-backward ordering goes to THAT stream (the engine runs the callback
                # on a worker thread whose current stream is the default one)
                mine = torch.cuda.Stream()
                mine.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(mine):
                    opt.zero_grad()
                    with K.pack_scope():
                        e2, _ = wrap.forward(b, args, True)
                        e2.mean().backward()
                    g_now = opt.flat_grad.clone()
                torch.cuda.synchronize()
                assert torch.equal(g_now, opt.flat_grad), "gradient read on the caller's stream right after backward()"
                # The caller's stream is taken where the backward pass starts (NetWrapper.forward / arm_early_reduce), not in
                # zero_grad(): zero_grad() issued on ANOTHER stream, then two ACCUMULATED backward passes on `mine` — the
                # gradient read on `mine` right after the second backward() must be complete and equal twice the single one
                other = torch.cuda.Stream()
                other.wait_stream(mine)
                with torch.cuda.stream(other):
                    opt.zero_grad()
                mine.wait_stream(other)
                with torch.cuda.stream(mine):
                    with K.pack_scope():
                        for _ in range(2):
                            e2, _ = wrap.forward(b, args, True)
                            e2.mean().backward()
                    g_twice = opt.flat_grad.clone()
                torch.cuda.synchronize()
                assert opt._caller_stream == mine
                assert torch.equal(g_twice, opt.flat_grad), "gradient read right after the second accumulated backward()"
                # (train-mode BatchNorm updates no parameter between the two passes: the second gradient equals the first)
                assert_close(g_twice.double().cpu(), 2.0 * g_now.double().cpu(), 1e-5 if prec == "f32" else 2e-2,
                             "two accumulated backward passes = twice the gradient")
                with pytest.raises(P.lib.AvsepError):          # a fold of a parameter that was never handed a scratch view
                    opt.fold_scratch([next(iter(frm.parameters()))])
    finally:
        K.set_precision("f32")
    tol = 1e-6 if prec == "f32" else 2e-3       # bf16: a last-bit difference of a statistic can flip a bf16 rounding downstream
    for x, y in zip(res[True][0], res[False][0]):
        assert abs(x - y) <= tol * max(1.0, abs(y)), (res[True][0], res[False][0])
    for (k, v), (_, w) in zip(res[True][1].items(), res[False][1].items()):
        assert_close(v, w, tol, "forked vs one stream: " + k)
    assert_close(res[True][2], res[False][2], tol * 10, "momentum buffers")


_CAPTURE_SCRIPT = r"""
import argparse, sys, torch
sys.path.insert(0, sys.argv[1])
import avsep_amd as P
K = P.kernels
dev = torch.device("cuda:0")
a = argparse.Namespace(num_mix=2, log_freq=0, weighted_loss=1, binary_mask=1, output_activation="sigmoid", img_activation="relu",
                       not_pool_vis=False, fusion_type="hidsep", match_weight=0.1, lr_sound=1e-3, lr_frame=1e-4, fix_vis=False,
                       beta1=0.9, weight_decay=1e-4, stft_frame=1022, stft_hop=256)
gen = torch.Generator().manual_seed(1)
srcs = [(torch.rand(2, 1, 64, 64, generator=gen) ** 2).to(dev) for _ in range(2)]
frames = [torch.randn(2, 3, 2, 64, 64, generator=gen).to(dev) for _ in range(2)]
def batch(): return {"mag_mix": srcs[0] + srcs[1], "mags": [s.clone() for s in srcs], "frames": list(frames)}
def model():
    torch.manual_seed(0)
    snd = P.models.Unet(fc_dim=2, num_downs=5, ngf=8, fusion_type="hidsep", att_type="sig").to(dev)
    frm = P.models.ResnetDilated(None, fc_dim=32, pool_type="maxpool").to(dev)
    mb = P.ModelBuilder()
    wrap = P.NetWrapper((snd, frm), mb.build_criterion("bce", use_pit=True), mb.build_criterion("bce"))
    assert wrap.fork_sources and wrap.early_trunk and getattr(snd, "fork_pair", True)          # the shipped topology
    return wrap, P.create_optimizer((snd, frm), a)
# eager: 5 warm-up steps + 3 more
wrap, opt = model()
for _ in range(8): e_eager, _, _ = P.net_wrapper.train_step_async(wrap, batch(), opt, True, a)
torch.cuda.synchronize()
ref = float(e_eager)
# captured: 5 eager warm-up steps on the capture stream, one captured step, replayed 3 times
wrap, opt = model()
s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(4): P.net_wrapper.train_step_async(wrap, batch(), opt, True, a)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=s):
    for ar in K._arenas.values():
        ar.buf.zero_(); ar.off = 0
    err, match, _ = P.net_wrapper.train_step_async(wrap, batch(), opt, True, a)
torch.cuda.synchronize()
for _ in range(4): g.replay()
torch.cuda.synchronize()
print("CAPTURE_OK eager %.8f replay %.8f" % (ref, float(err)))
assert abs(float(err) - ref) <= 1e-5 * max(1.0, abs(ref)), (float(err), ref)
"""


def test_train_step_captures_into_a_hip_graph(dev, tmp_path):
    """The whole AV train step (forked trunk passes, forked decoder pair, autograd backward, end-of-backward callback, fused
    SGD) captured into ONE HIP graph and replayed: same loss trajectory as eager.  Under capture NetWrapper.forward keeps
    source 0's trunk pass on the capturing stream (with every pass on a side stream hipStreamEndCapture crashed on ROCm 7.0:
    DESIGN.md 8c).  Run once, in a child process: a crash inside the HIP runtime must not take the test session down."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "capture_step.py"
    script.write_text(_CAPTURE_SCRIPT)
    r = subprocess.run([sys.executable, str(script), root], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "CAPTURE_OK" in r.stdout, (r.returncode, r.stdout[-2000:], r.stderr[-3000:])


def test_forked_streams_at_the_benched_size(dev):
    """The same comparison where the streams really overlap: the full-size model at the benched batch (unet7 +
    resnet18dilated, batch 64, 3 x 224^2 frames per source, fp32).  Four AV steps from identical seeds with the trunk's
    passes and the decoder pair forked over streams, against everything on one stream: losses equal to 2e-6 relative (run-to-run
    noise of the statistics atomics: 1e-7 on the first steps), parameters and BatchNorm buffers to 5e-5."""
    import os
    import sys
    P = _pkg()
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    res = {}
    for fork in (True, False):
        a, snd, frm, wrap = bench.build(P, dev, 0, "hip")
        wrap.fork_sources = snd.fork_pair = fork
        opt = P.create_optimizer((snd, frm), a)
        raw = P.synth.make_batch(64, a.num_mix, a.num_frames, 224, a.audLen, seed=1, device=dev)
        # first a forward + backward WITHOUT the optimizer step, and the flat gradient read on the calling stream at once, with
        # no device synchronisation in between: .backward()'s contract (results usable on the caller's stream) is restored by
        # FlatSGD._end_of_backward for the nodes that ran on forked streams.  (At the initial weights: after a few steps the
        # gradient of this random problem is chaotic — two forked runs differ by 25 % — while losses and weights still agree.)
        opt.zero_grad()
        with P.kernels.pack_scope():
            b = {"audios": list(raw["audios"]), "audio_mix": raw["audio_mix"], "frames": list(raw["frames"])}
            err, _ = wrap.forward(b, a, True)
            err.mean().backward()
        g_now = opt.flat_grad.clone()                      # ordered on the calling stream only
        losses = []
        for _ in range(4):
            b = {"audios": list(raw["audios"]), "audio_mix": raw["audio_mix"], "frames": list(raw["frames"])}
            err, _, _ = P.net_wrapper.train_step_async(wrap, b, opt, True, a)
            losses.append(float(err))
        torch.cuda.synchronize()
        nets = (snd, frm)
        res[fork] = (losses, torch.cat([p.detach().reshape(-1) for n in nets for p in n.parameters()]).double().cpu(),
                     torch.cat([b_.detach().reshape(-1).double() for n in nets for b_ in n.buffers()]).cpu(), g_now.double().cpu())
        del wrap, opt, snd, frm, raw, nets, g_now, err, b
        torch.cuda.empty_cache()
    for x, y in zip(res[True][0], res[False][0]):
        assert abs(x - y) <= 2e-6 * max(1.0, abs(y)), (res[True][0], res[False][0])
    # (two forked runs differ by 4e-5 after twelve steps: the noise grows with the step count; a race shows as 1e-2 and more)
    assert_close(res[True][1], res[False][1], 5e-5, "parameters after 4 full-size steps, forked vs one stream")
    assert_close(res[True][2], res[False][2], 5e-5, "BatchNorm buffers after 4 full-size steps, forked vs one stream")
    assert_close(res[True][3], res[False][3], 1e-4, "flat gradient read right after the first backward(), forked vs one stream")


@pytest.mark.parametrize("ftype,att,loss,binary,weighted,log_freq", [
    ("CoLoc_Sel", "cos", "l1", 0, 0, 0), ("hidsep", "cos", "l2", 0, 1, 0), ("CoLoc_Sel", "sig", "bce", 1, 1, 1)])
def test_step_variants_vs_oracle(dev, ftype, att, loss, binary, weighted, log_freq):
    """The remaining flag combinations of the train step: CoLoc_Sel fusion, cosine attention, ratio
    masks with L1 / L2, unweighted loss, with and without the log-frequency warp; AV and AO steps on small nets
    against the CPU oracle (same weights, same inputs): losses, masks and the parameters after SGD."""
    P = _pkg()
    from oracle import nets as O, step as OS, criterion as OC
    torch.manual_seed(8)
    gen = torch.Generator().manual_seed(8)
    osnd = O.Unet(fc_dim=2, num_downs=5, ngf=8, fusion_type=ftype, att_type=att)
    O.wide_init(osnd, gen)
    ofrm = O.VisualNet(fc_dim=32, pool_type="maxpool", dilate_scale=16)
    snd = P.models.Unet(fc_dim=2, num_downs=5, ngf=8, fusion_type=ftype, att_type=att)
    frm = P.models.ResnetDilated(None, fc_dim=32, pool_type="maxpool")
    snd.load_state_dict(osnd.state_dict()); frm.load_state_dict(ofrm.state_dict())
    snd, frm = snd.to(dev), frm.to(dev)
    args = _args(fusion_type=ftype, att_type=att, loss=loss, binary_mask=binary, weighted_loss=weighted, log_freq=log_freq)
    F_in = 128 if log_freq else 64                      # the warp always produces 256 bins; keep the small case small
    srcs = [torch.rand(2, 1, F_in, 64, generator=gen) ** 2 for _ in range(2)]
    frames = [torch.randn(2, 3, 2, 64, 64, generator=gen) for _ in range(2)]

    def batch(d):
        return {"mag_mix": (srcs[0] + srcs[1]).to(d), "mags": [s.clone().to(d) for s in srcs], "frames": [f.to(d) for f in frames]}
    mb = P.ModelBuilder()
    wrap = P.NetWrapper((snd, frm), mb.build_criterion(loss, use_pit=True), mb.build_criterion(loss))
    opt = P.create_optimizer((snd, frm), args)
    owrap = OS.NetWrapper((osnd, ofrm), OC.build_criterion(loss, True), OC.build_criterion(loss))
    oopt = OS.create_optimizer((osnd, ofrm), args)
    for it, use_vis in enumerate((True, False, True)):
        draws = torch.tensor([it % 2 == 0, True])
        snd.ao_draws = draws
        osnd.levels()[-1].fusion.ao_draws = draws
        err, match, outs = P.net_wrapper.train_step_async(wrap, batch(dev), opt, use_vis, args)
        oerr, omatch, oouts = OS.train_step(owrap, batch("cpu"), oopt, use_vis, args)
        assert abs(err.item() - oerr) < 3e-4 * max(1.0, abs(oerr)), (it, err.item(), oerr)
        if use_vis:
            assert abs(match.item() - omatch) < 3e-4
        for n in range(2):
            assert ((outs["pred_masks"][n].detach().cpu() - oouts["pred_masks"][n].detach()) ** 2).mean().item() < 1e-6
    osd = osnd.state_dict()
    for k, v in snd.state_dict().items():
        if v.dtype.is_floating_point and "running" not in k:
            assert_close(v, osd[k], 2e-3, "after 3 steps: " + k)


def test_training_driver_end_to_end(dev, tmp_path):
    """avsep_amd.train (the loop of main.py:572-763): AV/AO alternation by the reference's schedule rule, history with
    the reference's keys, evaluation (SI-SDR/SDR) at eval_iter, checkpoint files, lr drop, then --mode eval from the
    best checkpoint — on a tiny on-disk wav/jpg dataset."""
    import os
    P = _pkg()
    from avsep_amd import train as T
    from test_dataset import _make_disk_dataset
    lst = _make_disk_dataset(str(tmp_path))
    flags = ("--id run --ckpt {ck} --av_list_train {l} --ao_list_train {l} --list_val {l} --start_av_first --num_fsteps 0 "
             "--arch_sound unet5 --arch_frame resnet18dilated --img_pool maxpool --num_channels 2 --img_activation relu "
             "--output_activation sigmoid --vis_channels 256 --fusion_type hidsep --not_pool_vis --att_type sig "
             "--binary_mask 1 --loss bce --weighted_loss 1 --num_mix 2 --log_freq 1 --num_frames 1 --stride_frames 2 "
             "--imgSize 64 --audLen 16383 --margin 1.0 --batch_size_per_gpu 2 --workers 0 --train_repeat 1 --val_repeat 1 "
             "--lr_steps 5 --num_iters 8 --iter_per_av 2 --eval_iter 4 --disp_iter 2 --max_silent 0.87").format(
                 ck=str(tmp_path / "ck"), l=lst).split()
    # the schedule rule itself (main.py:578-581)
    a = P.ArgParser().parse_train_arguments(flags, verbose=False)
    assert [T.av_ao_schedule(i, a) for i in range(1, 7)] == [False, True, False, True, False, True]
    a.start_av_first, a.num_fsteps = False, 3
    assert [T.av_ao_schedule(i, a) for i in range(1, 7)] == [False, False, False, True, False, True]
    hist = T.cli(flags)
    assert hist["train"]["iter"] == [2, 4, 6] and all(e == e and 0 < e < 5 for e in hist["train"]["err"])
    assert hist["train_av"]["iter"] == [2, 4, 6] and hist["train_ao"]["iter"] == [2, 4, 6]
    assert hist["val_av"]["iter"] == [4] and hist["val_ao"]["iter"] == [4]
    assert all(v == v for v in hist["val_ao"]["si_sdr"] + hist["val_av"]["sdr"])        # finite, not NaN
    ck = str(tmp_path / "ck" / "run")
    assert sorted(os.listdir(ck)) == ["frame_best.pth", "frame_latest.pth", "history_latest.pth", "optim_latest.pth",
                                      "sound_best.pth", "sound_latest.pth"]
    ev = T.cli(flags + ["--mode", "eval"])
    assert ev["val_av"]["iter"] == [0] and ev["val_ao"]["iter"] == [0]
    assert abs(ev["val_ao"]["si_sdr"][0] - hist["val_ao"]["si_sdr"][0]) < 1e-3      # same weights as at iteration 4


def test_three_stage_driver_checkpoint_resume_and_lr(dev, tmp_path):
    """SoP++ three-stage driver (train.py --train_steps): checkpoint() must write the synthesizer and the attention module
    like SoP++/main.py:599-631, --load_ckpt must restore all four nets + momentum + iteration + DECAYED learning rates
    (the lr step of iteration 4 falls on the checkpoint iteration, as with the shipped flags), and --mode eval must score
    the weights that were saved (a randomly re-initialised synthesizer would not reproduce the SI-SDR)."""
    import os
    P = _pkg()
    from avsep_amd import train as T
    from test_dataset import _make_disk_dataset
    lst = _make_disk_dataset(str(tmp_path))
    flags = ("--id run3 --ckpt {ck} --av_list_train {l} --ao_list_train {l} --list_val {l} --start_av_first --num_fsteps 0 "
             "--arch_sound unet5 --arch_frame resnet18dilated --arch_synthesizer linear --img_pool maxpool --num_channels 8 "
             "--img_activation sigmoid --sound_activation no --output_activation sigmoid --vis_channels 8 --fusion_type Base "
             "--not_pool_vis --att_type sig --binary_mask 1 --loss bce --weighted_loss 1 --num_mix 2 --log_freq 1 --num_frames 1 "
             "--stride_frames 2 --imgSize 64 --audLen 16383 --margin 1.0 --batch_size_per_gpu 2 --workers 0 --train_repeat 1 "
             "--val_repeat 1 --lr_steps 4 --num_iters 5 --iter_per_av 2 --eval_iter 4 --disp_iter 2 --max_silent 0.87 "
             "--lr_synthesizer 1e-2 --train_steps 2 3 20").format(ck=str(tmp_path / "ck"), l=lst).split()
    hist = T.cli(flags)
    ck = str(tmp_path / "ck" / "run3")
    assert sorted(os.listdir(ck)) == ["frame_best.pth", "frame_latest.pth", "history_latest.pth", "net_pit_best.pth",
                                      "net_pit_latest.pth", "optim_latest.pth", "sound_best.pth", "sound_latest.pth",
                                      "synthesizer_best.pth", "synthesizer_latest.pth"]
    blob = torch.load(os.path.join(ck, "optim_latest.pth"))
    assert blob["itera"] == 4
    lrs = {g["name"]: g["lr"] for g in blob["state"]["groups"]}
    assert abs(lrs["sound"] - 1e-4) < 1e-12 and abs(lrs["synthesizer"] - 1e-3) < 1e-12, lrs    # decayed BEFORE the checkpoint
    syn_saved = torch.load(os.path.join(ck, "synthesizer_latest.pth"))
    assert (syn_saved["scale"] - 1.0).abs().max().item() > 1e-7, "the synthesizer trained: its scale left the init (ones)"
    # eval mode scores the saved weights: same SI-SDR as the in-training evaluation at iteration 4
    ev = T.cli(flags + ["--mode", "eval"])
    assert abs(ev["val_ao"]["si_sdr"][0] - hist["val_ao"]["si_sdr"][0]) < 1e-3
    assert abs(ev["val_av"]["si_sdr"][0] - hist["val_av"]["si_sdr"][0]) < 1e-3
    # resume: iteration counter, decayed rates and all four nets come back
    seen = {}
    orig = P.sopp.train_step_3stage

    def spy(model, batch, optimizer, use_vis, i, args):
        if not seen:
            seen["i"], seen["lr"] = i, {g["name"]: g["lr"] for g in optimizer.param_groups}
            seen["scale"] = model.net_synthesizer.scale.detach().cpu().clone()
        return orig(model, batch, optimizer, use_vis, i, args)
    T.sopp.train_step_3stage = spy
    try:
        T.cli([f if f != "5" else "7" for f in flags] + ["--load_ckpt", "1"])
    finally:
        T.sopp.train_step_3stage = orig
    assert seen["i"] == 5 and abs(seen["lr"]["sound"] - 1e-4) < 1e-12 and abs(seen["lr"]["synthesizer"] - 1e-3) < 1e-12, seen
    assert torch.equal(seen["scale"], syn_saved["scale"])
