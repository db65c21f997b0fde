"""Generate tests/golden/*.npz from the REFERENCE ITSELF (runs only in the build container).

TEST INFRASTRUCTURE ONLY.  The reference has no golden vectors of its own
(SURVEY.md §4), so the oracle is pinned against outputs of the reference's
importable modules, run here on CPU with the stub recipe of SURVEY.md §8(c):
third-party imports that are absent (torchvision, librosa, mmcv, ...) are
replaced by MagicMock *only so that `import main` succeeds*; none of their
arithmetic is used.  The reference cannot travel to the GPU box, so the vectors
(inputs + expected outputs, data only) are committed under tests/golden/.

Usage:  python -B oracle/gen_golden.py          (needs /root/reference)
Every case also asserts oracle == reference here, so a fixture is only written
for behaviour the oracle already reproduces.
"""
import argparse
import importlib.util
import os
import sys
import types
from unittest.mock import MagicMock

sys.dont_write_bytecode = True
import numpy as np
import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
OUT = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, ROOT)

from oracle import nets as O          # noqa: E402
from oracle import criterion as OC    # noqa: E402
from oracle import step as OS         # noqa: E402


def import_reference():
    for name in ["tkinter", "tkinter.messagebox", "imageio", "soundfile", "asteroid", "asteroid.metrics",
                 "librosa", "cv2", "torchvision", "torchvision.transforms",
                 "torchvision.transforms.functional", "torchaudio", "mmcv", "mmaction", "mmaction.models",
                 "mmaction.datasets", "mmaction.datasets.pipelines", "nis", "turtle", "curses"]:
        sys.modules.setdefault(name, MagicMock())
    sys.path.insert(0, REF)
    argv, sys.argv = sys.argv, ["x"]
    import main as ref_main  # noqa
    sys.argv = argv
    torch.Tensor.cuda = lambda self, *a, **k: self   # fusion_net.py:96 AO branch on a CPU box
    spec = importlib.util.spec_from_file_location("ref_sopp_att", REF + "/SoP++/attention_net.py")
    att = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(att)
    pkg = types.ModuleType("ref_sopp")
    pkg.__path__ = [REF + "/SoP++", REF + "/models"]
    sys.modules["ref_sopp"] = pkg
    spec = importlib.util.spec_from_file_location("ref_sopp.audio_net", REF + "/SoP++/audio_net.py")
    saud = importlib.util.module_from_spec(spec)
    sys.modules["ref_sopp.audio_net"] = saud
    spec.loader.exec_module(saud)
    return ref_main, att, saud


def close(a, b, tol=1e-5, what=""):
    a, b = torch.as_tensor(a), torch.as_tensor(b)
    err = (a - b).abs().max().item() if a.numel() else 0.0
    scale = max(1.0, b.abs().max().item() if b.numel() else 1.0)
    assert err <= tol * scale, f"{what}: oracle vs reference max|d|={err}"
    return err


def npy(d):
    return {k: (v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)) for k, v in d.items()}


def save(name, d):
    os.makedirs(OUT, exist_ok=True)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **npy(d))
    sz = os.path.getsize(os.path.join(OUT, name + ".npz"))
    print(f"wrote {name}.npz  ({sz/1024:.0f} KiB, {len(d)} arrays)")


class TVLike(nn.Module):
    """torch-only stand-in with torchvision.models.resnet18's child order."""

    def __init__(self):
        super().__init__()
        t = O.resnet18_trunk()
        (self.conv1, self.bn1, self.relu, self.maxpool,
         self.layer1, self.layer2, self.layer3, self.layer4) = list(t.children())
        self.avgpool = nn.AdaptiveAvgPool2d(1)
        self.fc = nn.Linear(512, 10)


def make_args(**kw):
    a = argparse.Namespace(num_mix=2, log_freq=1, weighted_loss=1, binary_mask=1, output_activation="sigmoid",
                           img_activation="relu", not_pool_vis=False, fusion_type="hidsep", match_weight=0.1,
                           device=torch.device("cpu"), lr_sound=1e-3, lr_frame=1e-4, fix_vis=False, beta1=0.9,
                           weight_decay=1e-4, load_clips=False)
    for k, v in kw.items():
        setattr(a, k, v)
    return a


# ----------------------------------------------------------------------------
def gen_prepare(ref_main):
    g = torch.Generator().manual_seed(11)
    B, T = 2, 24
    srcs = [torch.rand(B, 1, 512, T, generator=g) ** 2 * 3 for _ in range(2)]
    mix = (srcs[0] + srcs[1]) * (0.5 + torch.rand(B, 1, 512, T, generator=g))
    out = {"mag_mix": mix, "mags0": srcs[0], "mags1": srcs[1]}
    for tag, kw in [("bin_w", dict(binary_mask=1, weighted_loss=1, log_freq=1)),
                    ("ratio_now", dict(binary_mask=0, weighted_loss=0, log_freq=1)),
                    ("nolog", dict(binary_mask=1, weighted_loss=1, log_freq=0))]:
        args = make_args(**kw)
        w = ref_main.NetWrapper.__new__(ref_main.NetWrapper)
        nn.Module.__init__(w)
        w.load_clips = False
        r = w.prepare({"mag_mix": mix.clone(), "mags": [s.clone() for s in srcs]}, args, False, False)
        o = OS.prepare({"mag_mix": mix.clone(), "mags": [s.clone() for s in srcs]}, args)
        names = ["mags", "mag_mix", "log_mag_mix", "gt_masks", "weights"]
        for nme, rr, oo in zip(names, r, o):
            if isinstance(rr, list):
                for i, (x, y) in enumerate(zip(rr, oo)):
                    close(y, x, 1e-6, f"prepare {tag} {nme}{i}")
                    out[f"{tag}.{nme}{i}"] = x
            else:
                close(oo, rr, 1e-6, f"prepare {tag} {nme}")
                out[f"{tag}.{nme}"] = rr
    out["warpgrid_8x5"] = torch.from_numpy(ref_main.warpgrid(1, 8, 5, warp=True))
    out["unwarpgrid_8x5"] = torch.from_numpy(ref_main.warpgrid(1, 8, 5, warp=False))
    save("prepare", out)


def gen_fusion(ref_main):
    import models.fusion_net as RF
    g = torch.Generator().manual_seed(12)
    B, D, H, W = 3, 64, 5, 4
    out = {}
    for ftype, cls in [("hidsep", RF.CoLoc), ("CoLoc_Sel", RF.CoLoc_Sel), ("MixVis", RF.MixVis)]:
        for att in ("cos", "sig"):
            x = torch.randn(B, D, 2, 2, generator=g, requires_grad=True)
            nv = 1 if ftype == "MixVis" else 2
            Wv = W * 2 if ftype == "MixVis" else W
            vs = [torch.randn(B, D // 2, H, Wv, generator=g).relu().requires_grad_(True) for _ in range(nv)]
            cot = torch.randn(B, 2 * D, 2, 2, generator=g)
            res = {}
            for who, mod in (("ref", cls(att_type=att)), ("ora", O.Fusion(ftype, att))):
                for t in [x] + vs:
                    t.grad = None
                y, (ml, maps) = mod(x, vs)
                ((y * cot).sum() + 0.7 * ml.sum() + 0.01 * (maps ** 2).sum()).backward()
                res[who] = [y.detach(), ml.detach().reshape(-1), maps.detach(), x.grad.clone()] + [v.grad.clone() for v in vs]
            tag = f"{ftype}.{att}"
            for i, (r, o) in enumerate(zip(res["ref"], res["ora"])):
                close(o, r, 2e-5, f"fusion {tag} #{i}")
            out.update({f"{tag}.x": x, f"{tag}.cot": cot, f"{tag}.y": res["ref"][0], f"{tag}.match": res["ref"][1],
                        f"{tag}.maps": res["ref"][2], f"{tag}.dx": res["ref"][3]})
            for i, v in enumerate(vs):
                out[f"{tag}.v{i}"] = v
                out[f"{tag}.dv{i}"] = res["ref"][4 + i]
    # AO branch: pin the random draw through torch's global RNG
    x = torch.randn(4, D, 2, 2, generator=g)
    for seed in (0, 1, 5):
        torch.manual_seed(seed)
        draws = torch.rand(4) > 0.5
        torch.manual_seed(seed)
        y, meta = RF.CoLoc(att_type="cos")(x, None)
        assert meta == (None, None)
        close(O.ao_swap(x, draws), y, 0, "ao swap")
        out[f"ao.draws{seed}"] = draws
        out[f"ao.y{seed}"] = y
    # degenerate all-zero draw (one_hot width 1)
    orig = torch.rand
    torch.rand = lambda *a, **k: torch.zeros(*a)
    try:
        y, _ = RF.CoLoc(att_type="cos")(x, None)
    finally:
        torch.rand = orig
    close(O.ao_swap(x, torch.zeros(4, dtype=torch.bool)), y, 0, "ao swap degenerate")
    out["ao.x"] = x
    out["ao.y_allzero"] = y
    save("fusion", out)


def _copy_state(dst, src):
    missing = dst.load_state_dict(src.state_dict(), strict=True)
    return missing


def gen_unet(ref_main):
    import models.audio_net as RA
    out = {}
    for tag, downs, ngf, size, B, ftype, att in [("u5", 5, 8, 64, 2, "hidsep", "sig"),
                                                   ("u7", 7, 4, 256, 1, "hidsep", "cos"),
                                                   ("u6sel", 6, 4, 128, 2, "CoLoc_Sel", "sig")]:
        g = torch.Generator().manual_seed(100 + downs)
        ref = RA.Unet(fc_dim=2, num_downs=downs, ngf=ngf, fusion_type=ftype, att_type=att)
        O.wide_init(ref, g)
        ora = O.Unet(fc_dim=2, num_downs=downs, ngf=ngf, fusion_type=ftype, att_type=att)
        assert list(ora.state_dict().keys()) == list(ref.state_dict().keys()), "state_dict key order"
        _copy_state(ora, ref)
        x = torch.randn(B, 1, size, size, generator=g) * 2 - 4
        vs = [torch.randn(B, 4 * ngf, 4, 3, generator=g).relu().requires_grad_(True) for _ in range(2)]
        cot = torch.randn(B, 2, size, size, generator=g) / size
        res = {}
        for who, net in (("ref", ref), ("ora", ora)):
            net.train()
            net.zero_grad()
            for v in vs:
                v.grad = None
            y, (ml, maps) = net(x.clone(), vs)
            ((y * cot).sum() + 0.3 * ml).backward()
            grads = {k: p.grad.clone() for k, p in net.named_parameters()}
            bufs = {k: b.clone() for k, b in net.named_buffers()}
            # AO forward (pinned draw) on the same weights, train mode
            torch.manual_seed(3)
            draws = torch.rand(B) > 0.5
            if who == "ref":
                torch.manual_seed(3)
                yao, _ = net(x.clone(), None)
            else:
                net.levels()[-1].fusion.ao_draws = draws
                yao, _ = net(x.clone(), None)
            net.eval()
            with torch.no_grad():
                yev, (mlev, _) = net(x.clone(), [v.detach() for v in vs])
            res[who] = dict(y=y.detach(), ml=ml.detach().reshape(1), maps=maps.detach(), grads=grads, bufs=bufs,
                            dv=[v.grad.clone() for v in vs], yao=yao.detach(), yev=yev, draws=draws)
        r, o = res["ref"], res["ora"]
        close(o["y"], r["y"], 2e-5, f"unet {tag} y")
        close(o["ml"], r["ml"], 2e-5, f"unet {tag} match")
        close(o["yao"], r["yao"], 2e-5, f"unet {tag} y_ao")
        close(o["yev"], r["yev"], 2e-5, f"unet {tag} y_eval")
        for k in r["grads"]:
            close(o["grads"][k], r["grads"][k], 5e-4, f"unet {tag} grad {k}")
        for k in r["bufs"]:
            close(o["bufs"][k], r["bufs"][k], 1e-5, f"unet {tag} buf {k}")
        for a, b in zip(o["dv"], r["dv"]):
            close(a, b, 5e-4, f"unet {tag} dv")
        out[f"{tag}.x"] = x
        out[f"{tag}.cot"] = cot
        out[f"{tag}.draws"] = r["draws"]
        for i, v in enumerate(vs):
            out[f"{tag}.v{i}"] = v
            out[f"{tag}.dv{i}"] = r["dv"][i]
        for k, p in ref.state_dict().items():
            out[f"{tag}.w.{k}"] = p
        full = (tag == "u5")
        out[f"{tag}.y"] = r["y"] if full else r["y"][:, :, ::8, ::8]
        out[f"{tag}.y_sum"] = r["y"].double().sum().reshape(1)
        out[f"{tag}.y_abs"] = r["y"].double().abs().sum().reshape(1)
        out[f"{tag}.yao"] = r["yao"] if full else r["yao"][:, :, ::8, ::8]
        out[f"{tag}.yev"] = r["yev"] if full else r["yev"][:, :, ::8, ::8]
        out[f"{tag}.match"] = r["ml"]
        out[f"{tag}.maps"] = r["maps"]
        for k, gk in r["grads"].items():
            out[f"{tag}.g.{k}"] = gk
        for k, bk in r["bufs"].items():
            out[f"{tag}.b.{k}"] = bk
    save("unet", out)


def gen_criterion(ref_main):
    import models.criterion as RC
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(14)
    out = {}
    B, Fq, T = 3, 16, 12
    preds = [torch.rand(B, 1, Fq, T, generator=g) * 0.98 + 0.01 for _ in range(2)]
    tg = [(torch.rand(B, 1, Fq, T, generator=g) > 0.5).float() for _ in range(2)]
    w = torch.rand(B, 1, Fq, T, generator=g) * 3
    for kind, rc in (("bce", RC.BCELoss()), ("l1", RC.L1Loss()), ("l2", RC.L2Loss())):
        oc = OC.build_criterion(kind)
        close(oc(preds, tg, w), rc(preds, tg, w), 1e-6, kind)
        close(oc(preds[0], tg[0]), rc(preds[0], tg[0]), 1e-6, kind + " tensor")
        out[f"{kind}.list"] = rc(preds, tg, w).reshape(1)
        out[f"{kind}.tensor_now"] = rc(preds[0], tg[0]).reshape(1)
    out.update({"p0": preds[0], "p1": preds[1], "t0": tg[0], "t1": tg[1], "w": w})
    # PIT (incl. a tie sample and a swapped sample)
    P = torch.stack([preds[0][:, 0], preds[1][:, 0]], -1)
    Tt = torch.stack([tg[0][:, 0], tg[1][:, 0]], -1)
    Tt[1] = Tt[1].flip(-1)            # sample 1: permuted targets
    P[2, ..., 1] = P[2, ..., 0]
    Tt[2, ..., 1] = Tt[2, ..., 0]     # sample 2: exact tie
    W2 = torch.stack([w[:, 0]] * 2, -1)
    rp = RC.PitWrapper(F.binary_cross_entropy)
    op = OC.PitWrapper("bce")
    rl, rperm = rp(P, Tt, W2)
    ol, operm = op(P, Tt, W2)
    close(ol, rl, 1e-6, "pit loss")
    assert [tuple(p) for p in rperm] == [tuple(p) for p in operm], (rperm, operm)
    close(op.reorder_tensor(P, operm), rp.reorder_tensor(P, rperm), 0, "reorder")
    out.update({"pit.P": P, "pit.T": Tt, "pit.W": W2, "pit.loss": rl, "pit.perms": np.array(rperm),
                "pit.reordered": rp.reorder_tensor(P, rperm), "pit.mat": op.loss_matrix(P, Tt, W2)})
    save("criterion", out)


def gen_synth(ref_main):
    import models.synthesizer_net as RS
    g = torch.Generator().manual_seed(15)
    B, K = 2, 8
    fi = torch.randn(B, K, generator=g)
    fs = torch.randn(B, K, 6, 5, generator=g)
    fim = torch.randn(B, K, 3, 2, generator=g)
    out = {"fi": fi, "fs": fs, "fim": fim}
    for name, r, o in (("innerprod", RS.InnerProd(K), O.InnerProd(K)), ("bias", RS.Bias(), O.Bias())):
        with torch.no_grad():
            if name == "innerprod":
                r.scale.copy_(torch.rand(K, generator=g) + 0.5)
            r.bias.fill_(0.25)
        o.load_state_dict(r.state_dict())
        for fn, a in (("forward", fi), ("forward_nosum", fi), ("forward_pixelwise", fim)):
            yr, yo = getattr(r, fn)(a, fs), getattr(o, fn)(a, fs)
            close(yo, yr, 1e-5, f"{name}.{fn}")
            out[f"{name}.{fn}"] = yr
        for k, v in r.state_dict().items():
            out[f"{name}.w.{k}"] = v
    save("synthesizer", out)


def build_small_nets(ref_main, seed, ngf=8, downs=5):
    import models.audio_net as RA
    import models.vision_net as RV
    torch.manual_seed(seed)
    g = torch.Generator().manual_seed(seed)
    snd = RA.Unet(fc_dim=2, num_downs=downs, ngf=ngf, fusion_type="hidsep", att_type="sig")
    O.wide_init(snd, g)
    frm = RV.ResnetDilated(TVLike(), fc_dim=4 * ngf, pool_type="maxpool")
    return snd, frm


def gen_step(ref_main):
    """End-to-end NetWrapper.forward (AV + AO) and three train_steps through the reference's
    own main.py code, small nets (unet5 ngf=8 on 64x64 tiles after the warp is skipped... the warp
    needs 256 output bins, so log_freq=0 here; the warp itself is pinned by prepare.npz)."""
    seed = 21
    g = torch.Generator().manual_seed(seed)
    B, T, S, Fr = 2, 2, 64, 64
    srcs = [torch.rand(B, 1, S, S, generator=g) ** 2 for _ in range(2)]
    batch = {"mag_mix": srcs[0] + srcs[1], "mags": srcs,
             "frames": [torch.randn(B, 3, T, Fr, Fr, generator=g) for _ in range(2)]}
    args = make_args(log_freq=0)
    ref_main.args = args

    def clone_batch():
        return {"mag_mix": batch["mag_mix"].clone(), "mags": [m.clone() for m in batch["mags"]],
                "frames": [f.clone() for f in batch["frames"]]}

    rs, rf = build_small_nets(ref_main, seed)
    os_ = O.Unet(fc_dim=2, num_downs=5, ngf=8, fusion_type="hidsep", att_type="sig")
    of = O.VisualNet(fc_dim=32, pool_type="maxpool", dilate_scale=16)
    assert list(of.state_dict().keys()) == list(rf.state_dict().keys())
    os_.load_state_dict(rs.state_dict())
    of.load_state_dict(rf.state_dict())
    import torch.nn.functional as F
    import models.criterion as RC
    rw = ref_main.NetWrapper((rs, rf), RC.PitWrapper(F.binary_cross_entropy), RC.BCELoss())
    ow = OS.NetWrapper((os_, of), OC.build_criterion("bce", True), OC.build_criterion("bce"))
    ropt = ref_main.create_optimizer((rs, rf), args)
    oopt = OS.create_optimizer((os_, of), args)
    out = {"mag_mix": batch["mag_mix"], "mags0": srcs[0], "mags1": srcs[1],
           "frames0": batch["frames"][0], "frames1": batch["frames"][1], "seed": np.array([seed])}
    sched = [True, False, True]
    for it, use_vis in enumerate(sched):
        torch.manual_seed(1000 + it)                  # pins the AO draw in the reference
        rerr, rmatch = ref_main.train_step(rw, clone_batch(), ropt, use_vis)
        torch.manual_seed(1000 + it)
        draws = torch.rand(B) > 0.5
        os_.levels()[-1].fusion.ao_draws = draws
        oerr, omatch, oouts = OS.train_step(ow, clone_batch(), oopt, use_vis, args)
        close(oerr, rerr, 2e-5, f"step{it} err")
        if use_vis:
            close(omatch, rmatch, 2e-5, f"step{it} match")
        out[f"it{it}.err"] = np.array([rerr], dtype=np.float64)
        out[f"it{it}.match"] = np.array([rmatch if rmatch is not None else np.nan])
        out[f"it{it}.draws"] = draws
        out[f"it{it}.pred0"] = oouts["pred_masks"][0].detach()
        out[f"it{it}.pred1"] = oouts["pred_masks"][1].detach()
        for (k, rp), (_, op) in zip(list(rs.named_parameters()) + list(rf.named_parameters()),
                                    list(os_.named_parameters()) + list(of.named_parameters())):
            close(op, rp, 2e-4, f"step{it} param {k}")
            if it == 0:
                close(op.grad, rp.grad, 1e-3, f"step{it} grad {k}")
    # after 3 steps: parameter checksums + a few full tensors
    for pre, net in (("sound", rs), ("frame", rf)):
        for k, p in net.state_dict().items():
            if p.dtype.is_floating_point:
                out[f"final.{pre}.{k}.sum"] = p.double().sum().reshape(1)
                out[f"final.{pre}.{k}.abs"] = p.double().abs().sum().reshape(1)
    out["final.sound.last_w"] = rs.state_dict()["unet_block.up_forward.2.weight"]
    out["final.frame.fc_b"] = rf.state_dict()["fc.bias"]
    # The MixVis step (main.py:150-156 -> forward_avmiximg, :162-192) as SHIPPED: main.py:181 hands PitWrapper the
    # un-stacked [B,1,F,T] weight.  Execute it through the reference's own NetWrapper.forward and record what happens,
    # so that the oracle's repair (oracle/step.py forward_avmiximg: forward_ao's per-target stacking) rests on a fixture
    # and not on a comment.
    import models.audio_net as RA
    torch.manual_seed(seed)
    rs_mv = RA.Unet(fc_dim=2, num_downs=5, ngf=8, fusion_type="MixVis", att_type="sig")
    O.wide_init(rs_mv, torch.Generator().manual_seed(seed))
    margs = make_args(log_freq=0, fusion_type="MixVis")
    ref_main.args = margs
    rw_mv = ref_main.NetWrapper((rs_mv, rf), RC.PitWrapper(F.binary_cross_entropy), RC.BCELoss())
    rw_mv.train()
    exc = None
    try:
        rw_mv.forward(clone_batch(), margs, True)
    except Exception as e:                              # noqa: BLE001 (whatever the reference raises is the datum)
        exc = e
    out["mixvis.raised"] = np.array([exc is not None])
    out["mixvis.exc_type"] = np.frombuffer((type(exc).__name__ if exc is not None else "").encode(), dtype=np.uint8).copy()
    out["mixvis.exc_msg"] = np.frombuffer((str(exc) if exc is not None else "").encode(), dtype=np.uint8).copy()
    print("reference forward_avmiximg as shipped:", type(exc).__name__ if exc is not None else "ran", "-", str(exc)[:200])
    ref_main.args = args
    save("step", out)


def gen_sopp(ref_main, att, saud):
    g = torch.Generator().manual_seed(16)
    out = {}
    B, K, H, W = 3, 8, 4, 6
    aud = [torch.randn(B, K, 2, 2, generator=g) for _ in range(2)]
    mix = torch.randn(B, K, H, W, generator=g).relu()
    sep = [torch.randn(B, K, H, W // 2, generator=g).relu() for _ in range(2)]
    out.update({"aud0": aud[0], "aud1": aud[1], "mix": mix, "sep0": sep[0], "sep1": sep[1]})
    from oracle import sopp as OSP
    for cname, rcls in (("AttModel", att.AttModel), ("MatchAtt", att.MatchAtt)):
        for at in ("cos", "sig"):
            r = rcls(att_type=at)
            o = OSP.AttModule(cname, at)
            tag = f"{cname}.{at}"
            ctx_r, none = r(aud, None, None)
            ctx_o, _ = o(aud, None, None)
            assert none is None
            close(ctx_o, ctx_r, 1e-6, tag + " ao")
            out[tag + ".ao.ctx"] = ctx_r
            ctx_r, (ml, maps) = r(aud, mix, None)
            ctx_o, (mlo, mapso) = o(aud, mix, None)
            close(ctx_o, ctx_r, 1e-5, tag + " infer ctx")
            close(mlo, ml, 1e-5, tag + " infer ml")
            close(mapso, maps, 1e-5, tag + " infer maps")
            out.update({tag + ".infer.ctx": ctx_r, tag + ".infer.match": ml, tag + ".infer.maps": maps})
            rr = r(aud, mix, sep)
            oo = o(aud, mix, sep)
            close(oo[0], rr[0], 1e-5, tag + " train ctx")
            out[tag + ".train.ctx"] = rr[0]
            for i, (a, b) in enumerate(zip(oo[1], rr[1])):
                close(a, b, 1e-5, tag + f" train meta{i}")
                out[tag + f".train.meta{i}"] = b
    # SoP++ U-Net (basis + per-source bottleneck weights)
    ref = saud.Unet(fc_dim=6, num_downs=5, ngf=4, extra_size=6)
    O.wide_init(ref, g)
    ora = O.Unet(fc_dim=6, num_downs=5, ngf=4, extra_size=6)
    assert list(ora.state_dict().keys()) == list(ref.state_dict().keys())
    ora.load_state_dict(ref.state_dict())
    x = torch.randn(2, 1, 64, 64, generator=g)
    ref.train(); ora.train()
    yr, (er,) = ref(x.clone())
    yo, (eo,) = ora(x.clone())
    close(yo, yr, 2e-5, "sopp unet basis")
    close(eo, er, 2e-5, "sopp unet extra")
    out.update({"unet.x": x, "unet.basis": yr, "unet.extra": er})
    for k, p in ref.state_dict().items():
        out[f"unet.w.{k}"] = p
    save("sopp", out)


def synthetic_raw_audio(path, center_t, n):
    """Deterministic stand-in for librosa.load in the dataset fixture (consumes nothing from `random`)."""
    import zlib
    rs = np.random.RandomState(zlib.crc32(("%s|%.6f" % (path, center_t)).encode()) & 0x7fffffff)
    return (rs.rand(n).astype(np.float32) - 0.5) * 2.4          # exceeds +-1 so the clip matters


def gen_dataset(ref_main):
    """The reference's MUSICMixDataset (dataset/music.py, dataset/base.py) on a synthetic 6-column list: which clips,
    centre times, gains (through an audio checksum), frame files, ids and labels it draws per index.  Decoding is
    replaced on both sides by `synthetic_raw_audio` / the list of frame paths, so only the sampling rules are pinned."""
    import json
    import random
    from dataset import MUSICMixDataset as RefDS
    from avsep_amd import dataset as PD
    from avsep_amd.arguments import ArgParser
    rs = random.Random(77)
    rows = []
    for c in PD.MUSIC11_CLASSES:
        for k in range(3):
            vid = "".join(rs.choice("abcdefghijklmnopqrstuvwxyzABCDEFGH0123456789_-") for _ in range(11))
            fps = rs.choice([24.0, 25.0, 29.97, 30.0])
            secs = round(rs.uniform(9.0, 240.0), 3)
            nf = int(secs * fps) - rs.randint(0, 40)
            rows.append([f"./data/audio/{c}/{vid}.wav", f"./data/frames/{c}/{vid}.mp4", str(nf), str(fps), str(secs), c])
    os.makedirs(OUT, exist_ok=True)
    lst = os.path.join(OUT, "dataset_list.csv")
    with open(lst, "w") as f:
        f.write("\n".join(",".join(r) for r in rows) + "\n\nshort\n")     # + rows the reader must skip
    cases = []
    for split, extra, kw in [("train", [], {}), ("val", [], {}), ("train", ["--one_frame"], {"seed": 10}),
                             ("train", ["--rate_dc", "0.3", "--rate_sc", "0.4", "--rate_sv", "1.0"], {}),
                             ("val", [], {"random_sample": True})]:
        argv = ["--num_frames", "3", "--stride_frames", "8", "--train_repeat", "2", "--val_repeat", "3"] + extra
        a = ArgParser().parse_train_arguments(argv)
        params = vars(a)

        class Ref(RefDS):
            def _load_audio_file(self, path, center_t):
                n = int((self.margin * 2 + self.audSec) * self.audRate)
                return synthetic_raw_audio(path, center_t, n), self.audRate

            def _load_frames(self, paths):
                self.seen.append(list(paths))
                return torch.zeros(1)

        class Mine(PD.MUSICMixDataset):
            _load_audio_file = Ref._load_audio_file
            _load_frames = Ref._load_frames
        ref, mine = Ref(lst, dict(params), split=split, **kw), Mine(lst, dict(params), split=split, **kw)
        assert ref.list_samples == mine.list_samples and len(ref) == len(mine)
        items = []
        for index in [0, 1, 5, len(ref) // 2, len(ref) - 1]:
            ref.seen, mine.seen = [], []
            r, m = ref[index], mine[index]
            rec = {"index": index, "infos": [list(i) for i in r["infos"]], "id": r["id"], "class": r["class"].tolist(),
                   "frame_paths": ref.seen, "audio_sum": [float(a.double().sum()) for a in r["audios"]],
                   "audio_abs": [float(a.double().abs().sum()) for a in r["audios"]],
                   "mix_abs": float(r["audio_mix"].double().abs().sum())}
            assert [list(i) for i in m["infos"]] == rec["infos"] and m["id"] == rec["id"] and mine.seen == ref.seen
            for x, y in zip(m["audios"] + [m["audio_mix"]], r["audios"] + [r["audio_mix"]]):
                assert torch.equal(x, torch.as_tensor(y))
            items.append(rec)
        cases.append({"split": split, "argv": argv, "kw": kw, "len": len(ref), "first_rows": ref.list_samples[:3],
                      "items": items})
    with open(os.path.join(OUT, "dataset.json"), "w") as f:
        json.dump({"cases": cases}, f, indent=0)
    print("wrote dataset.json + dataset_list.csv (%d cases)" % len(cases))


if __name__ == "__main__":
    torch.set_num_threads(8)
    ref_main, att, saud = import_reference()
    which = sys.argv[1:] or ["prepare", "fusion", "unet", "criterion", "synth", "step", "sopp", "dataset"]
    for w in which:
        fn = globals()["gen_" + w]
        fn(ref_main, att, saud) if w == "sopp" else fn(ref_main)
    print("golden vectors written to", OUT)
