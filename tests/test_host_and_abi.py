"""not-gpu: the C-ABI library loads and exports every symbol include/avsep.h declares; host logic
(flag system, ModelBuilder surface, state_dict keys, loud failure without a GPU)."""
import ctypes
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _pkg():
    import avsep_amd
    return avsep_amd


def test_library_exports_every_declared_symbol():
    P = _pkg()
    hdr = open(os.path.join(ROOT, "include", "avsep.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(avsep_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 41
    lib = ctypes.CDLL(P.lib.LIB_PATH)
    missing = [n for n in sorted(declared) if not hasattr(lib, n)]
    assert not missing, f"declared in avsep.h but not exported: {missing}"
    assert set(P.lib.SIGNATURES) == declared, set(P.lib.SIGNATURES) ^ declared
    L = P.lib.load()
    assert L.avsep_arch() == b"gfx950" and L.avsep_version() >= 100
    assert L.avsep_strerror(-1).startswith(b"invalid argument")


def test_no_cpu_fallback():
    P = _pkg()
    net = P.ModelBuilder().build_sound(arch="unet5", fc_dim=2, fusion_type="hidsep", att_type="sig")
    with pytest.raises(P.lib.AvsepError):
        net(torch.zeros(1, 1, 64, 64), None)
    with pytest.raises(P.lib.AvsepError):
        P.kernels.prepare(torch.zeros(1, 1, 512, 8), torch.zeros(2, 1, 1, 512, 8), 1, 1, 1)
    import inspect
    src = "".join(inspect.getsource(m) for m in (P.kernels, P.net_wrapper, P.models.audio_net, P.models.criterion,
                                                 P.models.fusion_net, P.lib))
    assert "oracle" not in src, "the product path must not reference the oracle"


def test_flag_system_matches_reference_semantics():
    P = _pkg()
    a = P.ArgParser().parse_train_arguments([], verbose=False)
    assert (a.fusion_type, a.num_channels, a.not_pool_vis, a.use_spec, a.load_ckpt) == ("con", 32, True, True, False)
    assert (a.stft_frame, a.stft_hop, a.audLen, a.seed, a.match_weight) == (1022, 256, 65535, 1234, 0.6)
    b = P.arguments.train_music_args()
    assert (b.arch_sound, b.num_channels, b.vis_channels, b.fusion_type, b.att_type) == ("unet7", 2, 256, "hidsep", "sig")
    assert b.not_pool_vis is False and b.one_frame is True and b.lr_steps == [50000, 70000, 90000]
    assert (b.loss, b.weighted_loss, b.binary_mask, b.log_freq, b.num_frames) == ("bce", 1, 1, 1, 3)


def test_model_builder_surface_and_keys():
    P = _pkg()
    from oracle import nets as O
    mb = P.ModelBuilder()
    for arch, downs in (("unet5", 5), ("unet7", 7)):
        net = mb.build_sound(arch=arch, fc_dim=2, fusion_type="hidsep", att_type="sig")
        ref = O.Unet(fc_dim=2, num_downs=downs, fusion_type="hidsep", att_type="sig")
        assert list(net.state_dict().keys()) == list(ref.state_dict().keys())
        assert [tuple(v.shape) for v in net.state_dict().values()] == [tuple(v.shape) for v in ref.state_dict().values()]
    assert sum(p.numel() for p in mb.build_sound(arch="unet7", fc_dim=2, fusion_type="hidsep").parameters()) == 32598916
    w = net.unet_block.down_forward.at(0).weight
    assert abs(w.std().item() - 1e-3) < 2e-4                       # weights_init: Conv ~ N(0, 1e-3)
    frm = mb.build_frame(arch="resnet18dilated", fc_dim=256, pool_type="maxpool")
    assert list(frm.state_dict().keys()) == list(O.VisualNet(256, "maxpool", 16).state_dict().keys())
    assert sum(p.numel() for p in frm.parameters()) == 12356416
    l4 = frm.features[7]
    assert l4[0].conv1.stride == (1, 1) and l4[0].conv1.dilation == (1, 1) and l4[0].conv2.dilation == (2, 2)
    assert l4[0].downsample[0].stride == (1, 1) and l4[1].conv1.padding == (2, 2)
    assert hasattr(frm, "fc") and hasattr(frm, "features")
    for bad in ("build_sound", "build_frame"):
        with pytest.raises(Exception, match="Architecture undefined!"):
            getattr(mb, bad)(arch="nope")
    with pytest.raises(Exception, match="Architecture undefined!"):
        mb.build_criterion("nope")
    with pytest.raises(Exception, match="Unkown activation!"):
        P.activate(torch.zeros(1), "nope")
    with pytest.raises(AssertionError):
        mb.build_sound(arch="unet5", fusion_type="con")             # the default flag value is invalid (S3)
    assert type(mb.build_criterion("l1", use_pit=True)).__name__ == "PitWrapper"
    assert mb.build_synthesizer("linear", fc_dim=8).scale.shape == (8,)


def test_best_permutations_tie_and_order():
    import numpy as np
    from avsep_amd.models.criterion import best_permutations
    m = np.array([[[1.0, 0.0], [0.0, 1.0]], [[0.0, 1.0], [1.0, 0.0]], [[0.5, 0.5], [0.5, 0.5]]])
    assert best_permutations(m) == [(1, 0), (0, 1), (0, 1)]


def test_synth_batch_contract():
    P = _pkg()
    b = P.synth.make_batch(2, 2, 3, 32, aud_len=4096, seed=7)
    assert b["audio_mix"].shape == (2, 4096) and len(b["audios"]) == 2 and b["frames"][0].shape == (2, 3, 3, 32, 32)
    assert torch.allclose(b["audio_mix"], b["audios"][0] + b["audios"][1])
    assert b["audios"][0].abs().max() <= 0.5 + 1e-6                 # clipped to +-1 then divided by N=2
    b2 = P.synth.make_batch(2, 2, 3, 32, aud_len=4096, seed=7)
    assert torch.equal(b["audio_mix"], b2["audio_mix"])


def test_checkpoint_files_match_the_reference_names(tmp_path):
    """main.py:506-533: sound_/frame_/history_ latest + best files, plain state_dicts with the reference's keys;
    the oracle nets (reference key names, pinned by the goldens) load them strictly."""
    import argparse
    import os
    import torch
    import avsep_amd as P
    from oracle import nets as O
    mb = P.ModelBuilder()
    snd = mb.build_sound(arch="unet5", fc_dim=2, fusion_type="hidsep", att_type="sig")
    frm = mb.build_frame(arch="resnet18dilated", fc_dim=16, pool_type="maxpool")
    a = argparse.Namespace(ckpt=str(tmp_path / "ck"), best_err=float("inf"))
    hist = {"val_ao": {"si_sdr": [2.0]}}
    P.checkpoint.checkpoint((snd, frm), hist, 10, a)
    names = sorted(os.listdir(a.ckpt))
    assert names == ["frame_best.pth", "frame_latest.pth", "history_latest.pth", "sound_best.pth", "sound_latest.pth"]
    assert a.best_err == -2.0
    hist["val_ao"]["si_sdr"].append(1.0)                      # worse: best files untouched
    before = os.path.getmtime(os.path.join(a.ckpt, "sound_best.pth"))
    P.checkpoint.checkpoint((snd, frm), hist, 20, a)
    assert os.path.getmtime(os.path.join(a.ckpt, "sound_best.pth")) == before and a.best_err == -2.0
    osnd = O.Unet(fc_dim=2, num_downs=5, ngf=64, fusion_type="hidsep", att_type="sig")
    ofrm = O.VisualNet(fc_dim=16, pool_type="maxpool", dilate_scale=16)
    osnd.load_state_dict(torch.load(os.path.join(a.ckpt, "sound_latest.pth")), strict=True)
    ofrm.load_state_dict(torch.load(os.path.join(a.ckpt, "frame_latest.pth")), strict=True)
    assert torch.load(os.path.join(a.ckpt, "history_latest.pth")) == hist
    assert P.checkpoint.resume_paths(a, best=True) == (os.path.join(a.ckpt, "sound_best.pth"),
                                                         os.path.join(a.ckpt, "frame_best.pth"))
