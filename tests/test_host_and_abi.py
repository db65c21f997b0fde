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


def test_library_reads_no_environment_variable():
    """include/avsep.h: "reads no environment variable".  The shared library must not even import getenv / secure_getenv /
    environ: every dispatch override travels in avsep_conv_desc.algo / .tune (the host layer reads AVSEP_ALGO_NO in Python),
    and the descriptor carries both fields."""
    import subprocess
    P = _pkg()
    out = subprocess.run(["nm", "-D", "--undefined-only", P.lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    bad = [l for l in out.splitlines() if re.search(r"\b(secure_)?getenv\b|\benviron\b", l)]
    assert not bad, bad
    names = [n for n, _ in P.lib.ConvDesc._fields_]
    assert names[22:24] == ["algo", "tune"] and P.lib.ALGO_NO["winograd4"] == 64
    K = P.kernels
    assert K.set_algo_mask("winograd", "flat") == 5 and K.set_algo_mask() == 0


def test_every_entry_point_rejects_null_and_zero_arguments():
    """Error behaviour at the boundary: an int-returning entry point handed null pointers, empty descriptors and zero sizes
    returns AVSEP_ERR_ARG before anything is launched (this runs without a GPU)."""
    import ctypes as C
    P = _pkg()
    L = P.lib.load()
    probed = 0
    for name, (res, args) in sorted(P.lib.SIGNATURES.items()):
        if res is not C.c_int or not args or name in ("avsep_strerror", "avsep_conv2d_head_applicable", "avsep_conv2d_dgrad_act_fused"):
            continue
        vals = []
        for a in args:
            if a is C.c_void_p or a is C.c_char_p:
                vals.append(None)
            elif a is P.lib._CD:
                vals.append(C.byref(P.lib.ConvDesc()))
            elif a is P.lib._KD:
                vals.append(C.byref(P.lib.CatDesc()))
            elif a in (C.c_float, C.c_double):
                vals.append(0.0)
            elif a.__name__.startswith("LP_"):
                vals.append(None)
            else:
                vals.append(0)
        assert getattr(L, name)(*vals) == -1, name
        probed += 1
    assert probed >= 55
    assert L.avsep_conv2d_head_applicable(C.byref(P.lib.ConvDesc())) == 0
    assert L.avsep_conv2d_dgrad_act_fused(C.byref(P.lib.ConvDesc())) == 0
    assert L.avsep_conv_packed_floats(C.byref(P.lib.ConvDesc()), 0) == 0


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


def test_direct_gradient_placement_host_logic():
    """FlatSGD.grad_dest / scratch_dest / fold_scratch + models.audio_net.ParamGrads on CPU tensors (the kernels are stood
    in for by tensor copies): the first contribution of a step lands in the parameter's slot of the flat gradient buffer,
    a second one — same node or another node of the step — goes through the scratch buffer and one add per contiguous
    run, a parameter whose .grad was detached is handed back to autograd, and zero_grad() re-arms everything.  Every
    tensor starts on a 256-byte boundary and the padding stays zero."""
    P = _pkg()
    FlatSGD, ParamGrads = P.net_wrapper.FlatSGD, P.models.audio_net.ParamGrads
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Conv2d(3, 5, 3), torch.nn.BatchNorm2d(5), torch.nn.Conv2d(5, 2, 1))
    params = list(net.parameters())
    opt = FlatSGD([{"params": params, "lr": 0.1, "name": "sound"}], require_gpu=False)
    P.net_wrapper.attach_grad_sink(opt, net)
    A = FlatSGD.ALIGN
    assert all(gv.storage_offset() % A == 0 for _, gv in opt._views) and opt.flat_grad.numel() % A == 0
    g1 = [torch.randn_like(p) for p in params]
    g2 = [torch.randn_like(p) for p in params]

    class Cv:                        # stands in for kernels.Conv: "writes" dw (and db) into the destinations it is handed
        def __init__(self, dw, db):
            self.dw, self.db = dw, db

        def wgrad(self, dy, want_bias=False, out=None, out_bias=None):
            dw = self.dw.clone() if out is None else out.copy_(self.dw)
            db = None
            if want_bias:
                db = self.db.clone() if out_bias is None else out_bias.copy_(self.db)
            return dw, db
    opt.zero_grad()
    node = ParamGrads(net)
    node.wgrad(Cv(g1[0], g1[1]), params[0], None, params[1])       # first contribution: straight into the flat views
    for i in (2, 3, 4, 5):
        node.add(params[i], g1[i].clone())
    node.wgrad(Cv(g2[0], g2[1]), params[0], None, params[1])       # second contribution of the SAME node: scratch + fold
    node.add(params[4], g2[4].clone())
    out = node.finish(params, "sound")
    assert all(o is None for o in out), "every gradient was placed directly"
    exp = [a + b if i in (0, 1, 4) else a for i, (a, b) in enumerate(zip(g1, g2))]
    for (p, gv), e in zip(opt._views, exp):
        assert torch.allclose(p.grad, e) and p.grad.data_ptr() == gv.data_ptr()
    # a second node of the same step (the visual trunk runs once per source): everything through the scratch buffer
    node2 = ParamGrads(net)
    node2.wgrad(Cv(g2[0], g2[1]), params[0], None, params[1])
    for i in (2, 3, 4, 5):
        node2.add(params[i], g2[i].clone())
    assert all(o is None for o in node2.finish(params, "sound"))
    exp2 = [e + b for e, b in zip(exp, g2)]
    for (p, _), e in zip(opt._views, exp2):
        assert torch.allclose(p.grad, e)
    used = torch.zeros_like(opt.flat_grad, dtype=torch.bool)
    for _, gv in opt._views:
        used[gv.storage_offset():gv.storage_offset() + gv.numel()] = True
    assert float(opt.flat_grad[~used].abs().sum()) == 0.0, "padding stays zero"
    # detached .grad (module.zero_grad(set_to_none=True) semantics): the node must hand the gradient back to autograd
    opt.zero_grad()
    params[2].grad = None
    node3 = ParamGrads(net)
    node3.add(params[2], g1[2].clone())
    node3.add(params[3], g1[3].clone())
    out3 = node3.finish(params, "sound")
    assert out3[2] is not None and torch.equal(out3[2], g1[2]) and out3[3] is None
    # no optimizer attached: plain dict semantics
    plain = torch.nn.Conv2d(2, 2, 1)
    n4 = ParamGrads(plain)
    n4.add(plain.weight, torch.ones_like(plain.weight))
    n4.add(plain.weight, torch.ones_like(plain.weight))
    w, b = n4.finish([plain.weight, plain.bias])
    assert torch.equal(w, 2 * torch.ones_like(plain.weight)) and b is None


def test_pack_scope_and_plan_batch_are_scoped():
    """kernels.pack_scope caches packed images only inside the with-block; plan_batch_scale defaults to 1 (plan_n = 0)."""
    P = _pkg()
    K = P.kernels
    assert K._pack_cache is None and K.plan_batch_scale == 1
    with K.pack_scope():
        assert K._pack_cache == {}
        with K.pack_scope():                       # nested scopes share the cache of the outermost
            K._pack_cache["x"] = 1
        assert K._pack_cache == {"x": 1}
    assert K._pack_cache is None
    assert P.lib.ConvDesc().plan_n == 0 and ctypes.sizeof(P.lib.ConvDesc) == 24 * 4 + 6 * 8
    # avsep_act_bwd: eleven pointers in the header's order, then the int32 activation code (padded to the pointer size)
    hdr = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "avsep.h")).read()
    body = hdr[hdr.index("typedef struct avsep_act_bwd {"):hdr.index("} avsep_act_bwd;")]
    names = re.findall(r"(?:const float\*|double\*|int32_t)\s+(\w+);", body)
    assert names == [n for n, _ in P.lib.ActBwd._fields_], names
    assert ctypes.sizeof(P.lib.ActBwd) == 11 * 8 + 8


def test_flat_sgd_state_dict_is_layout_independent_and_reads_older_blobs():
    """FlatSGD.state_dict() carries one momentum tensor per parameter in its logical shape (no trace of ALIGN), and
    load_state_dict() also accepts the two flat layouts earlier builds wrote into optim_latest.pth — unpadded offsets
    (rounds 1-2) and offsets padded to 64 elements (round 3) — and degrades to a history-only resume (warning, zero
    momentum, learning rates by group name) on a blob that matches nothing instead of raising."""
    import warnings
    P = _pkg()
    FlatSGD = P.net_wrapper.FlatSGD
    torch.manual_seed(1)

    def nets():
        torch.manual_seed(2)
        return (torch.nn.Sequential(torch.nn.Conv2d(3, 5, 3), torch.nn.BatchNorm2d(5)), torch.nn.Conv2d(5, 7, 1))

    def make():
        a, b = nets()
        return FlatSGD([{"params": list(a.parameters()), "lr": 0.1, "name": "sound"},
                        {"params": list(b.parameters()), "lr": 0.01, "name": "frame_fc"}], require_gpu=False)
    opt = make()
    params = [p for g in opt.param_groups for p in g["params"]]
    mom = [torch.randn_like(p) for p in params]
    offs, total = opt._offsets(FlatSGD.ALIGN)
    for o, m in zip(offs, mom):
        opt.flat_buf[o:o + m.numel()] = m.reshape(-1)
    opt.param_groups[0]["lr"], opt.param_groups[0]["started"] = 0.05, True
    sd = opt.state_dict()
    assert sd["format"] == 2 and [tuple(m.shape) for m in sd["momentum"]] == [tuple(p.shape) for p in params]
    assert "range" not in sd["groups"][0]
    fresh = make()
    assert fresh.load_state_dict(sd) is True
    assert torch.equal(fresh.flat_buf, opt.flat_buf) and fresh.param_groups[0]["lr"] == 0.05 and fresh.param_groups[0]["started"]
    # older flat blobs: padded to 64 (round 3) and unpadded (rounds 1-2)
    for align in (64, 1):
        src, tot = opt._offsets(align)
        flat = torch.zeros(tot)
        for o, m in zip(src, mom):
            flat[o:o + m.numel()] = m.reshape(-1)
        ranges, off = [], 0
        for g in opt.param_groups:
            beg = off
            off += sum((p.numel() + align - 1) // align * align for p in g["params"])
            ranges.append([beg, off])
        legacy = {"momentum_buffer": flat, "layout": "oihw",
                  "groups": [{"name": g["name"], "lr": 0.5, "started": True, "range": r} for g, r in zip(opt.param_groups, ranges)]}
        fresh = make()
        assert fresh.load_state_dict(legacy) is True, align
        assert torch.equal(fresh.flat_buf, opt.flat_buf), align
        assert fresh.param_groups[1]["lr"] == 0.5
    # a blob of another architecture: warning + history-only resume
    bad = {"momentum_buffer": torch.zeros(17), "groups": [{"name": "sound", "lr": 0.3, "started": True, "range": [0, 17]}]}
    fresh = make()
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        assert fresh.load_state_dict(bad) is False
    assert w and "momentum starts from zero" in str(w[0].message)
    assert float(fresh.flat_buf.abs().sum()) == 0.0 and fresh.param_groups[0]["lr"] == 0.3 and not fresh.param_groups[0].get("started")


def test_pack_scope_key_follows_in_place_weight_updates():
    """kernels.pack_scope caches packed weight images per (weight, geometry); an in-place update of the weight inside a
    scope bumps the tensor's version counter, which is part of the key, so the stale image is not reused (no GPU: the
    packing call itself is replaced by a counter)."""
    P = _pkg()
    K = P.kernels
    calls = []

    class FakeLib:
        def avsep_conv_packed_floats(self, ref, mode):
            return 4
    cv = K.Conv.__new__(K.Conv)
    cv.d = P.lib.ConvDesc()
    cv.ref = None
    w = torch.zeros(4)
    orig_load, orig_call, orig_f32 = K.lib.load, K.call, K._f32
    K.lib.load = lambda: FakeLib()
    K.call = lambda name, *a: calls.append(name)
    K._f32 = lambda shape, like: torch.empty(shape)
    orig_ptr = K.ptr
    K.ptr = lambda t: 0
    try:
        with K.pack_scope():
            a = cv.pack(w, 0)
            assert cv.pack(w, 0) is a and len(calls) == 1
            w.add_(1.0)                                     # e.g. an optimizer step / EMA / clamp inside the scope
            b = cv.pack(w, 0)
            assert b is not a and len(calls) == 2
            assert cv.pack(w.detach(), 0) is b              # detach() shares the version counter
    finally:
        K.lib.load, K.call, K._f32, K.ptr = orig_load, orig_call, orig_f32, orig_ptr


def test_storage_format_helpers_and_conversion_cache_host_logic():
    """kernels.py's format layer without a GPU (the conversion launches are stood in for by torch ops): a B16 image is a
    torch.bfloat16 tensor [N, C/16, H, W, 16] whose dtype is the format tag; dims / channels / per_channel agree for both
    formats; to_b16 / to_f32 remember the converted twin on the tensor, hand the ORIGINAL back when asked to convert the twin
    again, and forget it when an in-place kernel is about to overwrite either side (_drop_twin)."""
    P = _pkg()
    K = P.kernels
    calls = []

    def fake_call(name, *a):
        calls.append(name)
    orig_call, orig_ptr = K.call, K.ptr
    K.call, K.ptr = fake_call, (lambda t: 0)
    try:
        x = torch.randn(2, 32, 5, 7)
        assert K.dims(x) == (2, 32, 5, 7) and K.channels(x) == 32 and K.per_channel(x) == 70 and K.fmt_of(x) == K.FMT_F32
        b = K.to_b16(x)
        assert calls == ["avsep_f32_to_b16"] and b.dtype == torch.bfloat16 and tuple(b.shape) == (2, 2, 5, 7, 16)
        assert K.is_b16(b) and K.dims(b) == (2, 32, 5, 7) and K.channels(b) == 32 and K.per_channel(b) == 70
        assert K.to_b16(x) is b and K.to_f32(b) is x and calls == ["avsep_f32_to_b16"], "twin cache, both directions"
        assert K.as_fmt(x, K.FMT_B16) is b and K.as_fmt(b, K.FMT_F32) is x and K.as_fmt(None, K.FMT_B16) is None
        K._drop_twin(b)                                   # b is about to be overwritten in place
        assert not hasattr(x, "_avsep_twin") and not hasattr(b, "_avsep_twin")
        assert K.to_f32(b) is not x and calls[-1] == "avsep_b16_to_f32"
        x.add_(1.0)                                       # a torch in-place op bumps the version: the twin is stale
        b2 = K.to_b16(x)
        assert b2 is not b and calls[-1] == "avsep_f32_to_b16"
        assert K.b16_ok(64) and not K.b16_ok(170)
        K.set_precision("bf16")
        assert K.want_b16(64) and not K.want_b16(170)
        K.set_precision("f32")
        assert not K.want_b16(64)
    finally:
        K.call, K.ptr = orig_call, orig_ptr
        K.set_precision("f32")
