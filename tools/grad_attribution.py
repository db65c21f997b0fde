"""Where does the fp32 gradient error of the HIP path come from?  ("What's weak" item 2.)

tests/test_gpu_model.py::_check_flat_grads compares every parameter gradient of a full-size AV step (batch 8, descriptors
planned for the bench batch 64) with the CPU oracle in FLOAT64 and uses the fp32 CPU oracle's own distance from it as the
yardstick: e(x) = ||x - g64|| / ||g64||.  This tool runs the same check several ways in ONE process (the kernel families
are selected per descriptor, kernels.set_algo_mask) and prints, per parameter family, the median and the worst
e(hip) / e(oracle fp32) and the median e(hip):

  shipped        as the step ships (Winograd F(4x4) / F(2x2) + Winograd weight gradients, passes on forked streams)
  no_wino4       F(4x4,3x3) layers back on F(2x2,3x3)
  no_winograd    direct-form forward / data / weight gradients everywhere
  one_stream     as shipped, every pass on one HIP stream (no concurrent fp64 statistics atomics)
  hip_stft       as shipped, but from the waveforms through the HIP STFT (fp32 DFT on the MFMA) instead of the oracle's
                 float64-FFT magnitudes

Usage (GPU box):  python tools/grad_attribution.py [--out gpurun_out/grad_attribution.txt]
The oracle (CPU fp32 + float64, batch 8, full size) takes a few minutes of host time; it is computed once.
"""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def family(name):
    nd = None
    if name.startswith("frame."):
        k = name[len("frame."):]
        if k.startswith("fc."):
            return "trunk fc conv"
        if ".bn" in k or k.startswith("features.1.") or "downsample.1" in k:
            return "trunk BatchNorm"
        layer = k.split(".")[1]
        return {"0": "trunk stem conv", "4": "trunk layer1 conv (56x56)", "5": "trunk layer2 conv (28x28)",
                "6": "trunk layer3 conv (14x14)", "7": "trunk layer4 conv (14x14, dil 2)"}.get(layer, "trunk other")
    k = name[len("sound."):]
    depth = k.count("mid_forward")
    if k.startswith("bn0"):
        return "U-Net bn0"
    if "down_forward" in k:
        return ("U-Net encoder conv d%d" % (depth + 1)) if k.endswith("1.weight") or k.endswith("0.weight") else "U-Net encoder BatchNorm"
    if k.endswith("up_forward.2.weight") or k.endswith("up_forward.2.bias"):
        return "U-Net decoder conv u%d" % (depth + 1)
    return "U-Net decoder BatchNorm"


def run_variant(P, tmod, dev, a, raw, init, ograds, ograds64, fork, hip_stft):
    K = P.kernels
    mb = P.ModelBuilder()
    snd = mb.build_sound(arch=a.arch_sound, fc_dim=a.num_channels, fusion_type=a.fusion_type, att_type=a.att_type)
    frm = mb.build_frame(arch=a.arch_frame, fc_dim=a.vis_channels, pool_type=a.img_pool)
    snd.load_state_dict(init[0]); frm.load_state_dict(init[1])
    snd, frm = snd.to(dev), frm.to(dev)
    wrap = P.NetWrapper((snd, frm), mb.build_criterion(a.loss, use_pit=True), mb.build_criterion(a.loss))
    wrap.fork_sources = snd.fork_pair = fork
    snd.encoder_bwd_on_side = fork
    opt = P.create_optimizer((snd, frm), a)
    if hip_stft:
        gb = {"audios": [w.to(dev) for w in raw["audios"]], "audio_mix": raw["audio_mix"].to(dev), "frames": [f.to(dev) for f in raw["frames"]]}
    else:
        omix, omags = tmod._ORACLE_CACHE["mags"]
        gb = {"mag_mix": omix.to(dev), "mags": [m.to(dev) for m in omags], "frames": [f.to(dev) for f in raw["frames"]]}
    wrap.train()
    opt.zero_grad()
    with K.pack_scope():
        err, _ = wrap.forward(gb, a, True)
        err.mean().backward()
    torch.cuda.synchronize()
    rows = []
    for prefix, net in (("sound.", snd), ("frame.", frm)):
        for k, p in net.named_parameters():
            og = ograds.get(prefix + k)
            if og is None:
                continue
            g = p.grad.detach().double().cpu()
            ref = ograds64[prefix + k]
            e_hip = ((g - ref).norm() / ref.norm().clamp_min(1e-300)).item()
            e_o32 = ((og.double() - ref).norm() / ref.norm().clamp_min(1e-300)).item()
            rows.append((prefix + k, e_hip, e_o32))
    return rows


def summarise(rows):
    fam = {}
    for name, e_hip, e_o32 in rows:
        fam.setdefault(family(name), []).append((e_hip / max(e_o32, 1e-300), e_hip, e_o32, name))
    out = {}
    for f, v in fam.items():
        r = sorted(x[0] for x in v)
        out[f] = (len(v), r[len(r) // 2], r[-1], sorted(x[1] for x in v)[len(v) // 2], sorted(x[2] for x in v)[len(v) // 2])
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=None)
    ap.add_argument("--variants", default="shipped,no_wino4,no_winograd,one_stream,hip_stft")
    o = ap.parse_args()
    import avsep_amd as P
    import test_gpu_model as tmod
    from oracle import nets as O, step as OS, criterion as OC, stft as OST
    K = P.kernels
    dev = torch.device("cuda:0")
    a, raw, init, draws, osteps = tmod._oracle_full_size_b8(P, O, OS, OC, OST, np)
    _, _, _, ograds, ograds64 = osteps[0]
    K.plan_batch_scale = tmod.BENCH_BATCH // tmod.DISPATCH_TEST_BATCH
    lines = []

    def emit(s=""):
        print(s, flush=True)
        lines.append(s)
    emit("# fp32 gradient error against the float64 oracle, full-size AV step at batch %d planned for batch %d" %
         (tmod.DISPATCH_TEST_BATCH, tmod.BENCH_BATCH))
    emit("# per family: n tensors | median e_hip/e_oracle32 | worst e_hip/e_oracle32 | median e_hip | median e_oracle32")
    table = {}
    for v in o.variants.split(","):
        K.set_algo_mask(*{"no_winograd": ("winograd", "winograd_wgrad"), "no_wino4": ("winograd4",)}.get(v, ()))
        rows = run_variant(P, tmod, dev, a, raw, init, ograds, ograds64, fork=(v != "one_stream"), hip_stft=(v == "hip_stft"))
        table[v] = summarise(rows)
        allr = sorted(r[1] / max(r[2], 1e-300) for r in rows)
        emit("\n== %s: %d tensors, median ratio %.2f, worst %.2f, median e_hip %.2e" %
             (v, len(rows), allr[len(allr) // 2], allr[-1], sorted(r[1] for r in rows)[len(rows) // 2]))
        for f in sorted(table[v]):
            n, med, worst, mh, mo = table[v][f]
            emit("   %-36s %3d | %7.2f | %7.2f | %.2e | %.2e" % (f, n, med, worst, mh, mo))
    K.set_algo_mask()
    K.plan_batch_scale = 1
    if o.out:
        os.makedirs(os.path.dirname(os.path.abspath(o.out)), exist_ok=True)
        with open(o.out, "w") as f:
            f.write("\n".join(lines) + "\n")


if __name__ == "__main__":
    main()
