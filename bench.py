#!/usr/bin/env python
"""bench.py — mixtures/sec of the mix-and-separate AV train step on MI355X (BASELINE.json metric).

Workload at every N (weak scaling): BASELINE.json configs[1] per GPU — 2-source mix, batch 32,
65535-sample waveforms -> HIP STFT (1022/256) -> 512x256 magnitudes -> log-frequency warp to
256x256 tiles, 3 RGB frames at 224^2 per source, fp32; one step = zero_grad + NetWrapper.forward
(AV: visual encoder on 2x3 frames, TWO U-Net passes, fusion, BCE) + backward + SGD(momentum, wd)
(reference main.py:557-569).  Inputs are resident in HBM before the timed region.
The U-Net / fusion / loss / STFT / prepare / SGD run on libavsep_gfx950.so; the ResNet-18 frame
encoder's convolutions run on PyTorch-ROCm/MIOpen (that is what configs[1] names), its BatchNorm / ReLU / residual
glue on this library's channels-last kernels (the "hybrid" backend of models/vision_net.py).

Launch: python bench.py --gpus N --steps K --warmup W   (N>1: under torch.distributed.run).
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch

PEAK_F32_MFMA_TFLOPS = 157.3     # MI355X_MICROARCH.md: Peak FP32 (matrix), dense
BATCH_PER_GPU = 32


def step_args(P):
    a = P.arguments.train_music_args()          # scripts/train_MUSIC.sh, the config of record
    a.stft_pad_mode = "reflect"
    return a


def build(P, dev, seed):
    torch.manual_seed(seed)
    a = step_args(P)
    mb = P.ModelBuilder()
    snd = mb.build_sound(arch=a.arch_sound, fc_dim=a.num_channels, weights="", fusion_type=a.fusion_type,
                         att_type=a.att_type)
    frm = mb.build_frame(arch=a.arch_frame, fc_dim=a.vis_channels, pool_type=a.img_pool, weights="")
    crit_ao, crit_av = mb.build_criterion(a.loss, use_pit=True), mb.build_criterion(a.loss)
    snd, frm = snd.to(dev), frm.to(dev)
    return a, snd, frm, P.NetWrapper((snd, frm), crit_ao, crit_av)


class ConvTimer:
    """HIP-event pairs around every implicit-GEMM conv launch (events are recorded on torch's current
    stream, which is the stream handed to the C ABI), with the algorithmic FLOPs of each launch."""

    def __init__(self, K):
        self.K, self.rec, self.on = K, [], False
        for name in ("fwd", "dgrad", "wgrad", "dgrad_up2x"):
            self._wrap(name)

    def _wrap(self, name):
        orig, timer = getattr(self.K.Conv, name), self

        def wrapped(cv, *a, **kw):
            if not timer.on:
                return orig(cv, *a, **kw)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            out = orig(cv, *a, **kw)
            e1.record()
            flops = 2.0 * cv.N * cv.Ho * cv.Wo * cv.Cout * cv.Cin * cv.KH * cv.KW
            # algorithmic HBM bytes: each of the three operands (input, output/cotangent, weights) touched once
            nbytes = 4.0 * (cv.N * cv.Cin * cv.H * cv.W + cv.N * cv.Cout * cv.Ho * cv.Wo + cv.Cout * cv.Cin * cv.KH * cv.KW)
            mode = "dgrad" if name == "dgrad_up2x" else name
            timer.rec.append((mode, flops, e0, e1, kernel_family(cv, mode), nbytes))
            return out
        setattr(self.K.Conv, name, wrapped)

    def summary(self):
        tot_ms, tot_fl, by, fam = 0.0, 0.0, {}, {}
        for name, fl, e0, e1, family, nbytes in self.rec:
            ms = e0.elapsed_time(e1)
            tot_ms += ms
            tot_fl += fl
            for table, key in ((by, name), (fam, family)):
                b = table.setdefault(key, [0.0, 0.0, 0, 0.0])
                b[0] += ms; b[1] += fl; b[2] += 1; b[3] += nbytes
        return tot_ms, tot_fl, by, fam


def kernel_family(cv, mode):
    """Which HIP kernel a Conv call dispatches to (mirrors the predicates in csrc/conv.hip)."""
    k3 = cv.KH == 3 and cv.KW == 3 and cv.d.stride == 1 and cv.d.pad == 1 and cv.d.dil == 1
    if getattr(cv, "head", False):
        return "head_" + mode + "_kernel"
    if k3 and cv.Cout <= 4 and mode in ("fwd", "wgrad") and cv.W % 16 == 0:
        return "smallco_" + mode
    if k3 and cv.W >= 12 and cv.H >= 4:
        if mode == "fwd" and cv.Cin % 4 == 0 and cv.d.C0 % 4 == 0 and cv.Cout > 4:
            return "conv3x3_kernel"
        if mode == "dgrad" and cv.Cout % 4 == 0 and cv.Cin >= 32:
            return "conv3x3_kernel"
        if mode == "wgrad" and cv.Cout > 4 and cv.Cin >= 32 and cv.W % 2 == 0:
            return "wgrad3x3_kernel"
    if cv.KH == 4 and cv.KW == 4 and cv.d.stride == 2 and cv.d.pad == 1 and cv.Wo >= 16 and cv.Ho >= 4:
        if mode == "fwd" and cv.Cin % 2 == 0 and cv.Cout >= 32:
            return "conv3x3_kernel"           # the same halo-patch kernel, KS = 4 / S = 2 instantiation
        if mode == "dgrad" and cv.Cout % 8 == 0 and cv.Cin >= 32 and cv.H % 2 == 0 and cv.W % 2 == 0:
            return "conv3x3_kernel"           # 4 parity-class launches of the KS = 2 instantiation
    return "igemm_kernel<%s>" % mode


def pmc_traffic(family):
    """PMC-measured HBM traffic of one kernel family, from the committed summary of the rocprofv3 --pmc passes
    (counters cannot be read from inside the process; the summary is regenerated by profiles/summarise_pmc.py)."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "pmc_traffic.json")
    if not os.path.exists(path):
        return None
    with open(path) as f:
        table = json.load(f)
    row = table.get("kernels", {}).get(family)
    if row is None:
        return None
    return {"traffic_bytes_per_step": row.get("traffic_bytes_per_step"),
            "source": "profiles/pmc_traffic.json: " + table.get("command", "")}


def cpu_baseline(P, seed):
    """The oracle (CPU restatement pinned to the reference) timed on this host's cores: AV train step,
    batch 2 (BASELINE configs[0]), 1 warm-up + 3 timed steps (~10-30 s of CPU work)."""
    from oracle import nets as O, step as OS, criterion as OC, stft as OST
    import numpy as np
    a = step_args(P)
    torch.manual_seed(seed)
    snd = O.build_sound(a.arch_sound, a.num_channels, a.fusion_type, a.att_type)
    frm = O.build_frame(a.arch_frame, a.vis_channels, a.img_pool)
    wrap = OS.NetWrapper((snd, frm), OC.build_criterion(a.loss, True), OC.build_criterion(a.loss))
    opt = OS.create_optimizer((snd, frm), a)
    B = 2
    raw = P.synth.make_batch(B, a.num_mix, a.num_frames, 224, a.audLen, seed=seed)

    def batch():   # the reference's loader does the STFT on the CPU too (dataset/base.py:142-147)
        mags = [torch.from_numpy(np.stack([OST.stft_mag_phase(w.numpy())[0] for w in src]))[:, None]
                for src in raw["audios"]]
        mix = torch.from_numpy(np.stack([OST.stft_mag_phase(w.numpy())[0] for w in raw["audio_mix"]]))[:, None]
        return {"mag_mix": mix, "mags": mags, "frames": raw["frames"]}
    times = []
    for it in range(4):
        t0 = time.perf_counter()
        OS.train_step(wrap, batch(), opt, True, a)
        times.append(time.perf_counter() - t0)
    med = sorted(times[1:])[1]
    return {"value": B / med, "unit": "mixtures/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"oracle AV train step (STFT+fwd+bwd+SGD), batch {B}, 1 warm-up + 3 timed steps, median {med:.2f} s/step"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=BATCH_PER_GPU, help="per-GPU batch (configs[1]: 32)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--ao", action="store_true", help="time the audio-only step instead (extra, not the headline)")
    ap.add_argument("--no-extra", action="store_true", help="skip the untimed AO-step / blend extras (profiling runs)")
    o = ap.parse_args()

    import avsep_amd as P
    rank, world, dev = P.dp.init_from_env()
    if world != o.gpus:
        raise SystemExit(f"--gpus {o.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    if dev.type != "cuda":
        raise SystemExit("bench.py needs an MI355X (no CPU fallback for the HIP path)")
    import torch.distributed as dist

    seed = 1234
    a, snd, frm, wrap = build(P, dev, seed)                 # identical replicas: same seed on every rank
    opt = P.create_optimizer((snd, frm), a, world_size=world)
    B = o.batch
    raw = P.synth.make_batch(B, a.num_mix, a.num_frames, 224, a.audLen, seed=seed + 1 + rank, device=dev)
    use_vis = not o.ao

    def batch():
        return {"audios": list(raw["audios"]), "audio_mix": raw["audio_mix"], "frames": list(raw["frames"])}

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    timer = ConvTimer(P.kernels)
    for _ in range(o.warmup):
        P.net_wrapper.train_step_async(wrap, batch(), opt, use_vis, a)
    sync()
    timer.on = True
    t0 = time.perf_counter()
    for _ in range(o.steps):
        err, match, _ = P.net_wrapper.train_step_async(wrap, batch(), opt, use_vis, a)
    sync()
    dt = time.perf_counter() - t0
    timer.on = False
    t = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = t.item()
    conv_ms, conv_fl, by, fam = timer.summary()
    # extra (outside the timed region, not part of `value`): the audio-only step and the 1:1 AV/AO alternation
    # the shipped flags produce (iter_per_av 2: AV on even iterations, scripts/train_MUSIC.sh:10-11,50)
    ao_rate = None
    if not o.ao and world == 1 and not o.no_extra:
        for _ in range(2):
            P.net_wrapper.train_step_async(wrap, batch(), opt, False, a)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(4):
            P.net_wrapper.train_step_async(wrap, batch(), opt, False, a)
        torch.cuda.synchronize()
        ao_rate = 4 * B / (time.perf_counter() - t1)

    if rank == 0:
        ach = conv_fl / (conv_ms * 1e-3) / 1e12 if conv_ms > 0 else 0.0
        dom = max(fam, key=lambda k: fam[k][0])                    # the kernel with the most time per step
        d_ms, d_fl, d_n, d_bytes = fam[dom]
        d_ach = d_fl / (d_ms * 1e-3) / 1e12
        traffic = pmc_traffic(dom)
        out = {
            "metric": "mixtures/sec (train step, 2-src MUSIC shape)", "value": world * B * o.steps / dt,
            "unit": "mixtures/s", "n_gpus": world, "steps": o.steps, "warmup": o.warmup,
            "ms_per_step": dt / o.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": ("AO" if o.ao else "AV") + " train step: 2-source mix, batch %d/GPU, 65535-sample "
                       "waveforms -> STFT 1022/256 -> 256x256 log-freq tiles, 3x224^2 frames/source, unet7+hidsep(sig)+"
                       "resnet18dilated, BCE, SGD; HIP STFT+prepare+U-Net+fusion+loss+SGD, visual convolutions on PyTorch-ROCm/"
                       "MIOpen with HIP BatchNorm/ReLU glue (BASELINE configs[1])" % B,
                       "global_batch": world * B, "parallelism": f"dp{world}"},
            "loss": float(err), "match_loss": float(match) if match is not None else None,
            "extra": None if ao_rate is None else {
                "ao_step_mixtures_per_s": ao_rate,
                "av_ao_1to1_blend_mixtures_per_s": 2.0 / (1.0 / (world * B * o.steps / dt) + 1.0 / ao_rate)},
            # dominant kernel = the HIP kernel with the largest time per step; achieved = its algorithmic FLOPs per
            # launch / its average launch duration (HIP events on the launch stream, inside the timed region)
            "roofline": {"bound": "mfma", "achieved": d_ach, "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                         "frac": d_ach / PEAK_F32_MFMA_TFLOPS,
                         # HBM bytes per launch from rocprofv3 PMC passes of this same command (2*FETCH_SIZE + WRITE_SIZE,
                         # the gfx950 correction of MI355X_MICROARCH.md); measured offline, see profiles/summarise_pmc.py
                         # (per conv call like `achieved`: a 4x4/s2 data gradient is 4 kernel launches, one per parity class)
                         "traffic": (traffic["traffic_bytes_per_step"] / (d_n / max(o.steps, 1))
                                     if traffic and traffic["traffic_bytes_per_step"] else None),
                         "traffic_source": traffic["source"] if traffic else None,
                         "algorithmic_bytes_per_launch": d_bytes / max(d_n, 1), "kernel": dom,
                         "launches_per_step": d_n / max(o.steps, 1), "avg_launch_ms": d_ms / max(d_n, 1),
                         "algorithmic_gflop_per_launch": d_fl / max(d_n, 1) / 1e9},
            "roofline_all_convs": {"achieved": ach, "frac": ach / PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                                   "launches_per_step": len(timer.rec) / max(o.steps, 1),
                                   "kernel_ms_per_step": conv_ms / max(o.steps, 1),
                                   "algorithmic_gflop_per_step": conv_fl / max(o.steps, 1) / 1e9,
                                   "by_kernel": {k: {"ms_per_step": v[0] / o.steps, "tflops": v[1] / (v[0] * 1e-3) / 1e12 if v[0] else 0.0,
                                                     "launches": v[2] // max(o.steps, 1)} for k, v in fam.items()},
                                   "by_mode": {k: {"ms_per_step": v[0] / o.steps, "tflops": v[1] / (v[0] * 1e-3) / 1e12 if v[0] else 0.0,
                                                   "launches": v[2] // max(o.steps, 1)} for k, v in by.items()}},
        }
        if world == 1 and not o.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(P, seed)
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
