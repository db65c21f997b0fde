#!/usr/bin/env python
"""bench.py — mixtures/sec of the mix-and-separate AV train step on MI355X (BASELINE.json metric).

Headline (`value`): the FULL-HIP path in fp32 — every convolution of the step (audio U-Net AND the ResNet-18 frame
encoder), STFT, prepare, fusion, loss, BatchNorm glue and SGD on libavsep_gfx950.so; no MIOpen / rocBLAS kernel in
the timed region.  Workload at every N (weak scaling): BASELINE.json configs[2]'s shape per GPU — 2-source mix,
batch 64, 65535-sample waveforms -> HIP STFT (1022/256) -> 512x256 magnitudes -> log-frequency warp to 256x256
tiles, 3 RGB frames at 224^2 per source; one step = zero_grad + NetWrapper.forward (AV: visual encoder on 2x3
frames, TWO U-Net passes over a shared encoder, fusion, BCE) + backward + SGD(momentum, wd) (reference
main.py:557-569).  Inputs are resident in HBM before the timed region.  The reference computes in fp32, so fp32 is
the headline arithmetic; the same run then times, outside the headline's timed region and explicitly labelled:
  * "bf16": configs[2] as BASELINE.json names it — bf16 conv operands AND bf16 channel-blocked activation / gradient images
    in HBM (AVSEP_FMT_B16), fp32 accumulation / BatchNorm statistics / loss / master weights / SGD — with the loss
    difference to the fp32 path;
  * with --compare-miopen only: "f32_miopen_hybrid", round 1's configs[1] path (visual convolutions on PyTorch-ROCm /
    MIOpen through tools/miopen_compare, which patches the trunk of that one model instance), for comparison;
  * the audio-only step and the 1:1 AV/AO alternation the shipped flags produce.

Launch: python bench.py --gpus N --steps K --warmup W.  With N > 1 and no WORLD_SIZE in the environment bench.py
starts its own N ranks (python -m torch.distributed.run on 127.0.0.1, one process per GPU) BEFORE anything touches the
GPU and relays rank 0's line; under an external torch.distributed.run it is a rank.  Prints ONE JSON line on rank 0.

The headline is timed WITHOUT per-kernel instrumentation; a second pass of the same configuration with HIP-event pairs
around every convolution launch gives `roofline` / `by_kernel` (its own ms/step is reported as
`instrumented_ms_per_step`, so the cost of the event pairs is visible).
"""
import argparse
import gc
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch

PEAK_TFLOPS = {"f32": 157.3, "bf16": 2500.0}   # MI355X_MICROARCH.md: dense matrix peaks (f32 MFMA; bf16 MFMA, no sparsity)
PEAK_HBM_GBS = 8000.0                           # HBM3E spec
BATCH_PER_GPU = 64


CONFIG5_BATCH = 32       # BASELINE.json configs[4] / SURVEY 8: per-device batch 32


def step_args(P, config=3):
    a = P.arguments.train_music_args()          # scripts/train_MUSIC.sh, the config of record
    a.stft_pad_mode = "reflect"
    if config == 5:
        # BASELINE.json configs[4]: 3-source mix, 512x256 STFT tiles (no log-frequency warp), 5 frames per source; the
        # N-source rules are build-defined: one logit per source, vis_channels = 512 // 3
        a.log_freq, a.num_mix, a.num_frames, a.num_channels = 0, 3, 5, 3
        a.vis_channels = 512 // 3
    return a


def build(P, dev, seed, backend, config=3):
    torch.manual_seed(seed)
    a = step_args(P, config)
    mb = P.ModelBuilder()
    snd = mb.build_sound(arch=a.arch_sound, fc_dim=a.num_channels, weights="", fusion_type=a.fusion_type,
                         att_type=a.att_type)
    frm = mb.build_frame(arch=a.arch_frame, fc_dim=a.vis_channels, pool_type=a.img_pool, weights="")
    crit_ao, crit_av = mb.build_criterion(a.loss, use_pit=True), mb.build_criterion(a.loss)
    snd, frm = snd.to(dev), frm.to(dev)
    if backend != "hip":                         # comparison runs only: the package itself has the HIP trunk and nothing else
        sys.path.insert(0, os.path.join(ROOT, "tools", "miopen_compare"))
        import backend as miopen_backend
        miopen_backend.install(frm, backend)
    return a, snd, frm, P.NetWrapper((snd, frm), crit_ao, crit_av)


class KernelTimer:
    """HIP-event pairs around every conv launch and every ReLU+bilinear-upsample launch of the step (events are
    recorded on torch's current stream, which is the stream handed to the C ABI), with the algorithmic FLOPs / HBM
    bytes of each launch.  The kernel family of a conv call is asked from the library (avsep_conv_kernel_name)."""

    def __init__(self, K):
        self.K, self.rec, self.on = K, [], False
        for name in ("fwd", "dgrad", "wgrad", "dgrad_up2x", "dgrad_act"):
            self._wrap_conv(name)
        for name in ("fwd", "bwd"):
            self._wrap_cat(name)

    def _time(self, orig, args, kw, family, mode, flops, nbytes, layer=None):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = orig(*args, **kw)
        e1.record()
        self.rec.append((mode, flops, e0, e1, family, nbytes, layer))
        return out

    def _wrap_conv(self, name):
        orig, timer = getattr(self.K.Conv, name), self

        def wrapped(cv, *a, **kw):
            if not timer.on:
                return orig(cv, *a, **kw)
            mode = "dgrad" if name in ("dgrad_up2x", "dgrad_act") else name
            with_stats = bool(len(a) > 2 and a[2] is not None) or kw.get("stats") is not None
            family = "head_dgrad_kernel" if name == "dgrad_up2x" else cv.kernel_name(mode, with_stats)
            flops = 2.0 * cv.N * cv.Ho * cv.Wo * cv.Cout * cv.Cin * cv.KH * cv.KW
            wts = cv.Cout * cv.Cin * cv.KH * cv.KW
            # bytes per activation element: the bf16 kernels read and write B16 images (2 bytes); weights stay fp32 masters
            # (the packed bf16 image is built from them once per step)
            eb = 2.0 if family.startswith(("convbf", "wgradb")) else 4.0
            if family.startswith("head_"):
                # the hi-res 128-channel input is never materialised: low-res sources (+ their gradients) and the logits
                lo = cv.N * cv.Cin * cv.H * cv.W // 4
                nbytes = 4.0 * ((2 * lo if mode == "dgrad" else lo) + cv.N * cv.Cout * cv.Ho * cv.Wo + wts)
            else:   # each of the three operands (input, output / cotangent, weights) touched once
                nbytes = eb * (cv.N * cv.Cin * cv.H * cv.W + cv.N * cv.Cout * cv.Ho * cv.Wo) + 4.0 * wts
            if name == "dgrad_act":
                # the activation gradient in the epilogue (or, for a family without it, as the second launch the call stands
                # for, booked on the conv's family): y, and residual / dz2 / add when given, are read once more
                extra = 1 + sum(kw.get(k) is not None for k in ("residual", "dz2", "add"))
                nbytes += 4.0 * extra * cv.N * cv.Cin * cv.H * cv.W
            layer = (cv.N, cv.Cin, cv.H, cv.W, cv.Cout, cv.KH, cv.d.stride, cv.d.pad, cv.d.dil, cv.d.up2x)
            return timer._time(orig, (cv,) + a, kw, family, mode, flops, nbytes, layer)
        setattr(self.K.Conv, name, wrapped)

    def _wrap_cat(self, name):
        orig, timer = getattr(self.K.Cat, name), self

        def wrapped(cat, *a, **kw):
            if not timer.on:
                return orig(cat, *a, **kw)
            N, C0, C1, H, W = cat.shape
            src = N * ((C0 if not cat.bcast0 else 0) + C1) * H * W + (N * C0 if cat.bcast0 else 0)
            hi = N * (C0 + C1) * 4 * H * W
            # forward: read the sources, write the hi-res tensor; backward: read its gradient (+ the sources for the
            # ReLU mask / BatchNorm sums), write the source gradients
            eb = 2.0 if timer.K.get_precision() == "bf16" else 4.0        # bf16 mode: B16 images on both sides
            nbytes = eb * (src + hi) if name == "fwd" else eb * (hi + 2 * src)
            return timer._time(orig, (cat,) + a, kw, "relu_up2x_" + name, "glue", 0.0, nbytes)
        setattr(self.K.Cat, name, wrapped)

    def summary(self, steps):
        fam = {}
        self.layers = {}
        for mode, fl, e0, e1, family, nbytes, layer in self.rec:
            if layer is not None:
                row = self.layers.setdefault((layer, mode, family), [0.0, 0.0, 0])
                row[0] += e0.elapsed_time(e1) / steps; row[1] += fl / steps; row[2] += 1
            b = fam.setdefault(family, {"ms": 0.0, "flops": 0.0, "n": 0, "bytes": 0.0, "modes": set()})
            b["ms"] += e0.elapsed_time(e1); b["flops"] += fl; b["n"] += 1; b["bytes"] += nbytes; b["modes"].add(mode)
        self.rec = []
        out = {}
        for k, b in fam.items():
            ms = b["ms"] / steps
            out[k] = {"ms_per_step": ms, "launches_per_step": b["n"] / steps,
                      "gflop_per_step": b["flops"] / steps / 1e9,
                      "tflops": b["flops"] / (b["ms"] * 1e-3) / 1e12 if b["ms"] and b["flops"] else 0.0,
                      "algorithmic_gbytes_per_step": b["bytes"] / steps / 1e9,
                      "algorithmic_gb_per_s": b["bytes"] / (b["ms"] * 1e-3) / 1e9 if b["ms"] else 0.0}
        return out


def pmc_traffic(family, prec="f32", config=3):
    """PMC-measured HBM traffic of one kernel family, from the committed summary of the rocprofv3 --pmc passes
    (counters cannot be read from inside the process; the summary is regenerated by profiles/summarise_pmc.py)."""
    name = "pmc_traffic.json" if prec == "f32" else "pmc_traffic_%s.json" % prec
    if config == 5:
        name = name.replace(".json", "_config5.json")
    path = os.path.join(ROOT, "profiles", name)
    if not os.path.exists(path):
        return None
    with open(path) as f:
        table = json.load(f)
    row = table.get("kernels", {}).get(family)
    if row is None:
        return None
    return {"traffic_bytes_per_step": row.get("traffic_bytes_per_step"), "batch": table.get("batch"),
            "source": "profiles/%s: %s" % (os.path.basename(path), table.get("command", ""))}


def physical_cores():
    """Physical cores among the CPUs this process may run on (hyper-thread siblings counted once)."""
    allowed = os.sched_getaffinity(0) if hasattr(os, "sched_getaffinity") else set(range(os.cpu_count() or 1))
    cores, cur = set(), {}
    try:
        with open("/proc/cpuinfo") as f:
            for line in f.read().split("\n") + [""]:
                if ":" in line:
                    k, v = [t.strip() for t in line.split(":", 1)]
                    cur[k] = v
                elif cur:
                    if int(cur.get("processor", -1)) in allowed:
                        cores.add((cur.get("physical id", "0"), cur.get("core id", cur.get("processor"))))
                    cur = {}
    except OSError:
        pass
    return max(1, len(cores) or len(allowed))


def cpu_baseline(P, seed):
    """The oracle (CPU restatement pinned to the reference) timed on this host's cores: batch 2 (BASELINE configs[0]),
    AV and AO train steps separately, 1 warm-up + 5 timed steps each, median; torch threads = physical cores."""
    from oracle import nets as O, step as OS, criterion as OC, stft as OST
    import numpy as np
    cores = physical_cores()
    torch.set_num_threads(cores)
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

    def median_step(use_vis):
        times = []
        for _ in range(6):
            t0 = time.perf_counter()
            OS.train_step(wrap, batch(), opt, use_vis, a)
            times.append(time.perf_counter() - t0)
        return sorted(times[1:])[2]
    av, ao = median_step(True), median_step(False)
    return {"value": B / av, "unit": "mixtures/s", "cores": cores, "kind": "port",
            "ao_value": B / ao, "av_ao_1to1_blend": 2.0 / (av / B + ao / B),
            "sample": f"oracle train step (STFT+fwd+bwd+SGD) on {cores} threads (= physical cores), batch {B}, 1 warm-up + 5 timed "
                      f"steps each, medians: AV {av:.2f} s/step, AO {ao:.2f} s/step"}


def self_launch(o):
    """--gpus N > 1 without a launcher: start N ranks of this script under torch.distributed.run (one process per GPU,
    rendezvous on 127.0.0.1) and relay their output.  This parent never initialises the GPU (importing torch does not),
    so nothing is re-executed from a process that has touched it; the children are ordinary child processes."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    # dmabuf IPC: this pool's host driver supports no legacy IPC handles (the image exports the variable already; a
    # launcher that builds its own env must keep it — cross-process RCCL cannot be exercised on a one-GPU box)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // o.gpus)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={o.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    for line in proc.stdout:                                 # rank 0's JSON line (and nothing else) arrives on stdout
        sys.stdout.write(line)
        sys.stdout.flush()
    return proc.wait()


def rehearse(o):
    """--rehearse: launcher + rendezvous + barrier / MAX-over-ranks timing + JSON protocol of the N > 1 path with the train
    step replaced by its ONLY collective — the all-reduce of one flat fp32 buffer of the step's gradient size (44,955,332
    elements) — so the path can be run on CPU ranks over gloo (tests/test_bench_launcher.py) or on ranks
    sharing one GPU.  It measures nothing about the kernels and says so in the line it prints."""
    import torch.distributed as dist
    import avsep_amd as P
    rank, world, dev = P.dp.init_from_env()
    n = 44955332 if o.rehearse_elems <= 0 else o.rehearse_elems
    flat = torch.full((n,), float(rank + 1), dtype=torch.float32, device=dev)

    def sync():
        if dev.type == "cuda":
            torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
    for _ in range(max(o.warmup, 1)):
        if world > 1:
            dist.all_reduce(flat)
    sync()
    t0 = time.perf_counter()
    for _ in range(o.steps):
        if world > 1:
            dist.all_reduce(flat)
    sync()
    t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    if rank == 0:
        print(json.dumps({"metric": "mixtures/sec (train step, 2-src MUSIC shape)", "value": None, "unit": "mixtures/s",
                          "rehearsal": True, "n_gpus": o.gpus, "n_ranks_seen": dist.get_world_size() if world > 1 else 1,
                          "backend": dist.get_backend() if world > 1 else None, "device": dev.type,
                          "steps": o.steps, "warmup": o.warmup, "allreduce_bytes_per_step": 4 * n,
                          "allreduce_ms": t.item() / o.steps * 1e3, "scaling": "weak",
                          "note": "launcher / rendezvous / collective rehearsal only: no train step was run"}))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def run_config(P, dev, world, seed, rank, prec, backend, B, steps, warmup, timer=None, use_vis=True, config=3, force=False):
    """Build the model from `seed`, run `warmup` untimed + `steps` timed train steps; returns the measurements."""
    import torch.distributed as dist
    P.kernels.set_precision(prec)
    a, snd, frm, wrap = build(P, dev, seed, backend, config)
    if timer is not None:
        # the instrumented pass prices KERNELS (roofline, by_kernel): every pass on one stream, so that an event pair brackets
        # one kernel alone on the chip.  The headline pass runs the step as shipped: the visual trunk's per-source passes and
        # the U-Net's two decoder passes on forked HIP streams (NetWrapper._frame_features, audio_net._UnetPairFn).
        wrap.fork_sources = snd.fork_pair = False
    opt = P.create_optimizer((snd, frm), a, world_size=world, force_collective=force)
    raw = P.synth.make_batch(B, a.num_mix, a.num_frames, 224, a.audLen, seed=seed + 1 + rank, device=dev)

    def batch():
        return {"audios": list(raw["audios"]), "audio_mix": raw["audio_mix"], "frames": list(raw["frames"])}

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()
    first = None
    for _ in range(max(warmup, 1)):          # at least one untimed step: its loss (identical init) compares precisions
        err, match, _ = P.net_wrapper.train_step_async(wrap, batch(), opt, use_vis, a)
        if first is None:
            first = (float(err), float(match) if match is not None else None)
    sync()
    if timer is not None:
        timer.on = True
    t0 = time.perf_counter()
    for _ in range(steps):
        err, match, _ = P.net_wrapper.train_step_async(wrap, batch(), opt, use_vis, a)
    sync()
    dt = time.perf_counter() - t0
    if timer is not None:
        timer.on = False
    t = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = t.item()
    active = [g for g in opt.param_groups if use_vis or g["name"] == "sound"]
    res = {"value": world * B * steps / dt, "ms_per_step": dt / steps * 1e3, "loss": float(err),
           "allreduce_bytes_per_step": 4 * sum(g["range"][1] - g["range"][0] for g in active) if (world > 1 or force) else 0,
           "early_allreduces": opt.early_reductions,
           "match_loss": float(match) if match is not None else None,
           "first_step_loss": first[0] if first else None, "first_step_match_loss": first[1] if first else None}
    if force:            # the step's one buffer through RCCL on its own (the collective's cost without overlap)
        sync()
        t1 = time.perf_counter()
        for _ in range(10):
            dist.all_reduce(opt.flat_grad)
        sync()
        res["allreduce_ms"] = (time.perf_counter() - t1) * 100.0
    del wrap, opt, snd, frm, raw
    gc.collect()
    torch.cuda.empty_cache()
    P.kernels.set_precision("f32")
    return res


def sdr_on_synthetic_val(P, dev, seed, steps=300, batch=16, val_batches=4, precisions=("f32", "bf16", "f32_reseed"), log=None):
    """"SDR on val", the second half of BASELINE.json's metric, on synthetic sources (the MUSIC media are not
    in the image), and the evidence that bf16 mode TRAINS: the full-size model (unet7 + hidsep(sig) + resnet18dilated, reference
    initialisation, the shipped flags of scripts/train_MUSIC.sh) is trained from the same seed in fp32 and in bf16 mode for
    `steps` train steps at batch `batch` on a seeded stream of fresh synthetic mixtures (synth.make_batch_on_device: harmonic
    tones; a source's frames encode its f0), with the schedule the shipped flags produce (AV step on even iterations, audio-only
    on odd ones: main.py:578-581), and evaluated before and after on a HELD-OUT seeded validation set with the reference's
    evaluate() protocol (main.py:421-503: eval mode, AV and audio-only, loss + get_metrics' SI-SDR / SDR / SIR / SAR — here
    evaluate.calc_metrics with the BSS-eval kernels).  "f32_reseed" is the yardstick for the bf16 - f32 differences: the fp32
    run once more with another weight-initialisation seed (same data stream, same validation set) — how far two equally
    valid fp32 trainings of this length end up from one another.  Outside every timed region."""
    import copy
    from avsep_amd.train import av_ao_schedule
    a0 = step_args(P)
    val = [P.synth.make_batch_on_device(batch, a0.num_mix, a0.num_frames, 224, a0.audLen, seed=seed + 100000 + i, device=dev)
           for i in range(val_batches)]

    def evaluate(wrap, a):
        out = {}
        wrap.eval()
        with torch.no_grad():
            for use_vis in (True, False):
                tot = torch.zeros(5, dtype=torch.float64, device=dev)
                for vb in val:
                    b = {"audios": list(vb["audios"]), "audio_mix": vb["audio_mix"], "frames": list(vb["frames"])}
                    err, outputs = wrap.forward(b, a, use_vis)
                    m = P.evaluate.calc_metrics(b, outputs, a, wrap.stft_plan, bss=True)
                    tot += torch.stack([err.mean().double(), m["si_sdr_mean"].double(), m["sdr_mean"].double(),
                                        m["sir_mean"].double(), m["sar_mean"].double()])
                loss, si_sdr, sdr, sir, sar = (tot / len(val)).tolist()
                out["val_av" if use_vis else "val_ao"] = {"loss": loss, "si_sdr": si_sdr, "sdr": sdr, "sir": sir, "sar": sar}
        torch.set_grad_enabled(True)
        return out

    res = {"steps": steps, "batch": batch, "val_mixtures": batch * val_batches,
           "data": "synthetic harmonic-tone mixtures, fresh seeded batch per step; held-out seeded validation set; frames encode f0",
           "schedule": "AV on even iterations, audio-only on odd ones (iter_per_av 2, start_av_first, num_fsteps 0)"}
    for prec in precisions:
        P.kernels.set_precision("f32" if prec == "f32_reseed" else prec)
        try:
            a, snd, frm, wrap = build(P, dev, seed + (977 if prec == "f32_reseed" else 0), "hip")
            a = copy.copy(a)
            opt = P.create_optimizer((snd, frm), a)
            r = {"before": evaluate(wrap, a)}
            curve = {"av": [], "ao": []}
            errs = []
            for i in range(steps):
                use_vis = av_ao_schedule(i, a)
                tb = P.synth.make_batch_on_device(batch, a.num_mix, a.num_frames, 224, a.audLen, seed=seed + 1 + i, device=dev)
                b = {"audios": list(tb["audios"]), "audio_mix": tb["audio_mix"], "frames": list(tb["frames"])}
                err, match, _ = P.net_wrapper.train_step_async(wrap, b, opt, use_vis, a)
                errs.append((use_vis, err if match is None else err - a.match_weight * match))      # like main.py:718
            window = max(2, steps // 10)
            for use_vis, e in errs:
                curve["av" if use_vis else "ao"].append(float(e))
            for k, v in curve.items():
                if v:
                    w = min(window // 2 or 1, len(v))
                    r["train_loss_" + k] = {"first": sum(v[:w]) / w, "last": sum(v[-w:]) / w, "window": w}
            r["after"] = evaluate(wrap, a)
            res[prec] = r
            if log is not None:
                log("sdr-on-synthetic-val %s: %s" % (prec, json.dumps(r)))
            del wrap, opt, snd, frm
            gc.collect()
            torch.cuda.empty_cache()
        finally:
            P.kernels.set_precision("f32")
    if "f32" in res and "bf16" in res:
        f, b = res["f32"]["after"], res["bf16"]["after"]
        res["bf16_minus_f32_after"] = {k: {m: b[k][m] - f[k][m] for m in ("loss", "si_sdr", "sdr")} for k in ("val_av", "val_ao")}
    if "f32" in res and "f32_reseed" in res:
        f, b = res["f32"]["after"], res["f32_reseed"]["after"]
        res["f32_reseed_minus_f32_after"] = {k: {m: b[k][m] - f[k][m] for m in ("loss", "si_sdr", "sdr")} for k in ("val_av", "val_ao")}
    return res


def config5_line(P, dev, world, seed, rank, o, timer):
    """BASELINE.json configs[4] on one GPU: 3-source mix, 5 frames per source, 512x256 tiles, batch 32 — the workload
    that stresses the fusion / mask head (C! = 6 permutations in the N-source fusion kernel, 15 frames per mixture
    through the visual trunk, 3 logits).  Same measurement protocol as the headline; roofline on the executed FLOPs."""
    B5 = min(o.batch, CONFIG5_BATCH)
    steps = max(3, o.steps // 2)
    r = run_config(P, dev, world, seed, rank, o.precision, "hip", B5, steps, 2, config=5)
    r_inst = run_config(P, dev, world, seed, rank, o.precision, "hip", B5, steps, 1, timer, config=5)
    k5 = timer.summary(steps)
    roof5, step5, _ = roofline_of(k5, o.precision, r["ms_per_step"], B5)
    add_traffic(roof5, o.precision, B5, config=5)
    r.update({"dtype": o.precision, "batch": B5, "roofline": roof5, "roofline_step": step5,
              "instrumented_ms_per_step": r_inst["ms_per_step"],
              "gflop_per_mixture_executed": step5["executed_gflop_per_step"] / B5,
              "gflop_per_mixture_direct_form": step5["algorithmic_gflop_per_step"] / B5,
              "workload": "BASELINE configs[4] per GPU: 3-source mix, batch %d, 512x256 tiles (log_freq 0), 5x224^2 frames/source, "
                          "unet7 + hidsep(sig) N-source fusion kernel (3! permutations) + resnet18dilated (vis_channels 170), "
                          "BCE, SGD; SURVEY 8(d) prices it at 558 GFLOP/mixture as written in the reference (two full U-Net "
                          "passes); the shared encoder executes less" % B5,
              "by_kernel": {k: {"ms_per_step": round(v["ms_per_step"], 3), "tflops": round(v["tflops"], 1)} for k, v in k5.items()}})
    return r


def add_traffic(roof, prec, B, config=3):
    traffic = pmc_traffic(roof["kernel"], prec, config)
    if traffic and traffic["traffic_bytes_per_step"]:
        scale = B / float(traffic["batch"] or B)                   # PMC passes may run at another batch: bytes scale with it
        roof["traffic"] = traffic["traffic_bytes_per_step"] * scale / roof["launches_per_step"]
        roof["traffic_source"] = traffic["source"]
    return roof


# F(2x2, 3x3): 16 products per 2x2 tile instead of 36; F(4x4, 3x3): 36 per 4x4 tile instead of 144
WINOGRAD_EXECUTED = {"wino_kernel": 4.0 / 9.0, "winow_kernel": 4.0 / 9.0, "wino4_kernel": 1.0 / 4.0, "winow4_kernel": 1.0 / 4.0}


def roofline_of(kernels, prec, step_ms, B_total_per_gpu):
    """dominant MFMA kernel + whole-step fraction + the HBM-bound families, from a KernelTimer summary."""
    vec = ("head_", "smallco", "smallci_dgrad")          # conv families that run on the vector ALU (M or K too small for an MFMA tile)
    mfma = {k: v for k, v in kernels.items() if v["gflop_per_step"] > 0 and not k.startswith(vec)}
    dom = max(mfma, key=lambda k: mfma[k]["ms_per_step"])
    d = mfma[dom]
    # a kernel family computes in bf16 only if it is one of the bf16 kernels; the rest of a bf16 step is exact f32
    peak = PEAK_TFLOPS["bf16"] if dom.startswith(("convbf", "wgradb")) else PEAK_TFLOPS["f32"]
    tot_fl = sum(v["gflop_per_step"] for v in kernels.values())
    tot_ms = sum(v["ms_per_step"] for k, v in kernels.items() if v["gflop_per_step"] > 0)
    # Winograd families execute 16 multiplies per 2x2 tile and channel pair where the direct form has 36: the MFMA roofline
    # is priced on the EXECUTED flops (4/9 of the direct-form count); the direct-form rate is reported beside it
    executed = WINOGRAD_EXECUTED.get(dom, 1.0)
    roof = {"bound": "mfma", "achieved": d["tflops"] * executed, "peak": peak, "unit": "TFLOP/s",
            "frac": d["tflops"] * executed / peak, "direct_form_tflops": d["tflops"], "executed_over_direct_flops": executed,
            "traffic": None, "kernel": dom, "launches_per_step": d["launches_per_step"],
            "avg_launch_ms": d["ms_per_step"] / d["launches_per_step"],
            "algorithmic_gflop_per_launch": d["gflop_per_step"] / d["launches_per_step"],
            "algorithmic_bytes_per_launch": d["algorithmic_gbytes_per_step"] * 1e9 / d["launches_per_step"]}
    exe_fl = sum(v["gflop_per_step"] * WINOGRAD_EXECUTED.get(k, 1.0) for k, v in kernels.items())
    step = {"algorithmic_gflop_per_step": tot_fl, "executed_gflop_per_step": exe_fl, "ms_per_step": step_ms,
            "achieved": exe_fl / step_ms,   # GFLOP/ms = TFLOP/s, executed (Winograd layers at 4/9 of their direct-form count)
            "direct_form_tflops": tot_fl / step_ms,
            "peak": PEAK_TFLOPS[prec], "unit": "TFLOP/s", "frac": exe_fl / step_ms / PEAK_TFLOPS[prec],
            "conv_kernel_ms_per_step": tot_ms, "all_convs_tflops": tot_fl / tot_ms if tot_ms else 0.0}
    hbm = {k: {"ms_per_step": v["ms_per_step"], "achieved": v["algorithmic_gb_per_s"], "peak": PEAK_HBM_GBS, "unit": "GB/s",
               "frac": v["algorithmic_gb_per_s"] / PEAK_HBM_GBS, "bound": "hbm"}
           for k, v in kernels.items() if k.startswith(vec + ("relu_up2x",))}
    return roof, step, hbm


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=BATCH_PER_GPU, help="per-GPU batch (configs[2]: 64)")
    ap.add_argument("--precision", default="f32", choices=["f32", "bf16"], help="arithmetic of the HEADLINE run")
    ap.add_argument("--backend", default="hip", choices=["hip"], help="visual trunk of the headline run (this library)")
    ap.add_argument("--compare-miopen", action="store_true", help="also time the step with the visual convolutions on MIOpen "
                    "(tools/miopen_compare; comparison only, never the headline)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="headline only (profiling runs)")
    ap.add_argument("--no-instrumented", action="store_true", help="profiling runs under rocprofv3: only the un-instrumented "
                    "headline pass (no HIP-event pairs, no per-kernel tables in the line)")
    ap.add_argument("--layers", default=None, help="write a per-convolution-call table of the headline step to this file")
    ap.add_argument("--config", type=int, default=3, choices=[3, 5], help="3: BASELINE configs[2] shape (headline); 5: configs[4] "
                    "(3 sources, 5 frames, 512x256, batch 32) as the headline workload of this run")
    ap.add_argument("--rehearse", action="store_true", help="launcher / rendezvous / all-reduce rehearsal without a train step "
                    "(runs on CPU ranks over gloo too); prints a line marked \"rehearsal\": true")
    ap.add_argument("--rehearse-elems", type=int, default=0, help="elements of the rehearsal's flat buffer (default: the step's)")
    ap.add_argument("--sdr-steps", type=int, default=300, help="train steps of the 'SDR on synthetic val' leg (extra.sdr_on_synthetic_val; "
                    "fp32 and bf16 from the same seed, batch 16; 0 = skip)")
    ap.add_argument("--sdr-only", action="store_true", help="run only the 'SDR on synthetic val' leg and print its JSON")
    ap.add_argument("--force-collective", action="store_true", help="--gpus 1 only: create a 1-rank RCCL group and issue the "
                    "step's early + late gradient all-reduce anyway; the line carries allreduce_ms and the no-collective time")
    o = ap.parse_args()

    if o.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(o))                 # before anything touches the GPU
    if o.rehearse:
        return rehearse(o)

    import avsep_amd as P
    if o.force_collective:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29541")
    rank, world, dev = P.dp.init_from_env(force=o.force_collective)
    if world != o.gpus:
        raise SystemExit(f"--gpus {o.gpus} but WORLD_SIZE={world}")
    if dev.type != "cuda":
        raise SystemExit("bench.py needs an MI355X (no CPU fallback for the HIP path)")
    import torch.distributed as dist

    seed, B = 1234, o.batch
    if o.config == 5:
        B = min(B, CONFIG5_BATCH)
    if o.sdr_only:
        print(json.dumps({"sdr_on_synthetic_val": sdr_on_synthetic_val(P, dev, seed, steps=o.sdr_steps,
                                                                        log=lambda s: print(s, file=sys.stderr, flush=True))}))
        return
    if o.force_collective:
        # RCCL readiness line (one GPU): the same step with and without the collectives of the N > 1 path
        plain = run_config(P, dev, world, seed, rank, o.precision, o.backend, B, o.steps, o.warmup, config=o.config)
        forced = run_config(P, dev, world, seed, rank, o.precision, o.backend, B, o.steps, o.warmup, config=o.config, force=True)
        print(json.dumps({"metric": "mixtures/sec (train step, 2-src MUSIC shape)", "value": forced["value"], "unit": "mixtures/s",
                          "n_gpus": 1, "steps": o.steps, "warmup": o.warmup, "ms_per_step": forced["ms_per_step"],
                          "ms_per_step_no_collective": plain["ms_per_step"], "allreduce_ms": forced["allreduce_ms"],
                          "allreduce_bytes_per_step": forced["allreduce_bytes_per_step"],
                          "early_allreduces": forced["early_allreduces"], "backend": dist.get_backend(),
                          "loss": forced["loss"], "loss_no_collective": plain["loss"], "dtype": o.precision,
                          "config": {"workload": "headline step at batch %d with a 1-rank RCCL group: early U-Net all-reduce + late "
                                                 "visual all-reduce forced (FlatSGD force_collective)" % B}}))
        dist.barrier()
        dist.destroy_process_group()
        return
    # headline: un-instrumented.  Then the same configuration once more with HIP-event pairs around every conv launch
    head = run_config(P, dev, world, seed, rank, o.precision, o.backend, B, o.steps, o.warmup, config=o.config)
    if o.no_instrumented:
        if rank == 0:
            print(json.dumps({"metric": "mixtures/sec (train step, 2-src MUSIC shape)", "value": head["value"], "unit": "mixtures/s",
                              "n_gpus": world, "steps": o.steps, "warmup": o.warmup, "ms_per_step": head["ms_per_step"],
                              "dtype": o.precision, "config": {"workload": "headline pass only (--no-instrumented), batch %d" % B},
                              "loss": head["loss"]}))
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return
    timer = KernelTimer(P.kernels)
    inst = run_config(P, dev, world, seed, rank, o.precision, o.backend, B, o.steps, min(o.warmup, 1), timer, config=o.config)
    kernels = timer.summary(o.steps)
    if o.layers and rank == 0:
        rows = sorted(timer.layers.items(), key=lambda kv: -kv[1][0])
        with open(o.layers, "w") as f:
            f.write("# %s step, batch %d: (N Cin H W Cout K stride pad dil up2x) mode family calls/step ms/step TFLOP/s\n" % (o.precision, B))
            for (layer, mode, family), (ms, fl, n) in rows:
                f.write("%-44s %-6s %-62s %5.1f %8.3f %7.1f\n" % (" ".join(map(str, layer)), mode, family[:62], n / o.steps, ms,
                                                                   fl / (ms * 1e-3) / 1e12 if ms else 0.0))
    extras = {}
    if world == 1 and not o.no_extra:
        other = "bf16" if o.precision == "f32" else "f32"
        r = run_config(P, dev, world, seed, rank, other, "hip", B, o.steps, o.warmup, config=o.config)
        r_inst = run_config(P, dev, world, seed, rank, other, "hip", B, o.steps, 1, timer, config=o.config)
        k2 = timer.summary(o.steps)
        roof2, step2, _ = roofline_of(k2, other, r["ms_per_step"], B)
        add_traffic(roof2, other, B)
        r.update({"dtype": other, "roofline": roof2, "roofline_step": step2, "instrumented_ms_per_step": r_inst["ms_per_step"],
                  "first_step_loss_abs_diff_vs_headline": abs(r["first_step_loss"] - head["first_step_loss"]),
                  "workload": "same step in bf16 mode: bf16 conv operands and bf16 channel-blocked activation / gradient images in HBM, "
                              "fp32 accumulate / BatchNorm statistics / loss / master weights / SGD (BASELINE configs[2])"
                              if other == "bf16" else "same step in fp32",
                  "by_kernel": {k: {"ms_per_step": round(v["ms_per_step"], 3), "tflops": round(v["tflops"], 1)} for k, v in k2.items()}})
        extras[other] = r
        if o.compare_miopen:
            hyb = run_config(P, dev, world, seed, rank, "f32", "hybrid", B, max(3, o.steps // 2), 2, config=o.config)
            extras["f32_miopen_hybrid"] = dict(hyb, workload="round-1 configs[1] path at the headline batch %d: visual convolutions "
                                               "on PyTorch-ROCm/MIOpen, HIP BatchNorm glue; comparison only, not this build's kernels" % B)
        ao = run_config(P, dev, world, seed, rank, o.precision, o.backend, B, max(3, o.steps // 2), 2, use_vis=False, config=o.config)
        extras["ao_step_mixtures_per_s"] = ao["value"]
        extras["av_ao_1to1_blend_mixtures_per_s"] = 2.0 / (1.0 / head["value"] + 1.0 / ao["value"])
        if o.config != 5:
            extras["config5"] = config5_line(P, dev, world, seed, rank, o, timer)
        if o.sdr_steps > 0 and o.config != 5:
            extras["sdr_on_synthetic_val"] = sdr_on_synthetic_val(P, dev, seed, steps=o.sdr_steps)

    if rank == 0:
        roof, roof_step, hbm = roofline_of(kernels, o.precision, head["ms_per_step"], B)
        add_traffic(roof, o.precision, B)
        vis = "visual trunk on this library (no MIOpen kernel in the step)"
        out = {
            "metric": "mixtures/sec (train step, 2-src MUSIC shape)", "value": head["value"],
            "unit": "mixtures/s", "n_gpus": world, "steps": o.steps, "warmup": o.warmup,
            "ms_per_step": head["ms_per_step"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "n_ranks_seen": dist.get_world_size() if world > 1 else 1,
            "allreduce_bytes_per_step": head["allreduce_bytes_per_step"],      # ONE flat fp32 gradient all-reduce per step and rank
            # the pass `roofline` / `by_kernel` come from: event pairs around every conv, every pass of the step on ONE stream
            # (kernel durations exclusive); the headline runs the independent passes on forked streams
            "instrumented_ms_per_step": inst["ms_per_step"],
            "streams": "headline: visual trunk per-source passes + the two decoder passes on forked HIP streams; instrumented: one stream",
            "dtype": o.precision, "data": "synthetic",
            "config": {"workload": ("full HIP path, AV train step: 2-source mix, batch %d/GPU (BASELINE configs[2] shape), 65535-sample "
                                    "waveforms -> HIP STFT 1022/256 -> 256x256 log-freq tiles, 3x224^2 frames/source, unet7+hidsep(sig)+"
                                    "resnet18dilated, BCE, SGD; %s arithmetic; %s" if o.config != 5 else
                                    "full HIP path, AV train step: 3-source mix, batch %d/GPU (BASELINE configs[4]), 512x256 tiles, "
                                    "5x224^2 frames/source, unet7+hidsep(sig) N-source fusion+resnet18dilated; %s arithmetic; %s")
                       % (B, o.precision, vis),
                       "global_batch": world * B, "parallelism": f"dp{world}"},
            "loss": head["loss"], "match_loss": head["match_loss"], "first_step_loss": head["first_step_loss"],
            # dominant kernel = the MFMA kernel family with the largest time per step; achieved = its algorithmic FLOPs /
            # its HIP-event time (events on the launch stream, inside the timed region)
            "roofline": roof,
            # whole step: algorithmic conv FLOPs executed per step / wall time per step / peak
            "roofline_step": roof_step,
            # the VALU / HBM-bound kernels of the conv path: algorithmic bytes / HIP-event time against the HBM peak
            "roofline_hbm_kernels": hbm,
            "by_kernel": {k: {"ms_per_step": round(v["ms_per_step"], 3), "tflops": round(v["tflops"], 1),
                              "launches_per_step": v["launches_per_step"]} for k, v in kernels.items()},
            "extra": extras or None,
        }
        if world == 1 and not o.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(P, seed)
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
