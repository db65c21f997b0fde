#!/usr/bin/env python
"""Which ATen operators (and from which Python lines) still launch kernels inside one AV train step.

    python tools/aten_audit.py [--batch 8] [--precision f32]

torch.profiler over ONE step after two warm-up steps: device kernels grouped by name (count, us) with the library's own
kernels summed into one row, then the ATen ops with their Python call sites.  Measurement bookkeeping only."""
import argparse
import collections
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch                                                                      # noqa: E402
from torch.profiler import ProfilerActivity, profile                              # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--precision", default="f32")
    ap.add_argument("--ao", action="store_true")
    o = ap.parse_args()
    import avsep_amd as P
    import bench
    dev = torch.device("cuda", 0)
    P.kernels.set_precision(o.precision)
    a, snd, frm, wrap = bench.build(P, dev, 1234, "hip")
    opt = P.create_optimizer((snd, frm), a)
    raw = P.synth.make_batch(o.batch, a.num_mix, a.num_frames, 224, a.audLen, seed=5, device=dev)

    def step():
        b = {"audios": list(raw["audios"]), "audio_mix": raw["audio_mix"], "frames": list(raw["frames"])}
        return P.net_wrapper.train_step_async(wrap, b, opt, not o.ao, a)
    for _ in range(2):
        step()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
        step()
        torch.cuda.synchronize()
    kern = collections.Counter()
    kus = collections.Counter()
    ours_n = ours_us = 0
    for e in prof.events():
        if e.device_type is not None and str(e.device_type).endswith("CUDA"):
            name = e.name
            if name.startswith(("void at::", "at::", "__amd_rocclr", "Memcpy", "Memset", "void (anonymous namespace)", "void c10")) \
                    or "elementwise" in name or "rocclr" in name:
                kern[name[:90]] += 1
                kus[name[:90]] += e.device_time
            else:
                ours_n += 1
                ours_us += e.device_time
    print("library kernels: %d launches, %.1f ms" % (ours_n, ours_us / 1e3))
    print("non-library device activities:")
    for k, n in kern.most_common():
        print("  %5d  %8.1f us  %s" % (n, kus[k], k))
    print("total non-library launches per step:", sum(kern.values()))
    sites = collections.Counter()
    for e in prof.events():
        if e.name.startswith("aten::") and e.stack and e.device_time > 0 and not e.cpu_children:
            frames = [f for f in e.stack if ROOT in f and "tools/aten_audit" not in f][:2]
            sites[(e.name, " <- ".join(f.replace(ROOT + "/", "") for f in frames))] += 1
    print("ATen leaf ops with device time, by call site:")
    for (name, where), n in sites.most_common(60):
        print("  %4d  %-28s %s" % (n, name, where))


if __name__ == "__main__":
    main()
