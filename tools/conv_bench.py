"""Times single convolution calls of the C ABI on cuda:0 (forward / data gradient / weight gradient) at the layer
shapes of the bench step.  Usage: python tools/conv_bench.py [fwd|dgrad|wgrad] [--prec f32|bf16] [--reps 20]
Prints ms and direct-form TFLOP/s (2*N*Ho*Wo*Cout*Cin*k*k / time) per shape and the kernel family that served it."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import avsep_amd as P  # noqa: E402

SHAPES = [
    # N, Cin, H, W, Cout, k, stride, pad, dil
    (64, 1024, 32, 32, 256, 3, 1, 1, 1),
    (64, 512, 64, 64, 128, 3, 1, 1, 1),
    (64, 256, 128, 128, 64, 3, 1, 1, 1),
    (64, 1024, 16, 16, 512, 3, 1, 1, 1),
    (192, 64, 56, 56, 64, 3, 1, 1, 1),
    (192, 128, 28, 28, 128, 3, 1, 1, 1),
    (192, 256, 14, 14, 256, 3, 1, 1, 1),
    (192, 512, 14, 14, 512, 3, 1, 2, 2),
    (192, 256, 14, 14, 512, 3, 1, 1, 1),
]
LEFTOVERS = [    # the shapes profiles/r02_layers_f32.txt lists on the im2col kernel
    (192, 3, 224, 224, 64, 7, 2, 3, 1), (64, 1, 256, 256, 64, 4, 2, 1, 1),
    (192, 64, 56, 56, 128, 3, 2, 1, 1), (192, 128, 28, 28, 256, 3, 2, 1, 1),
    (192, 256, 14, 14, 512, 1, 1, 0, 1), (192, 64, 56, 56, 128, 1, 2, 0, 1), (192, 128, 28, 28, 256, 1, 2, 0, 1),
    (64, 512, 16, 16, 512, 4, 2, 1, 1), (64, 512, 8, 8, 512, 4, 2, 1, 1), (64, 512, 4, 4, 512, 4, 2, 1, 1),
    (64, 1024, 4, 4, 512, 3, 1, 1, 1),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("mode", nargs="?", default="fwd", choices=["fwd", "dgrad", "wgrad"])
    ap.add_argument("--prec", default="f32")
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--leftovers", action="store_true", help="the layer shapes still on the im2col kernel in round 2")
    ap.add_argument("--affine", action="store_true", help="forward over a folded BatchNorm + ReLU input, with statistics")
    ap.add_argument("--tune", type=lambda v: int(v, 0), default=0, help="avsep_conv_desc.tune (include/avsep.h): bits 0-3 Winograd "
                    "group shape + 1, bits 4-7 Winograd weight-gradient group shape + 1, bits 8-23 its target workgroup count")
    ap.add_argument("--algo-no", default="", help="comma list of kernel families the calls must not use (lib.ALGO_NO)")
    ap.add_argument("--act-epilogue", default="", choices=["", "bn1", "tail"], help="dgrad only: time avsep_conv2d_dgrad_act (bn1: through "
                    "relu(bn(y)); tail: through relu(bn(y) + residual) with a second gradient branch) against dgrad + affine_act_bwd")
    o = ap.parse_args()
    K = P.kernels
    K.conv_tune = o.tune
    K.set_algo_mask(*[n for n in o.algo_no.split(",") if n])
    K.set_precision(o.prec)
    dev = torch.device("cuda:0")
    for (N, Cin, H, W, Cout, k, s, p, d) in (LEFTOVERS if o.leftovers else SHAPES):
        x = torch.randn(N, Cin, H, W, device=dev)
        w = torch.randn(Cout, Cin, k, k, device=dev) * 0.02
        kw = {}
        if o.affine:
            kw = dict(sc0=torch.rand(Cin, device=dev) + 0.5, sh0=torch.randn(Cin, device=dev), act0=1)
        cv = K.Conv(x, Cout, k, s, p, d, **kw)
        st = K.zeros_stats(Cout, x) if o.affine else None
        dy = torch.randn(N, Cout, cv.Ho, cv.Wo, device=dev)
        wp = cv.pack(w, 1 if o.mode == "dgrad" else 0)
        run = {"fwd": lambda: cv.fwd(wp, None, st), "dgrad": lambda: cv.dgrad(wp, dy), "wgrad": lambda: cv.wgrad(dy)}[o.mode]
        if o.act_epilogue and o.mode == "dgrad":
            C = Cin
            yb, res, dz2 = (torch.randn(N, C, H, W, device=dev) for _ in range(3))
            sc, sh, mu, isd = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev), torch.randn(C, device=dev), torch.rand(C, device=dev) + 0.5
            bst = K.zeros_stats(C, x)
            tail = o.act_epilogue == "tail"
            kw2 = dict(residual=res, dz2=dz2) if tail else {}
            fused = lambda: cv.dgrad_act(wp, dy, yb, sc, sh, mu, isd, 1, bst, **kw2)
            two = lambda: K.affine_act_bwd_(cv.dgrad(wp, dy), yb, sc, sh, res if tail else None, None, mu, isd, 1, bst, dz2=dz2 if tail else None)
            for tag, fn in (("dgrad_act", fused), ("two launches", two)):
                for _ in range(3):
                    fn()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                torch.cuda.synchronize()
                e0.record()
                for _ in range(o.reps):
                    fn()
                e1.record()
                torch.cuda.synchronize()
                print("%-40s %-14s %-12s fused=%d %8.3f ms" % ((N, Cin, H, W, Cout, k, s, p, d), tag, o.act_epilogue, cv.dgrad_act_fused(),
                                                             e0.elapsed_time(e1) / o.reps))
            continue
        for _ in range(3):
            run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(o.reps):
            run()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / o.reps
        fl = 2.0 * N * cv.Ho * cv.Wo * Cout * Cin * k * k
        print("%-40s %-6s %-18s %8.3f ms %7.1f TFLOP/s" % ((N, Cin, H, W, Cout, k, s, p, d), o.mode, cv.kernel_name(o.mode, o.affine), ms,
                                                           fl / ms / 1e9))


if __name__ == "__main__":
    main()
