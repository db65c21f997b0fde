"""Times the HBM-bound glue kernels of the step (BatchNorm / ReLU / pooling / upsample passes, decoder head) on cuda:0 at
the layer shapes of the batch-64 bench step.  Prints ms and ALGORITHMIC GB/s (every operand touched once) per call.
Usage: python tools/elem_bench.py [--reps 20] [--only name]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import avsep_amd as P  # noqa: E402

TRUNK = [(192, 64, 112, 112), (192, 64, 56, 56), (192, 128, 28, 28), (192, 256, 14, 14), (192, 512, 14, 14)]
UNET = [(64, 64, 128, 128), (64, 128, 64, 64), (64, 256, 32, 32), (64, 512, 16, 16)]


def timeit(fn, reps):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--only", default="")
    o = ap.parse_args()
    K = P.kernels
    dev = torch.device("cuda:0")
    rows = []

    def rec(name, shape, ms, tensors):
        n = 1
        for s in shape:
            n *= s
        gb = 4.0 * n * tensors / 1e9
        rows.append((name, shape, ms, gb / (ms * 1e-3)))
        print("%-28s %-22s %8.3f ms %8.1f GB/s (%d tensor passes)" % (name, shape, ms, gb / (ms * 1e-3), tensors), flush=True)

    for shape in TRUNK[1:] + UNET:
        N, C, H, W = shape
        y, dz, res = (torch.randn(shape, device=dev) for _ in range(3))
        sc, sh = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev)
        mean, inv = torch.randn(C, device=dev), torch.rand(C, device=dev) + 0.5
        pqr = torch.randn(3, C, device=dev)
        if not o.only or o.only in "affine_act":
            rec("affine_act(res)", shape, timeit(lambda: K.affine_act(y, sc, sh, res, 1), o.reps), 3)
        if not o.only or o.only in "affine_act_bwd":
            st = K.zeros_stats(C, y)
            rec("affine_act_bwd(res)", shape, timeit(lambda: K.affine_act_bwd_(dz, y, sc, sh, res, None, mean, inv, 1, st), o.reps), 4)
            rec("affine_act_bwd", shape, timeit(lambda: K.affine_act_bwd_(dz, y, sc, sh, None, None, mean, inv, 1, st), o.reps), 3)
        if not o.only or o.only in "bn_bwd_apply":
            rec("bn_bwd_apply", shape, timeit(lambda: K.bn_bwd_apply_(dz, y, pqr), o.reps), 3)
            out = torch.empty_like(dz)
            rec("bn_bwd_apply(out)", shape, timeit(lambda: K.bn_bwd_apply_(dz, y, pqr, out=out), o.reps), 3)
    if not o.only or o.only in "maxpool":
        shape = TRUNK[0]
        N, C, H, W = shape
        y = torch.randn(shape, device=dev)
        sc, sh = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev)
        rec("maxpool3x3s2(bn,relu)", shape, timeit(lambda: K.maxpool3x3s2(y, sc, sh, 1), o.reps), 1.5)
    if not o.only or o.only in "relu_up2x":
        for (N, C, H, W) in [(64, 64, 128, 128), (64, 128, 64, 64), (64, 256, 32, 32), (64, 512, 16, 16), (64, 512, 8, 8)]:
            # decoder level: cat(skip [N,C,H,W], inner [N,C,H,W]) -> relu + x2 -> [N,2C,2H,2W]
            x0, x1 = torch.randn(N, C, H, W, device=dev), torch.randn(N, C, H, W, device=dev)
            s0, h0, s1, h1 = (torch.rand(C, device=dev) + 0.5 for _ in range(4))
            cat = K.Cat(x0, x1, s0, h0, s1, h1)
            ms = timeit(lambda: cat.fwd(), o.reps)
            rec("relu_up2x_fwd", (N, 2 * C, 2 * H, 2 * W), ms, 1.25)
            dout = torch.randn(N, 2 * C, 2 * H, 2 * W, device=dev)
            mean, inv = torch.randn(C, device=dev), torch.rand(C, device=dev) + 0.5
            st = K.zeros_stats(C, x0)
            ms = timeit(lambda: cat.bwd(dout, mean, inv, st), o.reps)
            rec("relu_up2x_bwd", (N, 2 * C, 2 * H, 2 * W), ms, 1.5)
    if not o.only or o.only in "head":
        N, C, H, W = 64, 64, 128, 128
        x0, x1 = torch.randn(N, C, H, W, device=dev), torch.randn(N, C, H, W, device=dev)
        s0, h0, s1, h1 = (torch.rand(C, device=dev) + 0.5 for _ in range(4))
        w = torch.randn(2, 2 * C, 3, 3, device=dev) * 0.05
        b = torch.randn(2, device=dev)
        cv = K.Conv(x0, 2, 3, 1, 1, x1=x1, sc0=s0, sh0=h0, act0=1, sc1=s1, sh1=h1, act1=1, up2x=True)
        assert cv.head_applicable()
        wp = cv.pack(w, 0)
        lo = N * 2 * C * H * W
        hi = N * 2 * 4 * H * W
        ms = timeit(lambda: cv.fwd(wp, b, None), o.reps)
        print("%-28s %-22s %8.3f ms %8.1f GB/s (low-res sources + logits), %6.1f TFLOP/s" %
              ("head_fwd", (N, 2 * C, 2 * H, 2 * W), ms, 4.0 * (lo + hi) / ms / 1e6, 2.0 * N * 4 * H * W * 2 * 2 * C * 9 / ms / 1e9), flush=True)
        dy = torch.randn(N, 2, 2 * H, 2 * W, device=dev)
        ms = timeit(lambda: cv.wgrad(dy, want_bias=True), o.reps)
        print("%-28s %-22s %8.3f ms %8.1f GB/s" % ("head_wgrad", (N, 2 * C, 2 * H, 2 * W), ms, 4.0 * (lo + hi) / ms / 1e6), flush=True)
        mean, inv = torch.randn(C, device=dev), torch.rand(C, device=dev) + 0.5
        st = K.zeros_stats(C, x0)
        ms = timeit(lambda: cv.dgrad_up2x(w, dy, mean1=mean, invstd1=inv, bstats1=st), o.reps)
        print("%-28s %-22s %8.3f ms %8.1f GB/s" % ("head_dgrad", (N, 2 * C, 2 * H, 2 * W), ms, 4.0 * (2 * lo + hi) / ms / 1e6), flush=True)


if __name__ == "__main__":
    main()
