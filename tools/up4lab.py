"""Lab: the decoders' three ConvTranspose2d(4, 2, 1) layers of the 512 x 512 generator as the fused F(2x2, 2x2) Winograd launch
(csrc/convt_winograd.hip) against the direct forms (four parity launches / one grid), per layer: launch time, executed TFLOP/s (2 M 36/4 Cin N per
input pixel: 36 products per 4 x 4 patch) against the fp32 matrix pipe (157.3), the direct time and max |wino - direct|.
usage: up4lab.py [--lib LIB.so] [--frames F] [--reps n] [--only i]"""
import argparse
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser()
ap.add_argument("--lib", default=None)
ap.add_argument("--frames", type=int, default=16)
ap.add_argument("--reps", type=int, default=10)
ap.add_argument("--only", type=int, default=-1)
args = ap.parse_args()
import torch
from ipercore_amd import _lib
if args.lib:
    _lib.LIB_PATH = os.path.abspath(args.lib)
from ipercore_amd import ops
from ipercore_amd.networks import packing

dev = "cuda:0"
SHAPES = [("up0 64^2 256->256", 64, 256, 256, False), ("up1 128^2 256->128", 128, 256, 128, False), ("up2 256^2 128->64 (quad planes)", 256, 128, 64, True)]


def rnd(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def timeit(fn, n):
    for _ in range(2):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


tw_all = td_all = fl_all = 0.0
for idx, (tag, S, Cin, N, q4) in enumerate(SHAPES):
    if args.only >= 0 and idx != args.only:
        continue
    B = args.frames
    specs = [packing.spec_to(s, dev) for s in packing.pack_conv_transpose(rnd((Cin, N, 4, 4), 3 + idx, (Cin * 4) ** -0.5), rnd((N,), 4 + idx, 0.1))]
    x = rnd((B, S, S, Cin), 5 + idx).to(dev)
    shape = (B, N // 4, 2 * S, 2 * S, 4) if q4 else (B, 2 * S, 2 * S, N)
    yw, yd = torch.empty(*shape, device=dev), torch.empty(*shape, device=dev)

    def run_w():
        with ops.conv_precision("winograd"):
            ops.conv_transpose2d(x, specs, yw, act=ops.ACT_RELU, q4=q4)

    def run_d():
        with ops.conv_precision("fp32"):
            ops.conv_transpose2d(x, specs, yd, act=ops.ACT_RELU, q4=q4)

    tw, td = timeit(run_w, args.reps), timeit(run_d, args.reps)
    torch.cuda.synchronize()
    M = B * S * S
    ex, al = 2.0 * M * 9 * Cin * N, 2.0 * M * 16 * Cin * N
    tw_all, td_all, fl_all = tw_all + tw, td_all + td, fl_all + ex
    print(f"{idx} {tag:34s} B={B:3d}: wino {tw * 1e3:8.1f} us  executed {ex / tw / 1e9:6.1f} TF/s = {ex / tw / 1e9 / 157.3:.3f} of the pipe  algorithmic {al / tw / 1e9:6.1f} TF/s"
          f"  | direct {td * 1e3:8.1f} us ({al / td / 1e9 / 157.3:.3f})  x{td / tw:.2f}  max|d| {float((yw - yd).abs().max()):.1e}", flush=True)
if tw_all and args.only < 0:
    print(f"sum: wino {tw_all * 1e3:.1f} us, executed {fl_all / tw_all / 1e9 / 157.3:.3f} of the pipe; direct {td_all * 1e3:.1f} us (x{td_all / tw_all:.2f})")
