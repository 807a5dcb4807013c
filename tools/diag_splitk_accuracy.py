"""Kernel-level accuracy of the launch forms the one-sample training step picks by shape, against an fp64 convolution of the SAME operands:
the direct fp32 MFMA kernel whole and split over K (lwg_conv2d_nhwc_f32_ws), the Winograd kernel whole and with its K loop in slices
(lwg_conv2d_winograd_f32_ws), on the background network's residual-block shape (1 x 64 x 64 x 256 -> 256) and a discriminator shape, forward
and data-gradient panels, with post-ReLU-like inputs.  Question: is a split launch LESS accurate than a whole one?
usage: python tools/diag_splitk_accuracy.py"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.nn.functional as F

from ipercore_amd import ops
from ipercore_amd.networks import packing

DEV = "cuda:0"


def rel(y, ref):
    d = y.double().cpu() - ref
    return (d.norm() / ref.norm()).item(), (d.abs().max() / ref.abs().max()).item()


def main():
    g = torch.Generator().manual_seed(3)
    rows = []
    for tag, (B, H, W, Cin, N, k, stride, pad) in (("res block 64^2 256->256 3x3", (1, 64, 64, 256, 256, 3, 1, 1)),
                                                    ("D 32^2 256->512 4x4 s2", (1, 64, 64, 256, 512, 4, 2, 1)),
                                                    ("SPADE shared 64^2 256->128 3x3", (1, 64, 64, 256, 128, 3, 1, 1))):
        w = torch.randn(N, Cin, k, k, generator=g) * (Cin * k * k) ** -0.5
        b = torch.randn(N, generator=g) * 0.1
        for xkind in ("normal", "post-relu"):
            x = torch.randn(B, H, W, Cin, generator=g)
            if xkind == "post-relu":
                x = x.relu()
            ref = F.conv2d(x.double().permute(0, 3, 1, 2), w.double(), b.double(), stride=stride, padding=pad).permute(0, 2, 3, 1)
            spec = packing.spec_to(packing.pack_conv(w, b, stride=stride, pad=pad), DEV)
            xd = x.to(DEV)
            OH, OW = ref.shape[1:3]
            forms = [("direct whole", "fp32", False), ("direct split-K", "fp32", True)]
            if k == 3 and stride == 1:
                forms += [("winograd whole", "winograd", False), ("winograd K slices", "winograd", True)]
            for name, prec, sk in forms:
                y = torch.empty(B, OH, OW, N, device=DEV)
                prev, ops.WINO_MIN_GRID = ops.WINO_MIN_GRID, 0
                try:
                    with ops.conv_precision(prec):
                        a = ops.conv_args(xd, spec, y)
                        if prec == "winograd":
                            plan = ops._wino_plan(a, spec, y, sk)
                            slices = 0 if not plan else plan // (a.M * spec.N)
                        else:
                            slices = int(ops._lib.lib().lwg_conv2d_ws_floats(a) // (a.M * a.N)) if sk else 0
                        ops.conv2d(xd, spec, y, splitk=sk)
                finally:
                    ops.WINO_MIN_GRID = prev
                torch.cuda.synchronize()
                l2, mx = rel(y, ref)
                rows.append((tag, xkind, name, slices, l2, mx))
                print(f"{tag:34s} {xkind:10s} {name:18s} slices {slices}  rel L2 {l2:.2e}  max |d| / max |ref| {mx:.2e}", flush=True)
    # torch-ROCm's own fp32 convolution on the first shape, for scale
    B, H, W, Cin, N = 1, 64, 64, 256, 256
    w = torch.randn(N, Cin, 3, 3, generator=g) * (Cin * 9) ** -0.5
    x = torch.randn(B, Cin, H, W, generator=g).relu()
    ref = F.conv2d(x.double(), w.double(), padding=1)
    yt = F.conv2d(x.to(DEV), w.to(DEV), padding=1)
    print("torch-ROCm fp32 conv2d, res block shape, post-relu: rel L2 %.2e  max %.2e" % rel(yt, ref))
    yc = F.conv2d(x, w, padding=1)
    print("torch CPU fp32 conv2d,  res block shape, post-relu: rel L2 %.2e  max %.2e" % rel(yc, ref))


if __name__ == "__main__":
    main()
