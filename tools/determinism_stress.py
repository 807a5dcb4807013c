"""Lab: run-to-run determinism of the persistent Winograd kernels at large batch (the XCD-aware block order, several blocks per workgroup): every launch of
a shape is repeated and compared bit for bit with the first result, and frame n of the batch with the frame alone.
usage: determinism_stress.py [--lib LIB.so] [--reps n] [--frames F]"""
import argparse
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser()
ap.add_argument("--lib", default=None)
ap.add_argument("--reps", type=int, default=20)
ap.add_argument("--frames", type=int, default=96)
args = ap.parse_args()
import torch
from ipercore_amd import _lib
if args.lib:
    _lib.LIB_PATH = os.path.abspath(args.lib)
from ipercore_amd import ops
from ipercore_amd.networks import packing

dev = "cuda:0"


def rnd(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def dev_spec(s):
    return packing.spec_to(s, dev)


FR = args.frames
bad = 0
# F(4x4,3x3): (tag, H, Cin, N, epi)
for tag, H, Cin, N, epi in (("res 64^2 256->256", 64, 256, 256, "res"), ("shared 64^2 256->128", 64, 256, 128, "none"), ("gb 64^2 128->2x256", 64, 128, 256, "spade"),
                            ("shared 128^2 128->128", 128, 128, 128, "none"), ("skip 128^2 256+128->256", 128, 384, 256, "none")):
    B = FR if H == 64 else FR // 4
    x = rnd((B, H, H, Cin), 1).to(dev)
    if epi == "spade":
        sp = dev_spec(packing.pack_spade_gamma_beta(rnd((N, Cin, 3, 3), 2, 0.03), rnd((N,), 3, 0.1), rnd((N, Cin, 3, 3), 4, 0.03), rnd((N,), 5, 0.1)))
        xn = rnd((B, H, H, N), 6).to(dev)
        mean, rstd = xn.reshape(B, -1, N).mean(1).contiguous(), (1 / torch.sqrt(xn.reshape(B, -1, N).var(1, unbiased=False) + 1e-5)).contiguous()
        kw = dict(epi=ops.EPI_SPADE, act=ops.ACT_RELU, xn=xn, mean=mean, rstd=rstd)
    else:
        sp = dev_spec(packing.pack_conv(rnd((N, Cin, 3, 3), 2, (Cin * 9) ** -0.5), rnd((N,), 3, 0.1), stride=1, pad=1))
        kw = dict(act=ops.ACT_RELU)
        if epi == "res":
            kw.update(epi=ops.EPI_RESIDUAL, res=rnd((B, H, H, N), 7).to(dev))
    outs = []
    with ops.conv_precision("winograd"):
        for r in range(args.reps):
            y = torch.empty(B, H, H, N, device=dev)
            ops.conv2d(x, sp, y, **kw)
            outs.append(y)
        y1 = torch.empty(1, H, H, N, device=dev)
        kw1 = {k: (v[-1:].contiguous() if torch.is_tensor(v) else v) for k, v in kw.items()}
        ops.conv2d(x[-1:].contiguous(), sp, y1, **kw1)
    torch.cuda.synchronize()
    nd = sum(0 if torch.equal(outs[0], o) else 1 for o in outs[1:])
    alone = torch.equal(outs[0][-1:], y1)
    bad += nd + (0 if alone else 1)
    print(f"F(4x4,3x3) {tag:28s} B={B:3d}: {nd} of {args.reps - 1} repeats differ from the first; last frame == the frame alone: {alone}")
# transposed: (tag, H, Cin, Cout, q4)
for tag, H, Cin, Cout, q4 in (("up0 64^2 256->256", 64, 256, 256, False), ("up1 128^2 256->128", 128, 256, 128, False), ("up2 256^2 128->64 q4", 256, 128, 64, True)):
    B = {64: FR, 128: FR // 4, 256: max(2, FR // 16)}[H]
    specs = [dev_spec(s) for s in packing.pack_conv_transpose(rnd((Cin, Cout, 4, 4), 8, (Cin * 4) ** -0.5), rnd((Cout,), 9, 0.1))]
    x = rnd((B, H, H, Cin), 10).to(dev)
    shape = (B, Cout // 4, 2 * H, 2 * H, 4) if q4 else (B, 2 * H, 2 * H, Cout)
    outs = []
    with ops.conv_precision("winograd"):
        for r in range(args.reps):
            y = torch.empty(*shape, device=dev)
            ops.conv_transpose2d(x, specs, y, act=ops.ACT_RELU, q4=q4)
            outs.append(y)
        y1 = torch.empty(1, *shape[1:], device=dev)
        ops.conv_transpose2d(x[-1:].contiguous(), specs, y1, act=ops.ACT_RELU, q4=q4)
    torch.cuda.synchronize()
    nd = sum(0 if torch.equal(outs[0], o) else 1 for o in outs[1:])
    alone = torch.equal(outs[0][-1:], y1)
    bad += nd + (0 if alone else 1)
    print(f"F(2x2,2x2)^T {tag:26s} B={B:3d}: {nd} of {args.reps - 1} repeats differ from the first; last frame == the frame alone: {alone}")
print("determinism:", "OK" if bad == 0 else f"FAILED ({bad})")
sys.exit(0 if bad == 0 else 1)
