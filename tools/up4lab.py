"""Lab: the decoders' three ConvTranspose2d(4, 2, 1) layers of the 512 x 512 generator as the fused F(2x2, 2x2) Winograd launch
(csrc/convt_winograd.hip) against the direct forms (four parity launches / one grid), per layer: launch time, executed TFLOP/s (2 M 36/4 Cin N per
input pixel: 36 products per 4 x 4 patch) against the fp32 matrix pipe (157.3), the direct time and max |wino - direct|.
usage: up4lab.py [--lib VARIANT.so] [--frames F] [--reps n] [--only i] [--ts]   (VARIANT.so: csrc/convt_winograd.hip alone, built with extra -D flags)"""
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
ap.add_argument("--ts", action="store_true", help="the library is a -DLWG_CTW_TS build: per-wave phase stamps (entry, K-loop entry, K-loop exit, end)")
args = ap.parse_args()
import torch
from ipercore_amd import _lib
from ipercore_amd import ops
from ipercore_amd.networks import packing

dev = "cuda:0"
ENTRY = _lib.lib().lwg_conv_transpose4_winograd_f32
if args.lib:          # a variant build of csrc/convt_winograd.hip alone: hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared [-D...] convt_winograd.hip -o tools/lab/X.so
    import ctypes
    ENTRY = ctypes.CDLL(os.path.abspath(args.lib)).lwg_conv_transpose4_winograd_f32
    ENTRY.restype, ENTRY.argtypes = ctypes.c_int, [ctypes.POINTER(_lib.LwgConvArgs), ctypes.c_void_p]
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

    aw = ops.conv_args(x, specs[0], yw, act=ops.ACT_RELU, q4=q4)
    aw.w = ops._ptr(ops._wwino_t(specs))

    def run_w():
        _lib.check(ENTRY(aw, ops._stream()), "lwg_conv_transpose4_winograd_f32")

    def run_d():
        with ops.conv_precision("fp32"):
            ops.conv_transpose2d(x, specs, yd, act=ops.ACT_RELU, q4=q4)

    if args.ts:
        nblk = ((S + 15) // 16) ** 2 * B * (N // 32)
        stamps = torch.zeros(nblk * 8 * 16 * 2, device=dev)
        a = ops.conv_args(x, specs[0], yw, act=ops.ACT_RELU, q4=q4)
        a.w, a.res = ops._ptr(ops._wwino_t(specs)), ops._ptr(stamps)
        for _ in range(3):
            _lib.check(ENTRY(a, ops._stream()), "lwg_conv_transpose4_winograd_f32")
        torch.cuda.synchronize()
        t = stamps.view(torch.int64).view(nblk, 8, 16).cpu().double()
        nst = Cin // 8
        pro, loop, epi = t[:, :, 1] - t[:, :, 0], t[:, :, 2] - t[:, :, 1], t[:, :, 3] - t[:, :, 2]
        print(f"[ts] {tag} B={B}: workgroups {nblk}, stages {nst}; cycles mean over waves (median): prologue {pro.mean():.0f} ({pro.median():.0f})  K loop {loop.mean():.0f} "
              f"= {loop.mean() / nst:.0f} per stage (ideal 4608: 36 MFMAs x 64 cycles x 2 waves per SIMD)  epilogue {epi.mean():.0f} ({epi.median():.0f})")
        print("     per wave, K loop per stage: " + " ".join("%.0f" % (loop[:, w].mean() / nst) for w in range(8)) + "   epilogue: " + " ".join("%.0f" % epi[:, w].mean() for w in range(8)))
        ok = (t[:, 0, 12] > 0) & (t[:, 0, 4] > 0)      # persistent form: the workgroup's block 1 and the prologue of block 2 (needs >= 3 blocks per workgroup)
        if int(ok.sum()) > 0:
            seg = [("K loop", 5, 4), ("barrier", 6, 5), ("output transform -> LDS", 7, 6), ("next block's set-up + loads issued", 8, 7), ("barrier", 9, 8), ("stores", 13, 9),
                   ("barrier + raw stages -> LDS + barrier", 11, 13), ("transform(0) + barrier + first fragments + clear", 12, 11)]
            for w in (0, 7):
                u = t[ok][:, w, :]
                print(f"     persistent block timeline, wave {w} ({int(ok.sum())} workgroups; medians): " + "; ".join(f"{n} {float((u[:, i1] - u[:, i0]).median()):.0f}" for n, i1, i0 in seg)
                      + f"; block 1 K-loop entry -> block 2 K-loop entry {float((u[:, 12] - u[:, 4]).median()):.0f}")
        wg = (t[:, :, 3].max(dim=1).values - t[:, :, 0].min(dim=1).values)
        print(f"     workgroup entry -> exit {wg.mean():.0f}; sum over workgroups / 256 CUs = {wg.sum() / 256:.0f}; first entry -> last exit {float(t[:, :, 3].max() - t[:, :, 0].min()):.0f}")
        continue
    tw, td = timeit(run_w, args.reps), timeit(run_d, args.reps)
    torch.cuda.synchronize()
    M = B * S * S
    ex, al = 2.0 * M * 9 * Cin * N, 2.0 * M * 16 * Cin * N
    tw_all, td_all, fl_all = tw_all + tw, td_all + td, fl_all + ex
    print(f"{idx} {tag:34s} B={B:3d}: wino {tw * 1e3:8.1f} us  executed {ex / tw / 1e9:6.1f} TF/s = {ex / tw / 1e9 / 157.3:.3f} of the pipe  algorithmic {al / tw / 1e9:6.1f} TF/s"
          f"  | direct {td * 1e3:8.1f} us ({al / td / 1e9 / 157.3:.3f})  x{td / tw:.2f}  max|d| {float((yw - yd).abs().max()):.1e}", flush=True)
if tw_all and args.only < 0:
    print(f"sum: wino {tw_all * 1e3:.1f} us, executed {fl_all / tw_all / 1e9 / 157.3:.3f} of the pipe; direct {td_all * 1e3:.1f} us (x{td_all / tw_all:.2f})")
