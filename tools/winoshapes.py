"""Lab: the fused Winograd convolution (csrc/conv_winograd.hip) on the launch shapes of the 512 x 512 generator, per shape: launch time, executed
TFLOP/s (2 M 4 Cin N - sixteen products per 2 x 2 outputs) against the fp32 matrix pipe (157.3), algorithmic TFLOP/s (2 M 9 Cin N), the direct kernel's
time for the same launch and max |wino - direct|.
usage: winoshapes.py [--lib LIB.so] [--frames F] [--only i] [--reps n] [--nodirect] [--splitk] [--vgg] [--ts]
  --lib     a variant library (tools/labvariant.sh NAME conv_winograd.hip -D...) instead of the tree's
  --only i  shape i only (PMC passes: rocprofv3 --pmc ... -- python tools/winoshapes.py --only 0 --nodirect)
  --ts      the library is a -DLWG_WINO_TS build: print the per-workgroup phase timeline (s_memtime) of shape --only (plain epilogue)"""
import argparse
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser()
ap.add_argument("--lib", default=None)
ap.add_argument("--frames", type=int, default=16)
ap.add_argument("--only", type=int, default=-1)
ap.add_argument("--reps", type=int, default=10)
ap.add_argument("--nodirect", action="store_true")
ap.add_argument("--ts", action="store_true")
ap.add_argument("--splitk", action="store_true", help="the direct side with splitk=True (how the training convolutions call it) + a third column: the Winograd kernel with splitk=True")
ap.add_argument("--vgg", action="store_true", help="the VGG19 perceptual loss's launch shapes (224 x 224 input) instead of the generator's")
ap.add_argument("--ts2", action="store_true", help="-DLWG_WINO_TS2 build: per-wave slot timeline of iterations 8 and 9 (shape --only, plain epilogue, Cin >= 96)")
args = ap.parse_args()
import torch
from ipercore_amd import _lib
if args.lib:
    _lib.LIB_PATH = os.path.abspath(args.lib)
from ipercore_amd import ops
from ipercore_amd.networks import packing

dev = "cuda:0"
F_ = args.frames
# (tag, frames multiplier, H = W, C0, C1, Cout, epilogue)
SHAPES = [("res 64^2 256->256 residual", 1, 64, 256, 0, 256, "res"),
          ("spade shared 64^2 256->128", 1, 64, 256, 0, 128, "none"),
          ("spade gamma|beta 64^2 128->2x256", 1, 64, 128, 0, 256, "spade"),
          ("spade shared 128^2 128->128", 1, 128, 128, 0, 128, "none"),
          ("spade gamma|beta 128^2 128->2x128", 1, 128, 128, 0, 128, "spade"),
          ("spade shared 256^2 64->128", 1, 256, 64, 0, 128, "none"),
          ("spade gamma|beta 256^2 128->2x64", 1, 256, 128, 0, 64, "spade"),
          ("skip0 128^2 256+128->256", 1, 128, 256, 128, 256, "none"),
          ("skip1 256^2 128+64->128", 1, 256, 128, 64, 128, "none")]
if args.vgg:
    SHAPES = [(f"vgg {S}^2 {ci}->{co}", 1, S, ci, 0, co, "none") for S, ci, co in
              ((224, 64, 64), (112, 64, 128), (112, 128, 128), (56, 128, 256), (56, 256, 256), (28, 256, 512), (28, 512, 512), (14, 512, 512))]


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


tot_w = tot_d = tot_fl = 0.0
for idx, (tag, mul, S, C0, C1, Co, epi) in enumerate(SHAPES):
    if args.only >= 0 and idx != args.only:
        continue
    B, Cin = F_ * mul, C0 + C1
    x0 = rnd((B, S, S, C0), 1 + idx).to(dev)
    x1 = rnd((B, S, S, C1), 2 + idx).to(dev) if C1 else None
    kw = dict(act=ops.ACT_RELU)
    if epi == "spade":
        spec = packing.spec_to(packing.pack_spade_gamma_beta(rnd((Co, Cin, 3, 3), 3, 0.03), rnd((Co,), 4, 0.1), rnd((Co, Cin, 3, 3), 5, 0.03), rnd((Co,), 6, 0.1)), dev)
        xn = (rnd((B, S, S, Co), 7, 2.0) + 0.5).to(dev)
        mean = xn.reshape(B, -1, Co).mean(1).contiguous()
        rstd = (1 / torch.sqrt(xn.reshape(B, -1, Co).var(1, unbiased=False) + 1e-5)).contiguous()
        kw.update(epi=ops.EPI_SPADE, xn=xn, mean=mean, rstd=rstd)
        N = 2 * Co
    else:
        spec = packing.spec_to(packing.pack_conv(rnd((Co, Cin, 3, 3), 3, (Cin * 9) ** -0.5), rnd((Co,), 4, 0.1), stride=1, pad=1), dev)
        N = Co
        if epi == "res":
            kw.update(epi=ops.EPI_RESIDUAL, res=rnd((B, S, S, Co), 8).to(dev))
    yw, yd = torch.empty(B, S, S, Co, device=dev), torch.empty(B, S, S, Co, device=dev)
    if args.ts2:
        nblk = ((S + 15) // 16) ** 2 * B * (N // 64)
        stamps = torch.zeros(nblk * 8 * 16 * 2, device=dev)
        with ops.conv_precision("winograd"):
            for _ in range(2):
                ops.conv2d(x0, spec, yw, x1=x1, act=ops.ACT_RELU, res=stamps)
        torch.cuda.synchronize()
        t = stamps.view(torch.int64).view(nblk, 8, 16).cpu()
        names = ["top", "slot4", "slot8", "slot16", "pre-barrier", "post-barrier", "slot29", "end"]
        for blk in (0, nblk // 2, nblk - 1):
            base = int(t[blk, :, 0].min())
            print(f"[ts2] {tag}: workgroup {blk}: cycles since the first wave's top of iteration 8 (rows = waves 0..7; columns = " + ", ".join(names) + " of iteration 8, then 9)")
            for w in range(8):
                print("   wave %d: " % w + " ".join("%6d" % (int(t[blk, w, i]) - base) for i in range(16)))
        d = (t[:, :, 8] - t[:, :, 0]).double()
        print(f"     iteration 8 top -> iteration 9 top, per wave: mean over workgroups " + " ".join("%.0f" % d[:, w].mean() for w in range(8)))
        bar = (t[:, :, 5] - t[:, :, 4]).double()
        print(f"     cycles parked at the barrier of iteration 8, per wave: " + " ".join("%.0f" % bar[:, w].mean() for w in range(8)))
        continue
    if args.ts:
        nblk = ((S + 15) // 16) ** 2 * B * (N // 64)
        stamps = torch.zeros(nblk * 128, device=dev)              # 64 x u64 per workgroup
        kw = dict(act=ops.ACT_RELU, res=stamps)
        with ops.conv_precision("winograd"):
            for _ in range(3):
                ops.conv2d(x0, spec, yw, x1=x1, **kw)
        torch.cuda.synchronize()
        t = stamps.view(torch.int64).view(nblk, 64).cpu()
        nst = Cin // 8
        d_pro = (t[:, 1] - t[:, 0]).double()
        d_loop = (t[:, 40] - t[:, 1]).double()
        d_e0 = (t[:, 41] - t[:, 40]).double()
        d_e1 = (t[:, 42] - t[:, 41]).double()
        tot = (t[:, 42] - t[:, 0]).double()
        pairs = (t[:, 3:2 + nst // 2 - 1] - t[:, 2:2 + nst // 2 - 2]).double() / 2 if nst >= 8 else None
        print(f"[ts] {tag} B={B}: workgroups {nblk}, stages {nst}; cycles (s_memtime) mean / median: prologue {d_pro.mean():.0f} / {d_pro.median():.0f}  K loop {d_loop.mean():.0f} / {d_loop.median():.0f}"
              f" (per stage {d_loop.mean() / nst:.0f}; ideal 4096)  epilogue halves {d_e0.mean():.0f} + {d_e1.mean():.0f}  total {tot.mean():.0f}")
        if pairs is not None:
            print(f"     steady-state stage (inside the loop, from stamp pairs): mean {pairs.mean():.0f} median {pairs.median():.0f} min {pairs.min():.0f} max {pairs.max():.0f}")
        # persistent form (round 6): the workgroup's block 1 (steady state) and the prologue of its block 2 (needs >= 3 blocks per workgroup: --frames 48)
        ok = (t[:, 53] > 0) & (t[:, 44] > 0)
        if int(ok.sum()) > 0:
            u = t[ok].double()
            seg = [("K loop", 45, 44), ("fold -> Ms", 54, 45), ("next block's set-up + loads issued", 46, 54), ("barrier", 47, 46), ("Ms reads + output transform + stores", 48, 47),
                   ("barrier", 49, 48), ("raw stages 0 / 1 -> LDS + barrier", 51, 50), ("transform(0) + barrier", 52, 51), ("first fragments + accumulator clear", 53, 52)]
            print(f"     persistent block timeline ({int(ok.sum())} workgroups; medians): " + "; ".join(f"{n} {float((u[:, i1] - u[:, i0]).median()):.0f}" for n, i1, i0 in seg)
                  + f"; block 1 K-loop entry -> block 2 K-loop entry {float((u[:, 53] - u[:, 44]).median()):.0f}")
        span = float(t[:, 42].max() - t[:, 0].min())
        print(f"     first entry -> last exit {span:.0f} ticks; sum of workgroup cycles / 256 CUs = {float(tot.sum()) / 256:.0f}")
        continue

    def run_w():
        with ops.conv_precision("winograd"):
            ops.conv2d(x0, spec, yw, x1=x1, **kw)

    def run_d():
        ops.conv2d(x0, spec, yd, x1=x1, splitk=args.splitk, **kw)

    def run_ws():                                    # the training form: the K loop in slices when the launch leaves the chip half empty
        with ops.conv_precision("winograd"):
            ops.conv2d(x0, spec, ys, x1=x1, splitk=True, **kw)

    run_w()
    tw = timeit(run_w, args.reps)
    ex, al = 2.0 * B * S * S * 4 * Cin * N, 2.0 * B * S * S * 9 * Cin * N
    line = f"{idx} {tag:36s} B={B:3d}: wino {tw * 1e3:8.1f} us  executed {ex / tw / 1e9:6.1f} TF/s = {ex / tw / 1e9 / 157.3:.3f} of the pipe  algorithmic {al / tw / 1e9:6.1f} TF/s"
    tot_w += tw
    tot_fl += ex
    if not args.nodirect:
        run_d()
        td = timeit(run_d, args.reps)
        tot_d += td
        torch.cuda.synchronize()
        line += f"  | direct {td * 1e3:8.1f} us ({al / td / 1e9 / 157.3:.3f})  x{td / tw:.2f}  max|d| {float((yw - yd).abs().max()):.1e}"
    if args.splitk:
        ys = torch.empty_like(yw)
        ops.WINO_MIN_GRID = 0
        run_ws()
        ts = timeit(run_ws, args.reps)
        line += f"  | wino split-K {ts * 1e3:8.1f} us (max|d| vs whole {float((ys - yw).abs().max()):.1e})"
    print(line, flush=True)
if tot_w and args.only < 0:
    print(f"sum: wino {tot_w * 1e3:.1f} us, executed {tot_fl / tot_w / 1e9 / 157.3:.3f} of the pipe" + (f"; direct {tot_d * 1e3:.1f} us (x{tot_d / tot_w:.2f})" if tot_d else ""))
