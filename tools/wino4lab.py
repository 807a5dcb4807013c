"""Lab: the fused F(4x4, 3x3) Winograd convolution (csrc/conv_winograd4.hip) against the F(2x2, 3x3) kernel and the direct kernel on the launch shapes of the
512 x 512 generator (+ ragged / tiny shapes with --parity), per shape: launch time, executed TFLOP/s (2 M 2.25 Cin N) against the fp32 matrix pipe (157.3),
algorithmic TFLOP/s (2 M 9 Cin N), relative L2 / max error of each form against torch's fp64 convolution of the same operands.
usage: wino4lab.py [--lib LIB.so] [--frames F] [--only i] [--reps n] [--parity] [--ts]"""
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
ap.add_argument("--w4only", action="store_true", help="time the F(4x4,3x3) kernel only (knock-out variant libraries: their results are wrong by construction)")
ap.add_argument("--parity", action="store_true", help="small / ragged / odd shapes: correctness only")
ap.add_argument("--half", action="store_true", help="--ts on a -DLWG_W4_SMALL=2 build: the half-block form's grid (2 workgroups per CU, 32 channels per block, 4-channel stages)")
ap.add_argument("--ts", action="store_true", help="-DLWG_W4_TS build: per-block stamps of wave 0 (second block of every workgroup; plain epilogue)")
args = ap.parse_args()
import torch
import torch.nn.functional as F
from ipercore_amd import _lib
if args.lib:
    _lib.LIB_PATH = os.path.abspath(args.lib)
from ipercore_amd import ops
from ipercore_amd.networks import packing

dev = "cuda:0"
# (tag, B, H, W, C0, C1, Cout, epilogue)
FR = args.frames
SHAPES = [("res 64^2 256->256 residual", FR, 64, 64, 256, 0, 256, "res"),
          ("spade shared 64^2 256->128", FR, 64, 64, 256, 0, 128, "none"),
          ("spade gamma|beta 64^2 128->2x256", FR, 64, 64, 128, 0, 256, "spade"),
          ("spade shared 128^2 128->128", FR, 128, 128, 128, 0, 128, "none"),
          ("spade gamma|beta 128^2 128->2x128", FR, 128, 128, 128, 0, 128, "spade"),
          ("spade shared 256^2 64->128", FR, 256, 256, 64, 0, 128, "none"),
          ("spade gamma|beta 256^2 128->2x64", FR, 256, 256, 128, 0, 64, "spade"),
          ("skip0 128^2 256+128->256", FR, 128, 128, 256, 128, 256, "none"),
          ("skip1 256^2 128+64->128", FR, 256, 256, 128, 64, 128, "none")]
if args.parity:
    SHAPES = [("tiny 8x8 32->64", 2, 8, 8, 32, 0, 64, "none"),
              ("ragged 19x45 32->64 res", 3, 19, 45, 32, 0, 64, "res"),
              ("1-row 1x70 64->64", 2, 1, 70, 64, 0, 64, "none"),
              ("ragged 33x17 32+32->128", 2, 33, 17, 32, 32, 128, "none"),
              ("two-input 50x34 64+32->64 res", 2, 50, 34, 64, 32, 64, "res"),
              ("two-input 37x41 96+32->128 spade", 2, 37, 41, 96, 32, 64, "spade"),
              ("spade 21x37 32->2x64", 2, 21, 37, 32, 0, 64, "spade"),
              ("spade 64^2 128->2x96 tanh", 2, 64, 64, 128, 0, 96, "spade"),
              ("res 40x40 160->192 mask", 2, 40, 40, 160, 0, 192, "mask"),
              ("sigmoid 35x35 32->64", 1, 35, 35, 32, 0, 64, "sig")]


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


def ref64(x0, x1, w, b, epi, kw, act):
    """fp64 reference from the same operands (NHWC in, NHWC out)."""
    x = x0 if x1 is None else torch.cat([x0, x1], 3)
    y = F.conv2d(x.permute(0, 3, 1, 2).double(), w.double().to(dev), b.double().to(dev), padding=1).permute(0, 2, 3, 1)
    if epi == "spade":
        Co = y.shape[3] // 2
        g, bt = y[..., :Co], y[..., Co:]
        xn = kw["xn"].double()
        y = (xn - kw["mean"].double()[:, None, None, :]) * kw["rstd"].double()[:, None, None, :] * (1 + g) + bt
    elif epi == "res":
        y = y + kw["res"].double()
    elif epi == "mask":
        return torch.where(kw["res"] > 0, y, torch.zeros_like(y))
    return {"relu": torch.relu, "tanh": torch.tanh, "sig": torch.sigmoid, "none": lambda t: t}[act](y)


tot = {"w4": 0.0, "w2": 0.0, "d": 0.0}
worst = 0.0
for idx, (tag, B, H, W, C0, C1, Co, epi) in enumerate(SHAPES):
    if args.only >= 0 and idx != args.only:
        continue
    Cin = C0 + C1
    x0 = rnd((B, H, W, C0), 1 + idx).to(dev)
    x1 = rnd((B, H, W, C1), 2 + idx).to(dev) if C1 else None
    act = "tanh" if "tanh" in tag else "sig" if epi == "sig" else "none" if epi == "mask" else "relu"
    kw = dict(act={"relu": ops.ACT_RELU, "tanh": ops.ACT_TANH, "sig": ops.ACT_SIGMOID, "none": ops.ACT_NONE}[act])
    if epi == "spade":
        wg, bg, wb, bb = rnd((Co, Cin, 3, 3), 3, 0.03), rnd((Co,), 4, 0.1), rnd((Co, Cin, 3, 3), 5, 0.03), rnd((Co,), 6, 0.1)
        spec = packing.spec_to(packing.pack_spade_gamma_beta(wg, bg, wb, bb), dev)
        wfull, bfull = torch.cat([wg, wb], 0), torch.cat([bg, bb], 0)
        xn = (rnd((B, H, W, Co), 7, 2.0) + 0.5).to(dev)
        mean = xn.reshape(B, -1, Co).mean(1).contiguous()
        rstd = (1 / torch.sqrt(xn.reshape(B, -1, Co).var(1, unbiased=False) + 1e-5)).contiguous()
        kw.update(epi=ops.EPI_SPADE, xn=xn, mean=mean, rstd=rstd)
        N = 2 * Co
    else:
        wfull, bfull = rnd((Co, Cin, 3, 3), 3, (Cin * 9) ** -0.5), rnd((Co,), 4, 0.1)
        spec = packing.spec_to(packing.pack_conv(wfull, bfull, stride=1, pad=1), dev)
        N = Co
        if epi == "res":
            kw.update(epi=ops.EPI_RESIDUAL, res=rnd((B, H, W, Co), 8).to(dev))
        elif epi == "mask":
            kw.update(epi=ops.EPI_RESIDUAL, res=rnd((B, H, W, Co), 8).to(dev), act=ops.ACT_RELU_MASK)
    y4, y2, yd = (torch.full((B, H, W, Co), float("nan"), device=dev) for _ in range(3))
    if args.ts:
        assert epi == "none"
        nwg = min(512, ((W + 31) // 32) * ((H + 15) // 16) * B * (N // 32)) if args.half else min(256, ((W + 31) // 32) * ((H + 15) // 16) * B * (N // 64))
        stamps = torch.zeros(nwg * 32, device=dev)
        ops.WINO4, ops.WINO4_MIN_CIN = True, 0
        with ops.conv_precision("winograd"):
            for _ in range(2):
                ops.conv2d(x0, spec, y4, x1=x1, act=ops.ACT_RELU, res=stamps)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        with ops.conv_precision("winograd"):
            ops.conv2d(x0, spec, y4, x1=x1, act=ops.ACT_RELU, res=stamps)
        e1.record()
        torch.cuda.synchronize()
        t = stamps.view(torch.int64).view(nwg, 16).cpu()
        tot_c, nb = (t[:, 12] - t[:, 11]).double(), t[:, 13].double()
        ms = e0.elapsed_time(e1)
        print(f"[ts] launch {ms * 1e3:.1f} us; per workgroup: {float(nb.mean()):.2f} blocks, {float(tot_c.median()):.0f} cycles entry -> exit (max {float(tot_c.max()):.0f}) = "
              f"{float(tot_c.median()) / max(float(nb.median()), 1):.0f} per block; counter rate {float(tot_c.max()) / ms / 1e3:.0f} MHz if the slowest workgroup spans the launch")
        ok = t[:, 3] > 0
        u = t[ok].double()
        nst = Cin // (4 if args.half else 8)
        print(f"[ts] {tag} B={B}: {int(ok.sum())} workgroups with a second block; medians (cycles): prologue {float((u[:, 1] - u[:, 0]).median()):.0f}  K loop {float((u[:, 2] - u[:, 1]).median()):.0f}"
              f" ({float((u[:, 2] - u[:, 1]).median()) / nst:.0f} per stage; ideal 4608)  epilogue {float((u[:, 3] - u[:, 2]).median()):.0f}")
        md = lambda i1, i0: float((u[:, i1] - u[:, i0]).median())
        print(f"     epilogue: fold {md(4, 2):.0f}, pass 0 writes + barriers {md(5, 4):.0f}, pass 0 output {md(6, 5):.0f}, "
              f"pass 1: barrier + writes + offsets {md(8, 6):.0f}, next block's set-up {md(9, 8):.0f}, its loads issued {md(10, 9):.0f}, barrier {md(7, 10):.0f}, output {md(3, 7):.0f}")
        continue

    def run(kind):
        if kind == "d":
            try:
                with ops.conv_precision("fp32"):
                    ops.conv2d(x0, spec, yd, x1=x1, **kw)
            except RuntimeError:                             # (shapes the direct kernel's contract does not take: SPADE with N % 128, C0 % 32 of two inputs)
                pass
        else:
            ops.WINO4, ops.WINO4_MIN_CIN = kind == "w4", 0
            with ops.conv_precision("winograd"):
                ops.conv2d(x0, spec, y4 if kind == "w4" else y2, x1=x1, **kw)

    if args.w4only:
        t4 = timeit(lambda: run("w4"), args.reps)
        ex = 2.0 * B * H * W * 2.25 * Cin * N
        tot["w4"] += t4
        print(f"{idx} {tag:36s} B={B:3d}: F(4,3) {t4 * 1e3:8.1f} us executed {ex / t4 / 1e9 / 157.3:.3f} of the pipe", flush=True)
        continue
    for k in ("w4", "w2", "d"):
        run(k)
    torch.cuda.synchronize()
    r = ref64(x0, x1, wfull, bfull, epi, kw, act)
    nrm = float(r.norm())
    err = {k: (float((t.double() - r).norm()) / nrm, float((t.double() - r).abs().max())) for k, t in (("w4", y4), ("w2", y2), ("d", yd))}
    bad = not bool(torch.isfinite(y4).all())
    worst = max(worst, err["w4"][1] / max(1.0, float(r.abs().max())))
    line = f"{idx} {tag:36s} B={B:3d}: rel L2 vs fp64: F(4,3) {err['w4'][0]:.2e} (max {err['w4'][1]:.1e}) F(2,3) {err['w2'][0]:.2e} direct {err['d'][0]:.2e}" + (" NON-FINITE" if bad else "")
    if not args.parity:
        t = {k: timeit(lambda k=k: run(k), args.reps) for k in ("w4", "w2", "d")}
        for k in t:
            tot[k] += t[k]
        ex, al = 2.0 * B * H * W * 2.25 * Cin * N, 2.0 * B * H * W * 9 * Cin * N
        line += (f" | F(4,3) {t['w4'] * 1e3:8.1f} us executed {ex / t['w4'] / 1e9 / 157.3:.3f} of the pipe, algorithmic {al / t['w4'] / 1e9:6.1f} TF/s"
                 f" | F(2,3) {t['w2'] * 1e3:8.1f} us (x{t['w2'] / t['w4']:.2f}) | direct {t['d'] * 1e3:8.1f} us (x{t['d'] / t['w4']:.2f})")
    print(line, flush=True)
if tot["w4"] and args.only < 0 and not args.w4only:
    print(f"sum: F(4,3) {tot['w4'] * 1e3:.1f} us, F(2,3) {tot['w2'] * 1e3:.1f} us (x{tot['w2'] / tot['w4']:.2f}), direct {tot['d'] * 1e3:.1f} us (x{tot['d'] / tot['w4']:.2f})")
print(f"worst max error of F(4,3) relative to max(1, |ref|): {worst:.2e}")
if args.parity:
    assert worst < 1e-4, worst
    print("parity ok")
