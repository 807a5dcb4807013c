#!/usr/bin/env python3
"""convlab - time variant builds of the implicit-GEMM conv kernel on the shapes of forward_tsf @512 (frame batch 8).

    python tools/convlab.py [--shapes a,b,..] [--iters 10] LIB [LIB ...]

Every LIB is a shared object exporting ``lwg_conv2d_nhwc_f32`` (the product library or an experimental build under
tools/lab/).  The first LIB is the baseline; outputs of the others are compared with it (max |d|).  Prints one row
per (shape, lib): microseconds per launch and algorithmic TFLOP/s.  GPU only; run through tools via gpurun.
"""
import argparse
import ctypes
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from ipercore_amd import _lib, ops  # noqa: E402
from ipercore_amd.networks import packing  # noqa: E402

DEV = "cuda:0"

# name: (B, H, W, C0, C1, N, k, stride, kind)   kind: conv | convT | spade | res | small
SHAPES = {
    "res64":    (8, 64, 64, 256, 0, 256, 3, 1, "conv"),       # res blocks, 72 launches / step
    "gb64":     (8, 64, 64, 128, 0, 256, 3, 1, "spade"),      # SPADE gamma|beta at 64^2 (N = 512), 42
    "shared64": (8, 64, 64, 256, 0, 128, 3, 1, "conv"),       # SPADE shared at 64^2, 42
    "skip1":    (8, 256, 256, 64, 128, 128, 3, 1, "conv"),    # decoder skipper 1 (two-pointer concat)
    "skip0":    (8, 128, 128, 128, 256, 256, 3, 1, "conv"),   # decoder skipper 0
    "up2":      (8, 256, 256, 128, 0, 64, 4, 2, "convT"),     # upconv 2 (4 parity launches)
    "up1":      (8, 128, 128, 256, 0, 128, 4, 2, "convT"),
    "up0":      (8, 64, 64, 256, 0, 256, 4, 2, "convT"),
    "gb256":    (8, 256, 256, 128, 0, 64, 3, 1, "spade"),     # enc site 0 gamma|beta (N = 128)
    "shared256": (8, 256, 256, 64, 0, 128, 3, 1, "conv"),
    "gb128":    (8, 128, 128, 128, 0, 128, 3, 1, "spade"),
    "fq64":     (8, 64, 64, 256, 0, 256, 1, 1, "conv"),       # 1x1
    "fq128":    (8, 128, 128, 128, 0, 128, 1, 1, "conv"),
    "fq256":    (8, 256, 256, 64, 0, 64, 1, 1, "conv"),
    "enc1":     (8, 256, 256, 64, 0, 128, 3, 2, "conv"),      # stride-2 encoder
    "enc0":     (8, 512, 512, 8, 0, 64, 3, 2, "small"),       # first layer, Cin 6 -> 8
    "resres":   (8, 64, 64, 256, 0, 256, 3, 1, "res"),        # residual epilogue
    # one training sample / 1-frame batches
    "res64_b1": (1, 64, 64, 256, 0, 256, 3, 1, "conv"),
    "res64_b2": (2, 64, 64, 256, 0, 256, 3, 1, "conv"),
    "sh128_b1": (1, 128, 128, 128, 0, 128, 3, 1, "conv"),
    "skip1_b1": (1, 256, 256, 64, 128, 128, 3, 1, "conv"),
}


def load(path):
    h = ctypes.CDLL(os.path.abspath(path))
    for sym in ("lwg_conv2d_nhwc_f32", "lwg_conv2d_nhwc_bf16mma", "lwg_conv2d_nhwc_f32_split"):
        if hasattr(h, sym):
            getattr(h, sym).restype = ctypes.c_int
            getattr(h, sym).argtypes = [ctypes.POINTER(_lib.LwgConvArgs), ctypes.c_void_p]
    return h


def build_case(name):
    B, H, W, C0, C1, N, k, stride, kind = SHAPES[name]
    g = torch.Generator(device="cpu").manual_seed(sum(map(ord, name)))
    Cin = C0 + C1
    rnd = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).to(DEV)  # noqa: E731
    x0 = rnd(B, H, W, C0)
    x1 = rnd(B, H, W, C1) if C1 else None
    launches = []
    if kind == "convT":
        w = rnd(Cin, N, 4, 4, sc=(Cin * 4) ** -0.5).cpu()
        y = torch.empty(B, 2 * H, 2 * W, N, device=DEV)
        for s in packing.pack_conv_transpose(w, torch.zeros(N)):
            launches.append((packing.spec_to(s, DEV), dict(act=ops.ACT_RELU)))
    elif kind == "spade":
        wg = rnd(N, Cin, 3, 3, sc=(Cin * 9) ** -0.5).cpu()
        wb = rnd(N, Cin, 3, 3, sc=(Cin * 9) ** -0.5).cpu()
        spec = packing.spec_to(packing.pack_spade_gamma_beta(wg, torch.zeros(N), wb, torch.zeros(N)), DEV)
        y = torch.empty(B, H, W, N, device=DEV)
        xn = rnd(B, H, W, N)
        mean, rstd = rnd(B, N, sc=0.1), rnd(B, N, sc=0.1) + 1.0
        launches.append((spec, dict(epi=ops.EPI_SPADE, xn=xn, mean=mean, rstd=rstd)))
    else:
        w = rnd(N, Cin if kind != "small" else 6, k, k, sc=(Cin * k * k) ** -0.5).cpu()
        spec = packing.spec_to(packing.pack_conv(w, torch.zeros(N), stride=stride, cin_pad=8 if kind == "small" else None), DEV)
        y = torch.empty(B, H // stride, W // stride, N, device=DEV)
        kw = dict(act=ops.ACT_RELU)
        if kind == "res":
            kw = dict(epi=ops.EPI_RESIDUAL, res=rnd(B, H // stride, W // stride, N))
        launches.append((spec, kw))
    return x0, x1, y, launches


BF16 = False      # False: fp32 MFMA; True: bf16 operands; "split": bf16x6


def run(h, x0, x1, y, launches, stream):
    for spec, kw in launches:
        a = ops.conv_args(x0, spec, y, x1=x1, **kw)
        if BF16 == "split" and spec.Cin % 32 == 0:
            a.w = ops._w16x3(spec).data_ptr()
            e = h.lwg_conv2d_nhwc_f32_split(a, stream)
        elif BF16 is True and spec.Cin % 32 == 0:
            a.w = ops._w16(spec).data_ptr()
            e = h.lwg_conv2d_nhwc_bf16mma(a, stream)
        else:
            e = h.lwg_conv2d_nhwc_f32(a, stream)
        if e != 0:
            return e
    return 0


def ref64(name, x0, x1, launches):
    """fp64 convolution of the case on the CPU (plain conv / res kinds only), NHWC, or None."""
    B, H, W, C0, C1, N, k, stride, kind = SHAPES[name]
    if kind not in ("conv", "res") or 2.0 * B * H * W * k * k * (C0 + C1) * N / stride ** 2 > 4e10:
        return None
    import torch.nn.functional as F
    spec, kw = launches[0]
    K4 = spec.w.shape[0]
    wk = spec.w.permute(0, 2, 1).reshape(K4 * 4, N).cpu().double()          # kernel K order -> (tap, c, n)
    Cin = C0 + C1
    w = wk.view(Cin // 32, k * k, 32, N).permute(3, 0, 2, 1).reshape(N, Cin, k, k)
    x = (x0 if x1 is None else torch.cat([x0, x1], dim=3)).cpu().double().permute(0, 3, 1, 2)
    y = F.conv2d(x, w, stride=stride, padding=k // 2).permute(0, 2, 3, 1)
    if kind == "res":
        y = y + kw["res"].cpu().double()
    elif kw.get("act") == ops.ACT_RELU:
        y = y.clamp_min(0)
    return y


def timestamps(h, x0, x1, y, launches, stream):
    """Lab builds with -DLWG_LAB_TS record s_memtime at kernel entry / K-loop entry / K-loop exit / end per workgroup."""
    import numpy as np
    run(h, x0, x1, y, launches, stream)
    torch.cuda.synchronize()
    spec = launches[0][0]
    M = y.shape[0] * y.shape[1] * y.shape[2] // (spec.omul ** 2)
    nwg = min(8192, ((M + 127) // 128) * (spec.N // (128 if spec.N % 128 == 0 else 64)))
    buf = (ctypes.c_ulonglong * (nwg * 8))()
    h.lwg_lab_read_ts.argtypes = [ctypes.c_void_p, ctypes.c_int]
    assert h.lwg_lab_read_ts(buf, nwg * 8) == 0
    raw = np.frombuffer(buf, dtype=np.uint64).reshape(nwg, 8).astype(np.int64)
    cyc, wall = raw[:, :4], raw[:, 4:] / 100.0          # s_memtime ticks; s_memrealtime at 100 MHz -> us
    wall = wall - wall[:, 0].min()
    dc, dw = np.diff(cyc, axis=1), np.diff(wall, axis=1)
    ghz = (cyc[:, 3] - cyc[:, 0]) / np.maximum(wall[:, 3] - wall[:, 0], 1e-3) / 1e3
    q = lambda v: "/".join(f"{np.percentile(v, p):.1f}" for p in (5, 50, 95))  # noqa: E731
    print(f"    ts[{nwg} wgs] wall us p5/p50/p95: prologue {q(dw[:, 0])}  kloop {q(dw[:, 1])}  epilogue {q(dw[:, 2])}  "
          f"start {q(wall[:, 0])}  end {q(wall[:, 3])}")
    print(f"    memtime ticks p50: prologue {np.median(dc[:, 0]):.0f} kloop {np.median(dc[:, 1]):.0f} epilogue {np.median(dc[:, 2]):.0f}"
          f"  ticks/us p50 {np.median(ghz) * 1e3:.0f}")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("libs", nargs="+")
    ap.add_argument("--shapes", default=",".join(SHAPES))
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--json", default=None)
    ap.add_argument("--wgrad", action="store_true", help="time lwg_conv2d_wgrad_nhwc_f32 (weight gradient) of the first launch instead")
    ap.add_argument("--bf16", action="store_true", help="after the fp32 pass of every lib, time its bf16-operand entry point too")
    ap.add_argument("--split", action="store_true", help="also time the bf16x6 (exact-split) entry point of every lib that has one")
    ap.add_argument("--ref64", action="store_true", help="error of every variant against an fp64 convolution on the CPU (small plain-conv shapes)")
    args = ap.parse_args()
    libs = [(os.path.basename(p), load(p)) for p in args.libs]
    stream = torch.cuda.current_stream().cuda_stream
    rows = []
    for name in args.shapes.split(","):
        x0, x1, y, launches = build_case(name)
        M = y.shape[0] * y.shape[1] * y.shape[2] // (launches[0][0].omul ** 2)      # GEMM rows per launch
        flops = sum(2.0 * M * s.algo_kn for s, _ in launches)
        base = None
        us_hint = flops / 100e12 * 1e6
        global BF16
        todo = [(n_, h_, False) for n_, h_ in libs if hasattr(h_, "lwg_conv2d_nhwc_f32")]
        todo += [(n_ + " [bf16]", h_, True) for n_, h_ in libs if hasattr(h_, "lwg_conv2d_nhwc_bf16mma")] if args.bf16 else []
        todo += [(n_ + " [x6]", h_, "split") for n_, h_ in libs if hasattr(h_, "lwg_conv2d_nhwc_f32_split")] if args.split else []
        y64 = ref64(name, x0, x1, launches) if args.ref64 else None
        if args.wgrad:
            spec, kw = launches[0]
            dy = torch.randn_like(y)
            Ktot = spec.ntaps * spec.Cin
            for lname, h in libs:
                h.lwg_conv2d_wgrad_nhwc_f32.restype = ctypes.c_int
                h.lwg_conv2d_wgrad_nhwc_f32.argtypes = [ctypes.POINTER(_lib.LwgConvArgs)] + [ctypes.c_void_p] * 4
                h.lwg_conv2d_wgrad_ws_floats.restype = ctypes.c_size_t
                a = ops.conv_args(x0, spec, dy, x1=x1)
                dw = torch.empty(Ktot, spec.N, device=DEV)
                ws = torch.empty(h.lwg_conv2d_wgrad_ws_floats(Ktot, spec.N, a.M), device=DEV)
                call = lambda: h.lwg_conv2d_wgrad_nhwc_f32(a, dy.data_ptr(), dw.data_ptr(), ws.data_ptr(), stream)   # noqa: E731
                assert call() == 0
                for _ in range(max(2, int(20000 / max(us_hint, 50.0)))):
                    call()
                s_, t_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s_.record()
                for _ in range(args.iters):
                    call()
                t_.record()
                torch.cuda.synchronize()
                us = s_.elapsed_time(t_) * 1e3 / args.iters
                f1 = 2.0 * a.M * spec.algo_kn
                print(f"{name:10s} {lname:24s} wgrad {us:9.1f} us  {f1 / us / 1e6:6.1f} TF/s  ({f1 / us / 1e6 / 157.3 * 100:4.1f}%)  ws {ws.numel() * 4 / 1e6:.0f} MB", flush=True)
            continue
        for lname, h, BF16 in todo:
            y.fill_(float("nan"))
            e = run(h, x0, x1, y, launches, stream)
            torch.cuda.synchronize()
            if e != 0:
                print(f"{name:10s} {lname:28s} ERROR {e}")
                continue
            out = y.clone()
            if base is None:
                base = out
                diff = 0.0
            else:
                diff = (out - base).abs().max().item() if torch.isfinite(out).all() else float("nan")
            for _ in range(max(2, int(20000 / max(us_hint, 50.0)))):      # >= 20 ms of warm-up: clocks settle
                run(h, x0, x1, y, launches, stream)
            s, t = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(args.iters):
                run(h, x0, x1, y, launches, stream)
            t.record()
            torch.cuda.synchronize()
            us = s.elapsed_time(t) * 1e3 / args.iters
            tf = flops / (us * 1e-6) / 1e12
            rows.append({"shape": name, "lib": lname, "us": round(us, 1), "tflops": round(tf, 1), "maxdiff": diff})
            acc = ""
            if y64 is not None:
                d = out.cpu().double() - y64
                rms = y64.pow(2).mean().sqrt().item()
                rows[-1]["err64_max_over_rms"], rows[-1]["err64_rms_over_rms"] = d.abs().max().item() / rms, d.pow(2).mean().sqrt().item() / rms
                acc = f"  vs fp64: max {rows[-1]['err64_max_over_rms']:.2e} rms {rows[-1]['err64_rms_over_rms']:.2e} (of ref rms)"
            print(f"{name:10s} {lname:28s} {us:9.1f} us  {tf:6.1f} TF/s  ({tf / 157.3 * 100:4.1f}%)  maxdiff {diff:.2e}{acc}", flush=True)
            if hasattr(h, "lwg_lab_read_ts"):
                timestamps(h, x0, x1, y, launches[:1], stream)
    if args.json:
        with open(args.json, "w") as fp:
            json.dump(rows, fp, indent=1)


if __name__ == "__main__":
    main()
