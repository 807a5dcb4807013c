#!/usr/bin/env python3
"""bf16lab - correctness and speed of the bf16 (activation-storage) kernels against the fp32 kernels on the layer shapes of
forward_tsf at 1024x1024, frame batch 2 (= 512x512, batch 8).   python tools/bf16lab.py [--shapes a,b] [--iters 20]
Prints per shape: max |d| / ref max of the bf16 kernel vs the fp32 kernel fed the SAME bf16-rounded operands (pure accumulation /
output-rounding differences: ~4e-3), microseconds and algorithmic TFLOP/s of both.  LWG_BF16_DMA_A=0/1 selects the A staging."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from ipercore_amd import ops  # noqa: E402
from ipercore_amd.networks import packing  # noqa: E402
from tools.convlab import SHAPES  # noqa: E402

DEV = "cuda:0"
BF = torch.bfloat16


def r16(t):
    return t.to(BF).float()


BATCH_MUL = 1


def build(name):
    B, H, W, C0, C1, N, k, stride, kind = SHAPES[name]
    B *= BATCH_MUL
    g = torch.Generator(device="cpu").manual_seed(sum(map(ord, name)))
    Cin = C0 + C1
    rnd = lambda *s, sc=1.0: r16((torch.randn(*s, generator=g) * sc)).to(DEV)  # noqa: E731
    x0 = rnd(B, H, W, C0)
    x1 = rnd(B, H, W, C1) if C1 else None
    launches = []
    if kind == "convT":
        w = r16(torch.randn(Cin, N, 4, 4, generator=g) * (Cin * 4) ** -0.5)
        yshape = (B, 2 * H, 2 * W, N)
        for s in packing.pack_conv_transpose(w, 0.1 * torch.randn(N, generator=g)):
            launches.append((packing.spec_to(s, DEV), dict(act=ops.ACT_RELU)))
    elif kind == "spade":
        wg = r16(torch.randn(N, Cin, 3, 3, generator=g) * (Cin * 9) ** -0.5)
        wb = r16(torch.randn(N, Cin, 3, 3, generator=g) * (Cin * 9) ** -0.5)
        spec = packing.spec_to(packing.pack_spade_gamma_beta(wg, 0.1 * torch.randn(N, generator=g), wb, 0.1 * torch.randn(N, generator=g)), DEV)
        yshape = (B, H, W, N)
        launches.append((spec, dict(epi=ops.EPI_SPADE, xn=rnd(B, H, W, N), mean=(torch.randn(B, N, generator=g) * 0.1).to(DEV),
                                    rstd=(torch.randn(B, N, generator=g) * 0.1 + 1.0).to(DEV))))
    elif kind == "small":
        w = r16(torch.randn(N, 6, k, k, generator=g) * (6 * k * k) ** -0.5)
        x0[..., 6:] = 0
        spec = packing.spec_to(packing.pack_conv(w, 0.1 * torch.randn(N, generator=g), stride=stride, cin_pad=8), DEV)
        yshape = (B, H // stride, W // stride, N)
        launches.append((spec, dict(act=ops.ACT_RELU)))
    else:
        w = r16(torch.randn(N, Cin, k, k, generator=g) * (Cin * k * k) ** -0.5)
        spec = packing.spec_to(packing.pack_conv(w, 0.1 * torch.randn(N, generator=g), stride=stride), DEV)
        yshape = (B, H // stride, W // stride, N)
        kw = dict(act=ops.ACT_RELU)
        if kind == "res":
            kw = dict(epi=ops.EPI_RESIDUAL, res=rnd(*yshape))
        launches.append((spec, kw))
    return x0, x1, yshape, launches


def run(x0, x1, y, launches, dt):
    if dt == BF and len(launches) == 4 and launches[0][0].omul == 2:        # a transposed convolution: the product call (fused when eligible)
        ops.conv_transpose2d(x0, [sp for sp, _ in launches], y, act=launches[0][1].get("act", ops.ACT_NONE))
        return
    for spec, kw in launches:
        kw2 = {k_: (v.to(dt) if k_ in ("res", "xn") else v) for k_, v in kw.items()}
        ops.conv2d(x0, spec, y, x1=x1, **kw2)


def timeit(fn, iters, warm_ms=20.0, hint_us=100.0):
    for _ in range(max(3, int(warm_ms * 1e3 / hint_us))):
        fn()
    s, t = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    t.record()
    torch.cuda.synchronize()
    return s.elapsed_time(t) * 1e3 / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shapes", default="res64,resres,gb64,shared64,skip1,skip0,up2,up1,up0,gb256,shared256,gb128,fq64,enc1")   # + fq128, fq256, enc0
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--no-f32", action="store_true")
    ap.add_argument("--convs-only", action="store_true")
    ap.add_argument("--batch-mul", type=int, default=1, help="multiply the batch of every shape (2 = frame batch 4 at 1024^2)")
    args = ap.parse_args()
    global BATCH_MUL
    BATCH_MUL = args.batch_mul
    ops.BF16_HR = os.environ.get("LWG_LAB_HR", "1") == "1"
    ops.BF16_PW = os.environ.get("LWG_LAB_PW", "1") == "1"
    ops.BF16_C8 = os.environ.get("LWG_LAB_C8", "1") == "1"
    ops.BF16_UP4 = os.environ.get("LWG_LAB_UP4", "1") == "1"
    print(f"batch x{BATCH_MUL} HR={ops.BF16_HR} PW={ops.BF16_PW} C8={ops.BF16_C8} UP4={ops.BF16_UP4} HALO={os.environ.get('LWG_BF16_HALO', '(default 1)')} BIG={os.environ.get('LWG_BF16_BIG', '(default 1)')} LWG_BF16_DMA_A={os.environ.get('LWG_BF16_DMA_A', '(default 1)')} TILE64={os.environ.get('LWG_BF16_TILE64', '(heuristic)')}")
    tot_f, tot_us = 0.0, 0.0
    for name in args.shapes.split(","):
        x0, x1, yshape, launches = build(name)
        M = yshape[0] * yshape[1] * yshape[2] // (launches[0][0].omul ** 2)
        flops = sum(2.0 * M * s.algo_kn for s, _ in launches)
        y32 = torch.full(yshape, float("nan"), device=DEV)
        run(x0, x1, y32, launches, torch.float32)
        y16 = torch.full(yshape, float("nan"), device=DEV, dtype=BF)
        x0b, x1b = (x0 if x0.shape[3] == 8 else x0.to(BF)), None if x1 is None else x1.to(BF)      # the first layer takes the fp32 input
        run(x0b, x1b, y16, launches, BF)
        torch.cuda.synchronize()
        ref = y32.abs().max().item()
        d = (y16.float() - y32).abs()
        ok = bool(torch.isfinite(y16.float()).all())
        us16 = timeit(lambda: run(x0b, x1b, y16, launches, BF), args.iters, hint_us=flops / 600e12 * 1e6)
        us32 = float("nan") if args.no_f32 else timeit(lambda: run(x0, x1, y32, launches, torch.float32), max(3, args.iters // 4), hint_us=flops / 130e12 * 1e6)
        byt = 2.0 * (x0.numel() + (0 if x1 is None else x1.numel()) + y16.numel())
        tot_f += flops
        tot_us += us16
        print(f"{name:10s} finite {ok}  max|d|/refmax {d.max().item() / ref:.2e} mean {d.mean().item() / ref:.1e} | bf16 {us16:8.1f} us {flops / us16 / 1e6:7.1f} TF/s "
              f"(HBM floor {byt / 8e12 * 1e6:6.1f} us, MFMA floor {flops / 2.5e15 * 1e6:6.1f} us) | fp32 {us32:8.1f} us {flops / us32 / 1e6:6.1f} TF/s", flush=True)
    print(f"sum: {tot_us:.0f} us, {tot_f / tot_us / 1e6:.0f} TF/s over the listed shapes")
    if args.convs_only:
        return
    # the HBM-bound bf16 kernels against their fp32 twins on the same bf16-rounded data
    g = torch.Generator(device="cpu").manual_seed(5)
    for (B, h, C, S, ns) in ((2, 256, 64, 512, 2), (2, 128, 128, 512, 2), (2, 64, 256, 512, 3)):
        q, Ks, Vs = (r16(torch.randn(n_, h, h, C, generator=g)).to(DEV) for n_ in (B, ns, ns))
        bk, bv = torch.randn(C, generator=g).to(DEV) * 0.1, torch.randn(C, generator=g).to(DEV) * 0.1
        T = (torch.rand(B, ns, S, S, 2, generator=g) * 2.2 - 1.1).to(DEV)
        T[:, :, : S // 5] = -2.0
        o32 = ops.lwb_attention(q, Ks, Vs, bk, bv, T, torch.empty_like(q))
        o16 = ops.lwb_attention(q.to(BF), Ks.to(BF), Vs.to(BF), bk, bv, T, torch.empty_like(q, dtype=BF))
        torch.cuda.synchronize()
        d = (o16.float() - o32).abs().max().item() / o32.abs().max().item()
        u16 = timeit(lambda: ops.lwb_attention(q.to(BF), Ks.to(BF), Vs.to(BF), bk, bv, T, o16), 10) if False else None
        qb, Kb, Vb = q.to(BF), Ks.to(BF), Vs.to(BF)
        u16 = timeit(lambda: ops.lwb_attention(qb, Kb, Vb, bk, bv, T, o16), 20)
        u32 = timeit(lambda: ops.lwb_attention(q, Ks, Vs, bk, bv, T, o32), 20)
        print(f"attention C={C} h={h}: max|d|/refmax {d:.2e} | bf16 {u16:.1f} us, fp32 {u32:.1f} us")
        mean16, rstd16, mean32, rstd32 = (torch.empty(B, C, device=DEV) for _ in range(4))
        nsplit = max(1, min(64, h * h // 64))
        ws = torch.empty(B * C * nsplit * 3, device=DEV)
        ops.instnorm_stats(q, mean32, rstd32, ws, nsplit=nsplit)
        ops.instnorm_stats(qb, mean16, rstd16, ws, nsplit=nsplit)
        torch.cuda.synchronize()
        u16 = timeit(lambda: ops.instnorm_stats(qb, mean16, rstd16, ws, nsplit=nsplit), 20)
        u32 = timeit(lambda: ops.instnorm_stats(q, mean32, rstd32, ws, nsplit=nsplit), 20)
        print(f"instnorm  C={C} h={h}: mean |d| {float((mean16 - mean32).abs().max()):.1e} rstd |d| {float((rstd16 - rstd32).abs().max()):.1e} | bf16 {u16:.1f} us, fp32 {u32:.1f} us")
    for (B, S) in ((2, 1024), (3, 200)):
        x = r16(torch.randn(B, S, S, 64, generator=g)).to(DEV)
        wi, wa = r16(torch.randn(3, 64, 5, 5, generator=g) * 0.03), r16(torch.randn(1, 64, 5, 5, generator=g) * 0.03)
        bg = torch.randn(1, 3, S, S, generator=g).to(DEV)
        p32, m32, i32 = ops.head_compose(x, packing.pack_head(wi, wa).to(DEV), bg, True, True, True)
        w16 = packing.pack_head_bf16(wi, wa).to(DEV)
        xb = x.to(BF)
        p16, m16, i16 = ops.head_compose(xb, w16, bg, True, True, True)
        torch.cuda.synchronize()
        u16 = timeit(lambda: ops.head_compose(xb, w16, bg, True, True, False), 20)
        u32 = timeit(lambda: ops.head_compose(x, packing.pack_head(wi, wa).to(DEV), bg, True, True, False), 5)
        print(f"head B={B} S={S}: pred |d| {float((p16 - p32).abs().max()):.2e} mask |d| {float((m16 - m32).abs().max()):.2e} img |d| {float((i16 - i32).abs().max()):.2e} "
              f"| bf16 {u16:.1f} us, fp32 {u32:.1f} us")


if __name__ == "__main__":
    main()
