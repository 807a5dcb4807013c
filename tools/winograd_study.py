"""Study (CPU, oracle): what an F(2x2, 3x3) Winograd form of the 3x3 stride-1 convolutions would do to the numbers.
Every 3x3 / stride 1 / pad 1 convolution with Cin % 32 == 0 of the oracle's per-frame path is computed three ways from the SAME fp32
inputs - direct fp32, Winograd fp32 (transforms, the 16 channel contractions and the inverse transform all in fp32), the bf16x6 form of
the "split" precision mode (emulated), direct fp64 - and
the relative L2 error of the two fp32 forms against fp64 is recorded per layer; the frame is rendered once with direct and once with
Winograd convolutions.  usage: python tools/winograd_study.py [S=256]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.nn.functional as F

from tests import parity_utils as pu

G = torch.tensor([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=torch.float32)
BT = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=torch.float32)
AT = torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=torch.float32)
_orig = F.conv2d
STATS, MODE = [], {"form": "direct"}


def winograd(x, w, b):
    n, c, h, wd = x.shape
    U = torch.einsum("ij,ocjk,lk->ocil", G, w, G)                                   # (O,C,4,4)
    xp = F.pad(x, [1, 1 + (wd & 1), 1, 1 + (h & 1)])
    t = xp.unfold(2, 4, 2).unfold(3, 4, 2)                                          # (n,C,th,tw,4,4)
    V = torch.einsum("ij,ncthjk,lk->ncthil", BT, t, BT)
    M = torch.einsum("ocil,ncthil->nothil", U, V)
    Y = torch.einsum("pi,nothil,ql->nothpq", AT, M, AT)                             # (n,O,th,tw,2,2)
    y = Y.permute(0, 1, 2, 4, 3, 5).reshape(n, w.shape[0], Y.shape[2] * 2, Y.shape[3] * 2)[:, :, :h, :wd]
    return y if b is None else y + b.view(1, -1, 1, 1)


def conv2d(x, w, b=None, stride=1, padding=0, *a, **k):
    st = stride if isinstance(stride, int) else stride[0]
    pd = padding if isinstance(padding, int) else padding[0]
    if tuple(w.shape[-2:]) != (3, 3) or st != 1 or pd != 1 or w.shape[1] % 32 or x.dtype != torch.float32 or a or k:
        return _orig(x, w, b, stride, padding, *a, **k)
    d32 = _orig(x, w, b, 1, 1)
    w32 = winograd(x, w, b)
    s32 = split6(x, w, b)
    d64 = _orig(x.double(), w.double(), None if b is None else b.double(), 1, 1)
    nrm = d64.norm().item()
    STATS.append((tuple(x.shape[1:]), w.shape[0], (d32.double() - d64).norm().item() / nrm, (w32.double() - d64).norm().item() / nrm,
                  (w32 - d32).abs().max().item(), d64.abs().max().item(), (s32.double() - d64).norm().item() / nrm))
    return {"direct": d32, "wino": w32, "split6": s32}[MODE["form"]]


def _split3(t):
    hi = t.to(torch.bfloat16).float()
    r = t - hi
    mid = r.to(torch.bfloat16).float()
    return hi, mid, (r - mid).to(torch.bfloat16).float()


def split6(x, w, b):
    """The bf16x6 form of csrc/conv_igemm_split.hip emulated: both operands split exactly into three bf16 planes, the six largest partial
    products (each bf16 x bf16 product is exact in fp32), fp32 accumulation."""
    xh, xm, xl = _split3(x)
    wh, wm, wl = _split3(w)
    y = _orig(xl, wh, None, 1, 1) + _orig(xh, wl, None, 1, 1) + _orig(xm, wm, None, 1, 1)
    y = y + _orig(xm, wh, None, 1, 1) + _orig(xh, wm, None, 1, 1)
    y = y + _orig(xh, wh, None, 1, 1)
    return y if b is None else y + b.view(1, -1, 1, 1)


def main():
    S = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    case = pu.build_case(image_size=S, n_frames=1, ns=2)
    F.conv2d = conv2d
    try:
        MODE["form"] = "direct"
        direct = pu.run_oracle(case, frames=[0])
        n_direct = len(STATS)
        MODE["form"] = "wino"
        wino = pu.run_oracle(case, frames=[0])
        MODE["form"] = "split6"
        spl = pu.run_oracle(case, frames=[0])
    finally:
        F.conv2d = _orig
    st = STATS[:n_direct]
    e_d, e_w, e_s = np.array([s[2] for s in st]), np.array([s[3] for s in st]), np.array([s[6] for s in st])
    print(f"{len(st)} 3x3 convolutions at {S}x{S} (per-frame path + source side), relative L2 error vs fp64 of the same inputs:")
    print(f"  direct fp32   median {np.median(e_d):.2e}  max {e_d.max():.2e}")
    print(f"  Winograd fp32 median {np.median(e_w):.2e}  max {e_w.max():.2e}   ratio of medians {np.median(e_w) / np.median(e_d):.1f}x")
    print(f"  bf16x6 (emu)  median {np.median(e_s):.2e}  max {e_s.max():.2e}   ratio of medians {np.median(e_s) / np.median(e_d):.1f}x")
    worst = max(st, key=lambda s: s[3])
    print(f"  worst Winograd layer: in {worst[0]} -> {worst[1]} channels, max |wino - direct| {worst[4]:.2e} at max |y| {worst[5]:.2e}")
    for name, v in (("Winograd", wino), ("bf16x6", spl)):
        d = (v - direct).abs()
        mse = (d.double() ** 2).mean().item()
        print(f"frame rendered with {name} convolutions vs direct: max |d| {d.max().item():.2e}, mean {d.mean().item():.2e}, "
              f"PSNR {10 * np.log10(4.0 / max(mse, 1e-30)):.1f} dB (frames in [-1, 1]); the parity tests allow 2e-3 max / 1e-4 mean")


if __name__ == "__main__":
    main()
